// Weight-gradient GEMM of the training step on TOKEN-MAJOR operands (gfx950):
//   dW[n, k] = sum_t dY[t, n] * X[t, k]      (+ db[n] = sum_t dY[t, n])
// i.e. the autograd of nn.Linear's weight / bias (models/dit.py:220-232, 256,
// 277, 306) under diffusion.py:390-397.  Both operands are read as the forward
// and backward kernels leave them -- rows = tokens, the contraction index --
// so no transposed copies are made: 64-token slabs of dY[.., 256 n] and
// X[.., 256 k] are staged by LDS-DMA (global_load_lds, 16-byte chunks) and the
// MFMA fragments, which need 8 consecutive TOKENS of one column per lane, are
// gathered with the gfx950 transposing LDS read ds_read_b64_tr_b16 (a 16-lane
// group reads 4 token rows x 16 columns and receives them column-major).
//
// LDS image of a slab: row = token (512 B = 256 columns); the 32-byte block
// index b of a row is stored at b ^ fx(row), fx = (row & 3) | ((row >> 3) & 1) << 2:
// the eight token rows a 32-lane half reads in one instruction ({r..r+3} of two
// groups 8 rows apart) then land in eight different 32-byte bank groups
// (conflict-free).  The swizzle is applied on the SOURCE address of the DMA.
//
// Split-K over the token dimension as in mdlm_gemm_bf16_splitk: the output is
// small (768x768 .. 3072x768), so `splits` token ranges run as independent
// tiles writing fp32 partial planes (mdlm_splitk_reduce adds them in order).
// 256x256 output tile, 8 waves (2 n x 4 k), wave tile 128 n x 64 k, MFMA
// 16x16x32 issued as X_frag x dY_frag: a lane then holds 4 consecutive k of one
// n and stores them as one float4.
#include "common.h"
#include <type_traits>

namespace mdlm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct TnArgs {
  const bf16_t* Y; int64_t ldy;   // dY (T, N)
  const bf16_t* X; int64_t ldx;   // X  (T, K)
  float* P;                       // partial planes [splits][N][K]
  float* Pb;                      // bias partials [splits][N] or null
  int N, K;
  int slabs_total;                // 64-token slabs in all
  int slabs_per_split;            // the last split may hold fewer
  int splits, tiles_n, tiles_k;
};

#define TN_GLDS16(gptr, lptr) \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr), \
                                   (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

constexpr int TN_SLAB = 64;                 // tokens per slab
constexpr int TN_OP_BYTES = TN_SLAB * 512;  // one operand slab: 64 rows x 256 columns bf16
constexpr int TN_STAGE = 2 * TN_OP_BYTES;   // dY slab + X slab

// One fragment = two transposing reads (tokens 8g..8g+3 and 8g+4..8g+7 of this
// lane's column).  Issued as inline asm: through the builtin hipcc treats the
// read as possibly aliasing the LDS-DMA in flight and puts s_waitcnt vmcnt(0) in
// front of it, which serialises the next slab's DMA with this slab's MFMAs.
// The asm reads are invisible to the compiler's waitcnt pass, so the kernel
// waits on lgkmcnt itself (tr_wait) before the MFMAs that consume them; LDS
// reads return in order, so "at most N outstanding" means all but the last N.
#define TR_READ(dst, addr, off) \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")

template <int OFF>
__device__ __forceinline__ void tr_frag_issue(s16x4& lo, s16x4& hi, unsigned addr) {
  TR_READ(lo, addr, OFF);
  TR_READ(hi, addr, OFF + 4 * 512);
}

__device__ __forceinline__ bf16x8 frag_join(s16x4 lo, s16x4 hi) {
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(512, 1) void gemm_tn_splitk_kernel(TnArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;
  // each XCD (workgroup id & 7) walks a contiguous range of the tile sequence
  const int per_split = g.tiles_n * g.tiles_k;
  const int total = per_split * g.splits;
  const int slots = gridDim.x >> 3;
  const int per_xcd = (total + 7) >> 3;
  const int s_begin = (blockIdx.x & 7) * per_xcd;
  const int s_end = total < s_begin + per_xcd ? total : s_begin + per_xcd;

  // ---- staging: DMA instruction c (0..31) of an operand covers token rows 2c, 2c+1
  const int srow = lane >> 5;          // row within the pair
  const int pch = lane & 31;           // physical 16-byte chunk of the row
  // instruction i of this wave covers rows row0 + 16*i: the swizzle term
  // (row & 3) | ((row >> 3) & 1) << 2 is the same for all four, so one per-lane
  // pointer per operand serves them (the 16*i*ld row offsets are wave-uniform)
  const int row0 = 2 * wid + srow;
  const int soff = (pch ^ (((row0 & 3) | (((row0 >> 3) & 1) << 2)) << 1)) * 8;
  const bf16_t* y_src;
  const bf16_t* x_src;
  auto set_src = [&](int tn, int tk, int sp) {
    const int64_t t0 = (int64_t)sp * g.slabs_per_split * TN_SLAB;
    y_src = g.Y + (t0 + row0) * g.ldy + tn * 256 + soff;
    x_src = g.X + (t0 + row0) * g.ldx + tk * 256 + soff;
  };
  auto stage = [&](int buf, int slab) {
    char* const sb = smem + buf * TN_STAGE;
    const int64_t yo = (int64_t)slab * TN_SLAB * g.ldy, xo = (int64_t)slab * TN_SLAB * g.ldx;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      TN_GLDS16(y_src + yo + (int64_t)(16 * i) * g.ldy, sb + (i * 8 + wid) * 1024);
      TN_GLDS16(x_src + xo + (int64_t)(16 * i) * g.ldx, sb + TN_OP_BYTES + (i * 8 + wid) * 1024);
    }
  };

  // ---- fragment read addresses (k-step ks adds ks*32 rows, the second half +4 rows)
  const int gq = lane >> 4;            // 16-lane group: tokens 8*gq .. 8*gq+7 of the k-step
  const int q = (lane & 15) >> 2, p = lane & 3;
  const int fx = q | ((gq & 1) << 2);  // swizzle of row 8*gq + q (+4, +32: unchanged)
  int ybase[8], xbase[4];
#pragma unroll
  for (int m = 0; m < 8; ++m)
    ybase[m] = (8 * gq + q) * 512 + (((wr * 8 + m) ^ fx) * 32) + p * 8;
#pragma unroll
  for (int n = 0; n < 4; ++n)
    xbase[n] = TN_OP_BYTES + (8 * gq + q) * 512 + (((wc * 4 + n) ^ fx) * 32) + p * 8;

  // LDS byte offset of smem (the asm reads take raw LDS addresses)
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  int j = blockIdx.x >> 3;
  if (s_begin + j >= s_end) return;
  auto decode = [&](int s, int& tn, int& tk, int& sp) {
    sp = s / per_split;
    const int r = s - sp * per_split;
    tn = r / g.tiles_k;
    tk = r - tn * g.tiles_k;
  };
  int tn, tk, sp;
  decode(s_begin + j, tn, tk, sp);
  set_src(tn, tk, sp);
  stage(0, 0);
  int buf = 0;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // explicit drain of the LDS-DMA
  __syncthreads();

  while (true) {
    const int s_next = s_begin + j + slots;
    const bool has_next = s_next < s_end;
    int tn2 = 0, tk2 = 0, sp2 = 0;
    if (has_next) decode(s_next, tn2, tk2, sp2);
    const bool do_bias = g.Pb != nullptr && tk == 0 && wc == 0;   // wave-uniform
    const int left = g.slabs_total - sp * g.slabs_per_split;
    const int n_slabs = left < g.slabs_per_split ? left : g.slabs_per_split;

    f32x4 acc[8][4];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) bsum[m] = 0.f;

    auto slab_loop = [&](auto with_bias) {
      constexpr bool BIAS = decltype(with_bias)::value;
      for (int sl = 0; sl < n_slabs; ++sl) {
        if (sl + 1 < n_slabs) {
          stage(buf ^ 1, sl + 1);
        } else if (has_next) {
          set_src(tn2, tk2, sp2);
          stage(buf ^ 1, 0);
        }
        const unsigned sb = lds0 + buf * TN_STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          s16x4 xl[4], xh[4], yl[8], yh[8];
          // group A: the X fragments and dY fragments 0..3 (16 reads), group B: dY 4..7 (8 reads)
#pragma unroll
          for (int n = 0; n < 4; ++n) {
            if (ks == 0) tr_frag_issue<0>(xl[n], xh[n], sb + xbase[n]);
            else tr_frag_issue<32 * 512>(xl[n], xh[n], sb + xbase[n]);
          }
#pragma unroll
          for (int m = 0; m < 8; ++m) {
            if (ks == 0) tr_frag_issue<0>(yl[m], yh[m], sb + ybase[m]);
            else tr_frag_issue<32 * 512>(yl[m], yh[m], sb + ybase[m]);
          }
          asm volatile("s_waitcnt lgkmcnt(8)"
                       : "+v"(xl[0]), "+v"(xh[0]), "+v"(xl[1]), "+v"(xh[1]), "+v"(xl[2]), "+v"(xh[2]),
                         "+v"(xl[3]), "+v"(xh[3]), "+v"(yl[0]), "+v"(yh[0]), "+v"(yl[1]), "+v"(yh[1]),
                         "+v"(yl[2]), "+v"(yh[2]), "+v"(yl[3]), "+v"(yh[3]));
          bf16x8 xf[4];
#pragma unroll
          for (int n = 0; n < 4; ++n) xf[n] = frag_join(xl[n], xh[n]);
#pragma unroll
          for (int m = 0; m < 8; ++m) {
            if (m == 4) {
              // the MFMAs of m = 0..3 stay in front of this wait (they cover group B's latency)
              __builtin_amdgcn_sched_barrier(0);
              asm volatile("s_waitcnt lgkmcnt(0)"
                           : "+v"(yl[4]), "+v"(yh[4]), "+v"(yl[5]), "+v"(yh[5]), "+v"(yl[6]), "+v"(yh[6]),
                             "+v"(yl[7]), "+v"(yh[7]));
            }
            const bf16x8 yf = frag_join(yl[m], yh[m]);
#pragma unroll
            for (int n = 0; n < 4; ++n)
              acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[n], yf, acc[m][n], 0, 0, 0);
            if constexpr (BIAS) {
              const s16x8 v = __builtin_bit_cast(s16x8, yf);
              float t = 0.f;
#pragma unroll
              for (int e = 0; e < 8; ++e) t += __uint_as_float((unsigned)(unsigned short)v[e] << 16);
              bsum[m] += t;
            }
          }
        }
        // this slab's MFMAs stay in front of the barrier (and of its vmcnt(0)): the
        // next slab's DMA then has the whole slab's compute to land
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next slab's LDS-DMA landed
        __syncthreads();
        buf ^= 1;
      }
    };
    if (do_bias) slab_loop(std::true_type{}); else slab_loop(std::false_type{});

    // ---- epilogue: acc[m][n][e] = dW[n0 + m*16 + (lane&15)][k0 + n*16 + 4*gq + e]
    float* const plane = g.P + (int64_t)sp * g.N * g.K;
    const int64_t nrow0 = (int64_t)tn * 256 + wr * 128 + (lane & 15);
    const int kcol0 = tk * 256 + wc * 64 + 4 * gq;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      float* const rowp = plane + (nrow0 + m * 16) * g.K + kcol0;
#pragma unroll
      for (int n = 0; n < 4; ++n)
        *reinterpret_cast<f32x4*>(rowp + n * 16) = acc[m][n];
    }
    if (do_bias) {
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        float t = bsum[m];
        t += __shfl_xor(t, 16);
        t += __shfl_xor(t, 32);
        if (gq == 0) g.Pb[(int64_t)sp * g.N + nrow0 + m * 16] = t;
      }
    }
    if (!has_next) break;
    j += slots; tn = tn2; tk = tk2; sp = sp2;
  }
}

}  // namespace mdlm

using namespace mdlm;

extern "C" int mdlm_gemm_tn_splitk(const void* dY, int64_t ldy, const void* X, int64_t ldx,
                                   float* partials, float* bias_partials, int64_t T, int N, int K,
                                   int splits, void* stream) {
  if (T == 0) return 0;
  MDLM_CHECK_ARG(dY && X && partials, "gemm_tn_splitk: null pointer");
  MDLM_CHECK_ARG(T > 0 && N > 0 && K > 0 && splits >= 1 && splits <= 4096, "gemm_tn_splitk: bad shape");
  MDLM_CHECK_ARG(N % 256 == 0 && K % 256 == 0, "gemm_tn_splitk: N=%d and K=%d must be multiples of 256", N, K);
  MDLM_CHECK_ARG(T % TN_SLAB == 0 && T / TN_SLAB < (1ll << 30) && splits <= T / TN_SLAB,
                 "gemm_tn_splitk: T=%lld must be a multiple of 64 with at least one slab per split", (long long)T);
  MDLM_CHECK_ARG(ldy >= N && ldx >= K && ldy % 8 == 0 && ldx % 8 == 0 &&
                 (((uintptr_t)dY | (uintptr_t)X | (uintptr_t)partials) & 15) == 0,
                 "gemm_tn_splitk: operands must be 16-byte aligned with leading dimensions %% 8 == 0");
  TnArgs g;
  g.Y = (const bf16_t*)dY; g.ldy = ldy; g.X = (const bf16_t*)X; g.ldx = ldx;
  g.P = partials; g.Pb = bias_partials; g.N = N; g.K = K;
  g.slabs_total = (int)(T / TN_SLAB);
  g.slabs_per_split = (g.slabs_total + splits - 1) / splits;
  // every split must own at least one slab
  MDLM_CHECK_ARG((int64_t)(splits - 1) * g.slabs_per_split < g.slabs_total,
                 "gemm_tn_splitk: %d splits leave an empty range for %d slabs", splits, g.slabs_total);
  g.splits = splits; g.tiles_n = N / 256; g.tiles_k = K / 256;
  const int64_t total = (int64_t)g.tiles_n * g.tiles_k * splits;
  MDLM_CHECK_ARG(total < (1ll << 30), "gemm_tn_splitk: too many tiles");
  constexpr int LDS = 2 * TN_STAGE;   // 128 KiB: one workgroup per CU
  static bool attr_done[MAX_DEVICES] = {};
  int dev = 0;
  if (int rc = ensure_dynamic_lds(&gemm_tn_splitk_kernel, LDS, attr_done, "gemm_tn_splitk", &dev)) return rc;
  int64_t grid = (total + 7) / 8 * 8;
  if (grid > device_cus(dev)) grid = device_cus(dev);
  gemm_tn_splitk_kernel<<<(unsigned)grid, 512, LDS, (hipStream_t)stream>>>(g);
  MDLM_LAUNCH_CHECK();
  return 0;
}
