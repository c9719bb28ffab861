// bf16 MFMA GEMM with fused epilogues for the DiT linears (gfx950).
//   C[M,N] = epilogue(A[M,K] @ W[N,K]^T)     (nn.Linear layout: both K-contiguous)
// Reference call sites: models/dit.py:256 (QKV), 277 (out-proj + gate/residual),
// 225-228 (MLP, GELU-tanh), 306/320 (vocab projection), 112-115 (rotary).
//
// Two tile configurations of one kernel template:
//   256x256x64, 512 threads = 8 waves (2 M x 4 N), wave tile 128x64  (M % 256 == 0)
//   128x128x64, 256 threads = 4 waves (2 M x 2 N), wave tile  64x64  (M % 128 == 0)
// MFMA 16x16x32 bf16, fp32 accumulators.  A and W tiles are staged global->LDS
// with 16-byte global_load_lds (LDS-DMA), double-buffered (2 x 64 KiB for the
// 256^2 tile: half the L2->LDS bytes per flop of the 128^2 tile, which is
// what bounds these K=768 GEMMs).  The LDS image is lane-linear with the
// 16-byte chunk index XOR-swizzled by (row & 7) on the SOURCE address and
// un-swizzled on the ds_read_b128 fragment reads (conflict-free for the
// 16x16x32 operand map).  Epilogues go through LDS (each wave its own 64x64
// fp32 region) so that global stores are row-contiguous 8/16-byte vectors.
// Block ids are remapped so that each XCD (private 4 MiB L2) walks groups of
// M-tiles x all N-tiles: the A panels stay L2-resident while W streams.
#include "common.h"

namespace mdlm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 64;

struct GemmArgs {
  const bf16_t* A; int64_t lda;
  const bf16_t* W;
  const float* bias;
  void* C; int64_t ldc;
  int64_t M; int N; int K;
  const float* gate; const float* residual;
  int64_t rows_per_mod; int n_mod; int64_t mod_ld;
  const int32_t* out_sample_map;
  const float* rot_cos; const float* rot_sin; int n_heads;
  const int32_t* m_rows_dev;
  int tiles_m, tiles_n;
  int vec_ok;  // ldc / alignment allow vector stores
};

#define GLDS16(gptr, lptr) \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr), \
                                   (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

__device__ __forceinline__ float gelu_tanh(float x) {
  // nn.GELU(approximate='tanh'), dit.py:227
  const float y = 0.7978845608028654f * (x + 0.044715f * x * x * x);
  const float e = __expf(2.f * y);
  const float th = 1.f - 2.f / (e + 1.f);
  return 0.5f * x * (1.f + th);
}

__device__ __forceinline__ float bf16_round(float x) { return bf16_to_f32(f32_to_bf16(x)); }

// Epilogue of one 64x64 wave sub-tile held as acc[4][4] (16x16 MFMA tiles):
// grow0/gcol0 = its first global row / column; ep = this wave's 16 KiB LDS
// staging region.  rows_per_mod % 64 == 0 whenever rows are decomposed into
// (sample, position), so a 64-row tile never straddles two samples.
template <int EPI, typename TC, int MT, int HALF>
__device__ __forceinline__ void epilogue_64x64(const f32x4 (&accf)[MT][4], float* ep,
                                               int64_t grow0, int gcol0,
                                               const GemmArgs& g, int lane) {
  const int fq = lane >> 4, fr = lane & 15;
  if (gcol0 >= g.N) return;  // wave tile entirely past the N edge (wave-uniform)
#define acc(m, n) accf[HALF * 4 + (m)][n]
  if constexpr (EPI == MDLM_EPI_QKV_ROTARY) {
    const int cg = gcol0 >> 6;               // (three, head) column group
    const int three = cg / g.n_heads;
    const int h = cg - three * g.n_heads;
    const int64_t L = g.rows_per_mod;
    const int64_t b = grow0 / L;
    const int64_t pos0 = grow0 - b * L;
    bf16_t* const base = reinterpret_cast<bf16_t*>(g.C) + (int64_t)three * g.M * g.n_heads * 64;
    if (three == 2) {
      // v^T[b,h,d,l]: a lane holds 4 consecutive l for one d -> 8-byte stores
      bf16_t* const vb = base + ((b * g.n_heads + h) * 64) * L + pos0;
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          const int d = n * 16 + fr;
          *reinterpret_cast<uint2*>(vb + (int64_t)d * L + m * 16 + fq * 4) =
            make_uint2(pack_bf16x2(acc(m, n)[0], acc(m, n)[1]),
                       pack_bf16x2(acc(m, n)[2], acc(m, n)[3]));
        }
      return;
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
          ep[(m * 16 + fq * 4 + jj) * 64 + n * 16 + fr] = acc(m, n)[jj];
    // (the wave reads back only its own region: no block barrier needed)
    const int c = (lane & 7) * 4;
    bf16_t* const qb = base + ((b * g.n_heads + h) * L + pos0) * 64 + c;
    const float* const cosb = g.rot_cos + pos0 * 32 + c;
    const float* const sinb = g.rot_sin + pos0 * 32 + c;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = it * 8 + (lane >> 3);
      const float4 x1 = *reinterpret_cast<const float4*>(ep + row * 64 + c);
      const float4 x2 = *reinterpret_cast<const float4*>(ep + row * 64 + c + 32);
      const float4 cs = *reinterpret_cast<const float4*>(cosb + row * 32);
      const float4 sn = *reinterpret_cast<const float4*>(sinb + row * 32);
      // Linear output is bf16 under autocast (dit.py:256, 365); the rotary
      // kernel reads bf16, rotates in fp32, writes bf16 (dit.py:112-115, 264)
      const float a0 = bf16_round(x1.x), a1 = bf16_round(x1.y), a2 = bf16_round(x1.z), a3 = bf16_round(x1.w);
      const float b0 = bf16_round(x2.x), b1 = bf16_round(x2.y), b2 = bf16_round(x2.z), b3 = bf16_round(x2.w);
      bf16_t* dst = qb + row * 64;
      *reinterpret_cast<uint2*>(dst) = make_uint2(
        pack_bf16x2(a0 * cs.x - b0 * sn.x, a1 * cs.y - b1 * sn.y),
        pack_bf16x2(a2 * cs.z - b2 * sn.z, a3 * cs.w - b3 * sn.w));
      *reinterpret_cast<uint2*>(dst + 32) = make_uint2(
        pack_bf16x2(a0 * sn.x + b0 * cs.x, a1 * sn.y + b1 * cs.y),
        pack_bf16x2(a2 * sn.z + b2 * cs.z, a3 * sn.w + b3 * cs.w));
    }
  } else {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
          ep[(m * 16 + fq * 4 + jj) * 64 + n * 16 + fr] = acc(m, n)[jj];
    const int c = (lane & 15) * 4;
    const int gc = gcol0 + c;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g.bias) {
      if (gc + 3 < g.N) bias4 = *reinterpret_cast<const float4*>(g.bias + gc);
      else {
        if (gc < g.N) bias4.x = g.bias[gc];
        if (gc + 1 < g.N) bias4.y = g.bias[gc + 1];
        if (gc + 2 < g.N) bias4.z = g.bias[gc + 2];
      }
    }
    // output row of tile row 0 (scatter of compacted samples, STORE only)
    int64_t orow0 = grow0;
    if constexpr (EPI == MDLM_EPI_STORE) {
      if (g.out_sample_map) {
        const int64_t sidx = grow0 / g.rows_per_mod;
        orow0 = (int64_t)g.out_sample_map[sidx] * g.rows_per_mod + (grow0 - sidx * g.rows_per_mod);
      }
    }
    float4 gt = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (EPI == MDLM_EPI_GATE_RESIDUAL) {
      // gate row is constant over the tile (rows_per_mod % 64 == 0)
      const int64_t mrow = g.n_mod == 1 ? 0 : (grow0 / g.rows_per_mod) % g.n_mod;
      gt = *reinterpret_cast<const float4*>(g.gate + mrow * g.mod_ld + gc);
    }
    TC* const Cp = reinterpret_cast<TC*>(g.C) + orow0 * g.ldc + gc;
    const bool fast = g.vec_ok && (gcol0 + 63 < g.N);  // wave-uniform
    if (fast) {
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int row = it * 4 + (lane >> 4);
        float4 v = *reinterpret_cast<const float4*>(ep + row * 64 + c);
        v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
        if constexpr (EPI == MDLM_EPI_BIAS_GELU) {
          v.x = gelu_tanh(v.x); v.y = gelu_tanh(v.y); v.z = gelu_tanh(v.z); v.w = gelu_tanh(v.w);
        }
        if constexpr (EPI == MDLM_EPI_GATE_RESIDUAL) {
          // residual + gate * x (bias_dropout_add_scale, eval), dit.py:28-34
          const float4 rs = *reinterpret_cast<const float4*>(g.residual + (grow0 + row) * g.ldc + gc);
          v.x = rs.x + gt.x * v.x; v.y = rs.y + gt.y * v.y;
          v.z = rs.z + gt.z * v.z; v.w = rs.w + gt.w * v.w;
        }
        TC* dst = Cp + (int64_t)row * g.ldc;
        if constexpr (sizeof(TC) == 4) {
          *reinterpret_cast<float4*>(dst) = v;
        } else {
          *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
        }
      }
    } else {
      // N edge / unaligned ldc: element-wise masked stores
      for (int it = 0; it < 16; ++it) {
        const int row = it * 4 + (lane >> 4);
        float4 v = *reinterpret_cast<const float4*>(ep + row * 64 + c);
        float vv[4] = {v.x + bias4.x, v.y + bias4.y, v.z + bias4.z, v.w + bias4.w};
        TC* dst = Cp + (int64_t)row * g.ldc;
        for (int e = 0; e < 4; ++e) {
          if (gc + e < g.N) {
            float o = vv[e];
            if constexpr (EPI == MDLM_EPI_BIAS_GELU) o = gelu_tanh(o);
            if constexpr (EPI == MDLM_EPI_GATE_RESIDUAL) {
              const float gte = e == 0 ? gt.x : e == 1 ? gt.y : e == 2 ? gt.z : gt.w;
              o = g.residual[(grow0 + row) * g.ldc + gc + e] + gte * o;
            }
            elem<TC>::store(dst + e, o);
          }
        }
      }
    }
  }
}
#undef acc

// WM x WN waves; each wave MT x 4 MFMA tiles (MT*16 rows x 64 columns).
template <int EPI, typename TC, int WM, int WN, int MT>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN) / 4)
void gemm_bf16_kernel(GemmArgs g) {
  constexpr int NW = WM * WN;
  constexpr int TBM = WM * MT * 16, TBN = WN * 64;
  constexpr int A_BYTES = TBM * BK * 2, B_BYTES = TBN * BK * 2;
  constexpr int A_CH = (TBM / 8) / NW, B_CH = (TBN / 8) / NW;  // 1 KiB chunks per wave
  constexpr int GROUP_M = 1024 / TBM;  // 1024 rows of A (1.5 MiB at K=768) per group
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // ---- XCD-aware tile mapping ------------------------------------------
  const int bid = blockIdx.x;
  const int xcd = bid & 7;
  const int j = bid >> 3;
  const int per_group = GROUP_M * g.tiles_n;
  const int gl = j / per_group;
  const int rem = j - gl * per_group;
  const int tn = rem / GROUP_M;
  const int mi = rem - tn * GROUP_M;
  const int tm = (gl * 8 + xcd) * GROUP_M + mi;
  if (tm >= g.tiles_m) return;
  const int64_t m0 = (int64_t)tm * TBM;
  if (g.m_rows_dev && m0 >= (int64_t)*g.m_rows_dev) return;
  const int n0 = tn * TBN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid / WN, wc = wid - wr * WN;

  // ---- staging addresses -------------------------------------------------
  // chunk c = i*NW + wid covers tile rows c*8 .. c*8+7; lane -> (row lane>>3,
  // physical 16-B chunk lane&7) holding logical chunk (lane&7)^(lane>>3).
  const int srow = lane >> 3;
  const int skc = ((lane & 7) ^ srow) * 8;  // element offset inside the 64-wide K slab
  const bf16_t* a_src[A_CH];
  const bf16_t* b_src[B_CH];
#pragma unroll
  for (int i = 0; i < A_CH; ++i)
    a_src[i] = g.A + (m0 + (i * NW + wid) * 8 + srow) * g.lda + skc;
#pragma unroll
  for (int i = 0; i < B_CH; ++i) {
    int brow = n0 + (i * NW + wid) * 8 + srow;
    brow = brow < g.N ? brow : g.N - 1;  // clamp the N edge (stores are masked)
    b_src[i] = g.W + (int64_t)brow * g.K + skc;
  }
  char* const ldsA = smem;                 // 2 x A_BYTES
  char* const ldsB = smem + 2 * A_BYTES;   // 2 x B_BYTES

  auto stage = [&](int buf, int k0) {
#pragma unroll
    for (int i = 0; i < A_CH; ++i)
      GLDS16(a_src[i] + k0, ldsA + buf * A_BYTES + (i * NW + wid) * 1024);
#pragma unroll
    for (int i = 0; i < B_CH; ++i)
      GLDS16(b_src[i] + k0, ldsB + buf * B_BYTES + (i * NW + wid) * 1024);
  };

  // ---- fragment read offsets ---------------------------------------------
  // frag (sub-tile m, k-step ks): row = base + m*16 + (lane&15), logical chunk
  // = ks*4 + (lane>>4), physical = logical ^ (row&7) = logical ^ (lane&7)
  int foff[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
    foff[ks] = (lane & 15) * 128 + (((ks * 4 + (lane >> 4)) ^ (lane & 7)) * 16);
  const int a_base = wr * (MT * 16) * 128;
  const int b_base = wc * 64 * 128;

  f32x4 acc[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int KT = g.K / BK;
  stage(0, 0);
  __syncthreads();  // hipcc drains the LDS-DMA (vmcnt(0)) before the barrier
  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < KT) stage(cur ^ 1, (kt + 1) * BK);
    const char* At = ldsA + cur * A_BYTES + a_base;
    const char* Bt = ldsB + cur * B_BYTES + b_base;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[MT], bfr[4];
#pragma unroll
      for (int n = 0; n < 4; ++n)
        bfr[n] = *reinterpret_cast<const bf16x8*>(Bt + n * 2048 + foff[ks]);
#pragma unroll
      for (int m = 0; m < MT; ++m)
        af[m] = *reinterpret_cast<const bf16x8*>(At + m * 2048 + foff[ks]);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], bfr[n], acc[m][n], 0, 0, 0);
    }
    __syncthreads();
  }

  // ---- epilogue: MT/4 passes of a 64x64 sub-tile through this wave's LDS ----
  float* ep = reinterpret_cast<float*>(smem) + wid * 4096;
  const int gcol0 = n0 + wc * 64;
  const int64_t growW = m0 + wr * (MT * 16);
  epilogue_64x64<EPI, TC, MT, 0>(acc, ep, growW, gcol0, g, lane);
  if constexpr (MT == 8) epilogue_64x64<EPI, TC, MT, 1>(acc, ep, growW + 64, gcol0, g, lane);
}

template <int EPI, typename TC, int WM, int WN, int MT>
static int launch_cfg(GemmArgs g, hipStream_t s) {
  constexpr int TBM = WM * MT * 16, TBN = WN * 64;
  constexpr int LDS = 2 * (TBM + TBN) * BK * 2;
  constexpr int GROUP_M = 1024 / TBM;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(
      reinterpret_cast<const void*>(&gemm_bf16_kernel<EPI, TC, WM, WN, MT>),
      hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) { set_error("gemm: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    attr_done = true;
  }
  g.tiles_m = (int)(g.M / TBM);
  g.tiles_n = (g.N + TBN - 1) / TBN;
  const int groups = (g.tiles_m + GROUP_M - 1) / GROUP_M;
  const int gl = (groups + 7) / 8;
  const int64_t grid = (int64_t)8 * gl * GROUP_M * g.tiles_n;
  if (grid >= (1ll << 31)) { set_error("gemm: grid too large"); return -1; }
  gemm_bf16_kernel<EPI, TC, WM, WN, MT><<<(unsigned)grid, 64 * WM * WN, LDS, s>>>(g);
  MDLM_LAUNCH_CHECK();
  return 0;
}

static int g_force_small_tile = 0;  // test hook (mdlm_gemm_set_tile)

template <int EPI, typename TC>
static int launch_gemm(const GemmArgs& g, hipStream_t s) {
  if (g.M % 256 == 0 && g.N >= 256 && !g_force_small_tile)
    return launch_cfg<EPI, TC, 2, 4, 8>(g, s);
  return launch_cfg<EPI, TC, 2, 2, 4>(g, s);
}

}  // namespace mdlm

using namespace mdlm;

extern "C" int mdlm_gemm_set_tile(int force_small) {
  g_force_small_tile = force_small;
  return 0;
}

extern "C" int mdlm_gemm_bf16(const void* A, int64_t lda, const void* W,
                              const float* bias, void* C, int64_t ldc,
                              int dtype_c, int64_t M, int N, int K, int epilogue,
                              const float* gate, const float* residual,
                              int64_t rows_per_mod, int n_mod, int64_t mod_ld,
                              const float* rot_cos, const float* rot_sin,
                              int n_heads, const int32_t* m_rows_dev,
                              const int32_t* out_sample_map, void* stream) {
  MDLM_CHECK_ARG(A && W && C, "gemm_bf16: null pointer");
  MDLM_CHECK_ARG(M >= 0 && N > 0 && K > 0, "gemm_bf16: bad shape");
  MDLM_CHECK_ARG(M % 128 == 0, "gemm_bf16: M=%lld must be a multiple of 128", (long long)M);
  MDLM_CHECK_ARG(K % BK == 0, "gemm_bf16: K=%d must be a multiple of %d", K, BK);
  MDLM_CHECK_ARG(lda >= K && lda % 8 == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0,
                 "gemm_bf16: A/W must be 16-byte aligned with lda %% 8 == 0");
  MDLM_CHECK_ARG(M * (N > K ? N : K) < (1ll << 40), "gemm_bf16: too large");
  if (M == 0) return 0;
  GemmArgs g;
  g.A = (const bf16_t*)A; g.lda = lda; g.W = (const bf16_t*)W; g.bias = bias;
  g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.gate = gate; g.residual = residual;
  g.rows_per_mod = rows_per_mod > 0 ? rows_per_mod : 1; g.n_mod = n_mod > 0 ? n_mod : 1;
  g.mod_ld = mod_ld; g.out_sample_map = out_sample_map;
  g.rot_cos = rot_cos; g.rot_sin = rot_sin; g.n_heads = n_heads; g.m_rows_dev = m_rows_dev;
  g.tiles_m = 0; g.tiles_n = 0;
  hipStream_t s = (hipStream_t)stream;
  g.vec_ok = (((uintptr_t)C & 15) == 0) && (ldc % 4 == 0) ? 1 : 0;
  switch (epilogue) {
    case MDLM_EPI_STORE:
      MDLM_CHECK_ARG(ldc >= N, "gemm_bf16: ldc < N");
      MDLM_CHECK_ARG(dtype_c == MDLM_F32 || dtype_c == MDLM_BF16, "gemm_bf16: bad dtype_c");
      MDLM_CHECK_ARG(!out_sample_map || g.rows_per_mod % 64 == 0, "gemm_bf16: out_sample_map needs rows_per_mod %% 64 == 0");
      if (dtype_c == MDLM_F32) return launch_gemm<MDLM_EPI_STORE, float>(g, s);
      return launch_gemm<MDLM_EPI_STORE, bf16_t>(g, s);
    case MDLM_EPI_BIAS_GELU:
      MDLM_CHECK_ARG(ldc >= N && dtype_c == MDLM_BF16, "gemm_bf16: BIAS_GELU writes bf16 with ldc >= N");
      return launch_gemm<MDLM_EPI_BIAS_GELU, bf16_t>(g, s);
    case MDLM_EPI_GATE_RESIDUAL:
      MDLM_CHECK_ARG(gate && residual && dtype_c == MDLM_F32, "gemm_bf16: GATE_RESIDUAL needs gate, residual and fp32 C");
      MDLM_CHECK_ARG(ldc >= N && N % 4 == 0 && ldc % 4 == 0 && ((uintptr_t)residual & 15) == 0 && ((uintptr_t)gate & 15) == 0 && g.vec_ok,
                     "gemm_bf16: GATE_RESIDUAL needs N, ldc multiples of 4 and 16-byte aligned buffers");
      MDLM_CHECK_ARG(mod_ld % 4 == 0 && (g.n_mod == 1 || (mod_ld >= N && g.rows_per_mod % 64 == 0)),
                     "gemm_bf16: bad gate leading dimension / rows_per_mod %% 64 != 0");
      return launch_gemm<MDLM_EPI_GATE_RESIDUAL, float>(g, s);
    case MDLM_EPI_QKV_ROTARY:
      MDLM_CHECK_ARG(rot_cos && rot_sin && n_heads > 0 && N == 3 * n_heads * 64,
                     "gemm_bf16: QKV_ROTARY needs head_dim 64 (N == 3*H*64) and rotary tables");
      MDLM_CHECK_ARG(rows_per_mod > 0 && rows_per_mod % 64 == 0 && M % rows_per_mod == 0,
                     "gemm_bf16: QKV_ROTARY needs M %% L == 0 and L %% 64 == 0");
      MDLM_CHECK_ARG(dtype_c == MDLM_BF16, "gemm_bf16: QKV_ROTARY writes bf16");
      return launch_gemm<MDLM_EPI_QKV_ROTARY, bf16_t>(g, s);
    default:
      MDLM_CHECK_ARG(false, "gemm_bf16: unknown epilogue %d", epilogue);
  }
  return -1;
}
