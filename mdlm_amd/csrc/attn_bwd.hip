// Bidirectional attention backward for the DiT block (gfx950), head_dim 64:
// the gradient of flash_attn_varlen_qkvpacked_func(qkv, ..., causal=False) at
// models/dit.py:265-275, which the reference gets from flash-attn's autograd.
//
// Inputs: q, k head-major (n_seq,H,L,64), vt (n_seq,H,64,L) as written by the
// QKV GEMM epilogue; o, do (n_seq*L, H*64) row-major; lse (n_seq,H,L) = log2-
// domain log-sum-exp of the scaled scores from mdlm_attn_fwd_lse.
// Outputs: dq, dk, dv head-major (n_seq,H,L,64) bf16.
//
// With P = exp2(S*sc - lse), S = Q K^T, sc = log2(e)/8, delta = rowsum(dO*O):
//   dV = P^T dO,  dP = dO V^T,  dS = P * (dP - delta),
//   dQ = dS K / 8,  dK = dS^T Q / 8.
// Two kernels, no atomics (bitwise reproducible): one workgroup per 64-key
// block accumulates dK/dV over the query blocks, one per 64-query block
// accumulates dQ over the key blocks; P and dS are recomputed from the saved
// lse.  Every matmul is v_mfma_f32_32x32x16_bf16 on 64x64 bf16 tiles staged
// row-major in LDS as they lie in HBM; an operand whose contraction index is
// the tile's ROW index (Q^T, dO^T, K^T, and V, which arrives as v^T) is
// gathered with the transposing LDS read ds_read_b64_tr_b16 instead of being
// staged a second time transposed (the scalar transposed stores of the first
// version were 16-way bank-conflicted).  P / dS go through LDS once as bf16.
// Four waves, one 32x32 output tile each.
#include "common.h"

namespace mdlm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BT = 64;            // tile edge
constexpr int LDP = 72;           // LDS row stride in elements (144 B: 16-byte aligned, de-phased banks)
constexpr int TILE_E = BT * LDP;  // elements per LDS tile

// stage a 64x64 bf16 tile (row stride ld elements) into LDS row-major; 256 threads
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ src, int64_t ld,
                                           bf16_t* dst, int tid) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int id = tid + i * 256;
    const int row = id >> 3, ch = id & 7;
    *reinterpret_cast<uint4*>(dst + row * LDP + ch * 8) =
      *reinterpret_cast<const uint4*>(src + (int64_t)row * ld + ch * 8);
  }
}

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// Operand fragment of v_mfma_f32_32x32x16_bf16 (lane: row r = lane & 31, k half
// hh = lane >> 5 -> contraction elements ks*16 + hh*8 .. +7):
//   frag_row: the tile holds [row][contraction]  (one 16-byte read)
//   frag_tr : the tile holds [contraction][row]  (two transposing reads; the
//             16-lane group (lane >> 4) covers rows 16*(group & 1) .. +15, lane
//             4q+p of it supplies the address of contraction row q, columns 4p..4p+3)
__device__ __forceinline__ bf16x8 frag_row(const bf16_t* T, int row0, int ks, int r, int hh) {
  return *reinterpret_cast<const bf16x8*>(T + (row0 + r) * LDP + ks * 16 + hh * 8);
}
__device__ __forceinline__ bf16x8 frag_tr(const bf16_t* T, int row0, int ks, int lane) {
  const int hh = lane >> 5, half = (lane >> 4) & 1, q = (lane & 15) >> 2, p = lane & 3;
  const bf16_t* a = T + (ks * 16 + hh * 8 + q) * LDP + row0 + half * 16 + p * 4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a + 4 * LDP));
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

// acc[i][j] += sum_k A[i][k] * B[j][k], 32x32 tile, k < 64.  TA / TB: the operand's
// tile is stored [contraction][row] and read transposed; a0 / b0 = first row of
// the 32-row operand block.  acc[e]: i = 8*(e>>2) + 4*hh + (e&3), j = r.
template <bool TA, bool TB>
__device__ __forceinline__ void mma64(f32x16& acc, const bf16_t* A, int a0, const bf16_t* B, int b0, int lane) {
  const int r = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const bf16x8 a = TA ? frag_tr(A, a0, ks, lane) : frag_row(A, a0, ks, r, hh);
    const bf16x8 b = TB ? frag_tr(B, b0, ks, lane) : frag_row(B, b0, ks, r, hh);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  }
}

__device__ __forceinline__ int acc_row(int e, int hh) { return 8 * (e >> 2) + 4 * hh + (e & 3); }

// delta[bh*L + l] = sum_d dO * O
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout,
                                                         float* __restrict__ delta, int64_t M, int H, int L) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // (m, h, chunk)
  const int ch = (int)(i & 7);
  const int64_t mh = i >> 3;
  float s = 0.f;
  int64_t m = 0; int h = 0;
  if (mh < M * H) {
    h = (int)(mh % H); m = mh / H;
    const int64_t off = m * ((int64_t)H * 64) + h * 64 + ch * 8;
    const uint4 a = *reinterpret_cast<const uint4*>(o + off);
    const uint4 b = *reinterpret_cast<const uint4*>(dout + off);
    const unsigned aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      s += __uint_as_float(aw[k] << 16) * __uint_as_float(bw[k] << 16) +
           __uint_as_float(aw[k] & 0xffff0000u) * __uint_as_float(bw[k] & 0xffff0000u);
  }
  s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
  if (mh < M * H && ch == 0) {
    const int64_t seq = m / L;
    delta[(seq * H + h) * L + (m - seq * L)] = s;
  }
}

__global__ __launch_bounds__(256) void attn_bwd_dkdv_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ vt,
    const bf16_t* __restrict__ dout, const float* __restrict__ lse, const float* __restrict__ delta,
    bf16_t* __restrict__ dk, bf16_t* __restrict__ dv, int H, int L) {
  __shared__ __attribute__((aligned(16))) bf16_t sm[6 * TILE_E];
  bf16_t* Ks = sm;                 // [key][d]
  bf16_t* VTs = sm + TILE_E;       // [d][key]  (v^T as stored)
  bf16_t* Qs = sm + 2 * TILE_E;    // [q][d]
  bf16_t* dOs = sm + 3 * TILE_E;   // [q][d]
  bf16_t* PTs = sm + 4 * TILE_E;   // P^T  [key][q]
  bf16_t* dSTs = sm + 5 * TILE_E;  // dS^T [key][q]
  const int kblocks = L / BT;
  const int kb = blockIdx.x % kblocks;
  const int bh = blockIdx.x / kblocks;
  const int seq = bh / H, head = bh - seq * H;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int wa = wid >> 1, wb = wid & 1;  // tile row (key) / tile column (q or d)
  const int key0 = kb * BT;
  const int64_t D = (int64_t)H * 64;
  stage_tile(k + ((int64_t)bh * L + key0) * 64, 64, Ks, tid);
  stage_tile(vt + (int64_t)bh * 64 * L + key0, L, VTs, tid);
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  f32x16 dk_acc = zero16, dv_acc = zero16;
  const float sc = 0.125f * 1.4426950408889634f;
  for (int qb = 0; qb < L / BT; ++qb) {
    const int q0 = qb * BT;
    stage_tile(q + ((int64_t)bh * L + q0) * 64, 64, Qs, tid);
    stage_tile(dout + ((int64_t)seq * L + q0) * D + head * 64, D, dOs, tid);
    const float lse_q = lse[(int64_t)bh * L + q0 + wb * 32 + r];
    const float del_q = delta[(int64_t)bh * L + q0 + wb * 32 + r];
    __syncthreads();
    f32x16 st = zero16, dpt = zero16;   // S^T, dP^T tiles [key (rows), q (cols)]
    mma64<false, false>(st, Ks, wa * 32, Qs, wb * 32, lane);
    mma64<true, false>(dpt, VTs, wa * 32, dOs, wb * 32, lane);   // V[key][d] read out of v^T
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float p = __builtin_amdgcn_exp2f(st[e] * sc - lse_q);
      const float ds = p * (dpt[e] - del_q);
      const int key = wa * 32 + acc_row(e, hh);
      PTs[key * LDP + wb * 32 + r] = f32_to_bf16(p);
      dSTs[key * LDP + wb * 32 + r] = f32_to_bf16(ds);
    }
    __syncthreads();
    // dV[key, d] += P^T[key, :] . dO[:, d];  dK[key, d] += dS^T[key, :] . Q[:, d]
    // (dO, Q: contraction over their row index q -> transposing reads)
    mma64<false, true>(dv_acc, PTs, wa * 32, dOs, wb * 32, lane);
    mma64<false, true>(dk_acc, dSTs, wa * 32, Qs, wb * 32, lane);
    __syncthreads();
  }
  bf16_t* dkp = dk + ((int64_t)bh * L + key0 + wa * 32) * 64 + wb * 32 + r;
  bf16_t* dvp = dv + ((int64_t)bh * L + key0 + wa * 32) * 64 + wb * 32 + r;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int row = acc_row(e, hh);
    dkp[row * 64] = f32_to_bf16(dk_acc[e] * 0.125f);
    dvp[row * 64] = f32_to_bf16(dv_acc[e]);
  }
}

__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ vt,
    const bf16_t* __restrict__ dout, const float* __restrict__ lse, const float* __restrict__ delta,
    bf16_t* __restrict__ dq, int H, int L) {
  __shared__ __attribute__((aligned(16))) bf16_t sm[5 * TILE_E];
  __shared__ float lse_s[BT], del_s[BT];
  bf16_t* Qs = sm;                // [q][d]
  bf16_t* dOs = sm + TILE_E;      // [q][d]
  bf16_t* Ks = sm + 2 * TILE_E;   // [key][d]
  bf16_t* VTs = sm + 3 * TILE_E;  // [d][key]  (v^T as stored)
  bf16_t* dSs = sm + 4 * TILE_E;  // [q][key]
  const int qblocks = L / BT;
  const int qb = blockIdx.x % qblocks;
  const int bh = blockIdx.x / qblocks;
  const int seq = bh / H, head = bh - seq * H;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int wa = wid >> 1, wb = wid & 1;  // tile row (q) / tile column (key or d)
  const int q0 = qb * BT;
  const int64_t D = (int64_t)H * 64;
  stage_tile(q + ((int64_t)bh * L + q0) * 64, 64, Qs, tid);
  stage_tile(dout + ((int64_t)seq * L + q0) * D + head * 64, D, dOs, tid);
  if (tid < BT) {
    lse_s[tid] = lse[(int64_t)bh * L + q0 + tid];
    del_s[tid] = delta[(int64_t)bh * L + q0 + tid];
  }
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  f32x16 dq_acc = zero16;
  const float sc = 0.125f * 1.4426950408889634f;
  for (int kb = 0; kb < L / BT; ++kb) {
    const int key0 = kb * BT;
    stage_tile(k + ((int64_t)bh * L + key0) * 64, 64, Ks, tid);
    stage_tile(vt + (int64_t)bh * 64 * L + key0, L, VTs, tid);
    __syncthreads();
    f32x16 s = zero16, dp = zero16;   // S, dP tiles [q (rows), key (cols)]
    mma64<false, false>(s, Qs, wa * 32, Ks, wb * 32, lane);
    mma64<false, true>(dp, dOs, wa * 32, VTs, wb * 32, lane);   // V[key][d] read out of v^T
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int qi = wa * 32 + acc_row(e, hh);
      const float p = __builtin_amdgcn_exp2f(s[e] * sc - lse_s[qi]);
      dSs[qi * LDP + wb * 32 + r] = f32_to_bf16(p * (dp[e] - del_s[qi]));
    }
    __syncthreads();
    // dQ[q, d] += dS[q, :] . K[:, d]  (K: contraction over its row index key)
    mma64<false, true>(dq_acc, dSs, wa * 32, Ks, wb * 32, lane);
    __syncthreads();
  }
  bf16_t* dqp = dq + ((int64_t)bh * L + q0 + wa * 32) * 64 + wb * 32 + r;
#pragma unroll
  for (int e = 0; e < 16; ++e) dqp[acc_row(e, hh) * 64] = f32_to_bf16(dq_acc[e] * 0.125f);
}

// ---------------------------------------------------------------------------
// Short sequences (L == 128, the LM1B recipe): ONE workgroup per (sequence,
// head) holds Q, K, dO, v^T and the whole 128x128 P / dS in LDS (139 KiB), so
// S and dP are computed once (5 matmuls instead of the 7 of the two-kernel
// scheme), nothing is staged twice and delta = rowsum(dO * O) is formed in
// the same kernel.  8 waves: two 32x32 tiles of S / dP each, then one 32x32
// tile each of dV, dK and dQ.
constexpr int LQ = 128;
constexpr int LDW = 136;   // row stride of the 128-wide tiles (272 B: 16-byte aligned, de-phased banks)

template <int LD>
__device__ __forceinline__ bf16x8 fragr(const bf16_t* T, int row0, int ks, int r, int hh) {
  return *reinterpret_cast<const bf16x8*>(T + (row0 + r) * LD + ks * 16 + hh * 8);
}
template <int LD>
__device__ __forceinline__ bf16x8 fragt(const bf16_t* T, int col0, int ks, int lane) {
  const int hh = lane >> 5, half = (lane >> 4) & 1, q = (lane & 15) >> 2, p = lane & 3;
  const bf16_t* a = T + (ks * 16 + hh * 8 + q) * LD + col0 + half * 16 + p * 4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a + 4 * LD));
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}
// acc[i][j] += sum_{k < 16*KS} A[i][k] B[j][k]; TA / TB: operand stored [contraction][row]
template <bool TA, bool TB, int LDA, int LDB, int KS>
__device__ __forceinline__ void mmak(f32x16& acc, const bf16_t* A, int a0, const bf16_t* B, int b0, int lane) {
  const int r = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    bf16x8 a, b;
    if constexpr (TA) a = fragt<LDA>(A, a0, ks, lane); else a = fragr<LDA>(A, a0, ks, r, hh);
    if constexpr (TB) b = fragt<LDB>(B, b0, ks, lane); else b = fragr<LDB>(B, b0, ks, r, hh);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  }
}

__global__ __launch_bounds__(512) void attn_bwd_l128_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ vt,
    const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout, const float* __restrict__ lse,
    bf16_t* __restrict__ dq, bf16_t* __restrict__ dk, bf16_t* __restrict__ dv, int H) {
  extern __shared__ __attribute__((aligned(16))) char l128_smem[];
  bf16_t* Qs = reinterpret_cast<bf16_t*>(l128_smem);   // [128 q][LDP]
  bf16_t* Ks = Qs + LQ * LDP;                          // [128 key][LDP]
  bf16_t* dOs = Ks + LQ * LDP;                         // [128 q][LDP]
  bf16_t* VTs = dOs + LQ * LDP;                        // [64 d][LDW]   (v^T as stored)
  bf16_t* Ps = VTs + 64 * LDW;                         // [128 q][LDW]
  bf16_t* dSs = Ps + LQ * LDW;                         // [128 q][LDW]
  float* lse_s = reinterpret_cast<float*>(dSs + LQ * LDW);
  float* del_s = lse_s + LQ;
  const int bh = blockIdx.x;
  const int seq = bh / H, head = bh - seq * H;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int64_t D = (int64_t)H * 64;
  // ---- stage (16-byte chunks): Q, K, dO: 128 rows x 8 chunks; v^T: 64 rows x 16 chunks
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int id = tid + i * 512;
    const int row = id >> 3, ch = id & 7;
    *reinterpret_cast<uint4*>(Qs + row * LDP + ch * 8) =
      *reinterpret_cast<const uint4*>(q + ((int64_t)bh * LQ + row) * 64 + ch * 8);
    *reinterpret_cast<uint4*>(Ks + row * LDP + ch * 8) =
      *reinterpret_cast<const uint4*>(k + ((int64_t)bh * LQ + row) * 64 + ch * 8);
    *reinterpret_cast<uint4*>(dOs + row * LDP + ch * 8) =
      *reinterpret_cast<const uint4*>(dout + ((int64_t)seq * LQ + row) * D + head * 64 + ch * 8);
    const int vrow = id >> 4, vch = id & 15;
    *reinterpret_cast<uint4*>(VTs + vrow * LDW + vch * 8) =
      *reinterpret_cast<const uint4*>(vt + ((int64_t)bh * 64 + vrow) * LQ + vch * 8);
  }
  // ---- delta[q] = sum_d dO * O (four lanes per row, 16 columns each), lse
  {
    const int row = tid >> 2, c0 = (tid & 3) * 16;
    const int64_t off = ((int64_t)seq * LQ + row) * D + head * 64 + c0;
    float sacc = 0.f;
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      const uint4 a = *reinterpret_cast<const uint4*>(o + off + h2 * 8);
      const uint4 b = *reinterpret_cast<const uint4*>(dout + off + h2 * 8);
      const unsigned aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        sacc += __uint_as_float(aw[e] << 16) * __uint_as_float(bw[e] << 16) +
                __uint_as_float(aw[e] & 0xffff0000u) * __uint_as_float(bw[e] & 0xffff0000u);
    }
    sacc += __shfl_xor(sacc, 1); sacc += __shfl_xor(sacc, 2);
    if ((tid & 3) == 0) { del_s[row] = sacc; lse_s[row] = lse[(int64_t)bh * LQ + row]; }
  }
  __syncthreads();
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const float sc = 0.125f * 1.4426950408889634f;
  // ---- phase 1: S = Q K^T, dP = dO V^T  ->  P, dS (bf16) for the whole 128 x 128
#pragma unroll
  for (int t2 = 0; t2 < 2; ++t2) {
    const int t = wid * 2 + t2, ti = t >> 2, tj = t & 3;   // tile (q block, key block)
    f32x16 s = zero16, dp = zero16;
    mmak<false, false, LDP, LDP, 4>(s, Qs, ti * 32, Ks, tj * 32, lane);
    mmak<false, true, LDP, LDW, 4>(dp, dOs, ti * 32, VTs, tj * 32, lane);   // V[key][d] out of v^T
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int qi = ti * 32 + acc_row(e, hh);
      const float p = __builtin_amdgcn_exp2f(s[e] * sc - lse_s[qi]);
      Ps[qi * LDW + tj * 32 + r] = f32_to_bf16(p);
      dSs[qi * LDW + tj * 32 + r] = f32_to_bf16(p * (dp[e] - del_s[qi]));
    }
  }
  __syncthreads();
  // ---- phase 2: one 32 x 32 tile each of dV, dK (rows = keys) and dQ (rows = queries)
  const int ta = wid >> 1, tb = wid & 1;
  f32x16 dv_acc = zero16, dk_acc = zero16, dq_acc = zero16;
  mmak<true, true, LDW, LDP, 8>(dv_acc, Ps, ta * 32, dOs, tb * 32, lane);    // sum_q P[q][key] dO[q][d]
  mmak<true, true, LDW, LDP, 8>(dk_acc, dSs, ta * 32, Qs, tb * 32, lane);    // sum_q dS[q][key] Q[q][d]
  mmak<false, true, LDW, LDP, 8>(dq_acc, dSs, ta * 32, Ks, tb * 32, lane);   // sum_key dS[q][key] K[key][d]
  const int64_t obase = ((int64_t)bh * LQ + ta * 32) * 64 + tb * 32 + r;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int row = acc_row(e, hh);
    dv[obase + row * 64] = f32_to_bf16(dv_acc[e]);
    dk[obase + row * 64] = f32_to_bf16(dk_acc[e] * 0.125f);
    dq[obase + row * 64] = f32_to_bf16(dq_acc[e] * 0.125f);
  }
}

}  // namespace mdlm

using namespace mdlm;

extern "C" int mdlm_attn_bwd(const void* q, const void* k, const void* vt, const void* o,
                             const void* dout, const float* lse, float* delta_ws,
                             void* dq, void* dk, void* dv, int n_seq, int H, int L,
                             void* stream) {
  if (n_seq == 0) return 0;
  MDLM_CHECK_ARG(q && k && vt && o && dout && lse && delta_ws && dq && dk && dv, "attn_bwd: null pointer");
  MDLM_CHECK_ARG(n_seq > 0 && H > 0 && L > 0 && L % BT == 0, "attn_bwd: L=%d must be a multiple of %d", L, BT);
  MDLM_CHECK_ARG((((uintptr_t)q | (uintptr_t)k | (uintptr_t)vt | (uintptr_t)o | (uintptr_t)dout | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 15) == 0,
                 "attn_bwd: buffers must be 16-byte aligned");
  const int64_t M = (int64_t)n_seq * L;
  const int64_t blocks = (int64_t)n_seq * H * (L / BT);
  MDLM_CHECK_ARG(blocks < (1ll << 31) && M * H * 8 / 256 < (1ll << 31), "attn_bwd: too many blocks");
  hipStream_t s = (hipStream_t)stream;
  if (L == LQ) {
    constexpr int LDS = (3 * LQ * LDP + 64 * LDW + 2 * LQ * LDW) * 2 + 2 * LQ * 4;
    static bool attr_done[MAX_DEVICES] = {};
    if (int rc = ensure_dynamic_lds(&attn_bwd_l128_kernel, LDS, attr_done, "attn_bwd")) return rc;
    attn_bwd_l128_kernel<<<(unsigned)((int64_t)n_seq * H), 512, LDS, s>>>(
      (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)vt, (const bf16_t*)o, (const bf16_t*)dout, lse,
      (bf16_t*)dq, (bf16_t*)dk, (bf16_t*)dv, H);
    MDLM_LAUNCH_CHECK();
    return 0;
  }
  attn_delta_kernel<<<(unsigned)((M * H * 8 + 255) / 256), 256, 0, s>>>((const bf16_t*)o, (const bf16_t*)dout, delta_ws, M, H, L);
  attn_bwd_dkdv_kernel<<<(unsigned)blocks, 256, 0, s>>>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)vt, (const bf16_t*)dout,
                                                        lse, delta_ws, (bf16_t*)dk, (bf16_t*)dv, H, L);
  attn_bwd_dq_kernel<<<(unsigned)blocks, 256, 0, s>>>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)vt, (const bf16_t*)dout,
                                                      lse, delta_ws, (bf16_t*)dq, H, L);
  MDLM_LAUNCH_CHECK();
  return 0;
}
