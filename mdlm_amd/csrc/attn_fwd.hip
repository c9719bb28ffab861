// Bidirectional flash attention forward for the DiT block (gfx950), head_dim 64.
// Replaces flash_attn_varlen_qkvpacked_func(qkv, cu_seqlens, L, 0.0,
// causal=False) at models/dit.py:265-275 (equal-length sequences, softmax
// scale 1/sqrt(64), no mask, no dropout).
//
// Inputs come from the QKV GEMM epilogue: q,k head-major (n_seq,H,L,64) and v
// already transposed vt (n_seq,H,64,L).  One workgroup = 4 waves = 128 query
// rows of one (sequence, head); each wave owns 32 query rows.  K / V^T tiles of
// 64 keys are staged global->LDS by 16-byte LDS-DMA, double buffered, with the
// 16-byte chunk index XOR-swizzled by (row>>1)&7 (conflict-free ds_read_b128
// for the 32x32x16 A-operand map).
//
// MFMA plan (v_mfma_f32_32x32x16_bf16):
//   S^T[key,q] = K[key,:] . Q[q,:]      (A = K rows, B = Q rows)  -> a lane holds
//                one query column and 16 keys per 32-key tile, so the softmax
//                row statistics are in-lane + one cross-half exchange;
//   O^T[d,q]  += V^T[d,key] . P^T[key,q] (A = V^T rows, B = the S^T accumulator
//                re-packed to bf16 in place: no LDS round trip).  The K rows fed
//                to A-row r are permuted (bits 2,3 of r swapped) so that the keys
//                each lane needs from V^T are 8 contiguous elements.
#include "common.h"
#include <type_traits>

namespace mdlm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int AT = 256;          // threads: 4 waves x 32 query rows
constexpr int QB = 128;          // query rows per workgroup
constexpr int KB = 64;           // keys per tile
constexpr int HD = 64;           // head dim
constexpr int KV_TILE = KB * HD * 2;  // 8 KiB

#define GLDS16(gptr, lptr) \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr), \
                                   (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

// QT = 32-row query tiles per wave: 2 when L % 256 == 0 (K / V^T fragments
// are then read once for 64 query rows), else 1.
template <int QT>
__global__ __launch_bounds__(AT, 2) void attn_fwd_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
    const bf16_t* __restrict__ vt, bf16_t* __restrict__ out, int H, int L,
    const int32_t* __restrict__ n_seq_dev, float* __restrict__ lse) {
  __shared__ __attribute__((aligned(16))) char smem[4 * KV_TILE];  // K0 K1 V0 V1
  constexpr int QBQ = QB * QT;           // query rows per workgroup
  const int qblocks = L / QBQ;
  // XCD-aware mapping: workgroup ids go round-robin over the 8 XCDs (private
  // L2s), so the query blocks of one (sequence, head) -- which share its K and
  // V^T -- are given ids of the same residue mod 8 when the grid allows it.
  int bid = blockIdx.x;
  if ((gridDim.x & 7) == 0 && ((gridDim.x >> 3) % qblocks) == 0) {
    const int xcd = bid & 7, j = bid >> 3;
    bid = ((j / qblocks) * 8 + xcd) * qblocks + (j % qblocks);
  }
  const int qb = bid % qblocks;
  const int bh = bid / qblocks;        // seq * H + head
  const int seq = bh / H, head = bh - seq * H;
  if (n_seq_dev && seq >= *n_seq_dev) return;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int q0 = qb * QBQ + wid * (32 * QT);

  const bf16_t* qp = q + ((int64_t)bh * L + q0 + r) * HD;
  const bf16_t* kp = k + (int64_t)bh * L * HD;
  const bf16_t* vp = vt + (int64_t)bh * HD * L;

  // Q fragments: B operand, lane (q=r, half hh) holds Q[q][16*ks + 8*hh + j]
  bf16x8 qf[QT][4];
#pragma unroll
  for (int t = 0; t < QT; ++t)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      qf[t][ks] = *reinterpret_cast<const bf16x8*>(qp + t * 32 * HD + ks * 16 + hh * 8);

  // staging: per tile 8 KiB for K and 8 KiB for V^T = 8 + 8 wave-instructions;
  // wave w issues chunks {w, w+4}.  chunk c -> tile rows c*8..c*8+7.
  const int srow = lane >> 3;
  auto stage = [&](int buf, int key0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = wid + i * 4;
      const int row = c * 8 + srow;
      const int logical = (lane & 7) ^ ((row >> 1) & 7);
      GLDS16(kp + (int64_t)(key0 + row) * HD + logical * 8, smem + buf * KV_TILE + c * 1024);
      GLDS16(vp + (int64_t)row * L + key0 + logical * 8, smem + (2 + buf) * KV_TILE + c * 1024);
    }
  };

  // fragment read offsets.  K: A-row r of key-tile kt2 reads tile row
  // kt2*32 + pi(r), pi = swap bits 2 and 3; logical chunk ks*2 + hh.
  const int pr = (r & ~0xC) | ((r & 4) << 1) | ((r & 8) >> 1);
  int koff[2][4];
#pragma unroll
  for (int kt2 = 0; kt2 < 2; ++kt2) {
    const int row = kt2 * 32 + pr;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      koff[kt2][ks] = row * 128 + (((ks * 2 + hh) ^ ((row >> 1) & 7)) * 16);
  }
  // V^T: A-row r of d-tile dt reads tile row d = dt*32 + r; k-step (kt2, s)
  // needs keys kt2*32 + 16*s + 8*hh + (0..7) = logical chunk kt2*4 + 2*s + hh.
  int voff[2][4];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
    const int row = dt * 32 + r;
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4)  // c4 = kt2*2 + s
      voff[dt][c4] = row * 128 + ((((c4 >> 1) * 4 + (c4 & 1) * 2 + hh) ^ ((row >> 1) & 7)) * 16);
  }

  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  f32x16 oacc[QT][2];
#pragma unroll
  for (int t = 0; t < QT; ++t)
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) oacc[t][dt] = zero16;
  float m_run[QT], l_run[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) { m_run[t] = -INFINITY; l_run[t] = 0.f; }
  const float sc = 0.125f * 1.4426950408889634f;  // softmax scale * log2(e)

  const int NT = L / KB;
  stage(0, 0);
  // a wave's LDS-DMA must have landed before any wave reads the tile: explicit
  // drain, not left to the compiler's waitcnt insertion
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int tt = 0; tt < NT; ++tt) {
    const int cur = tt & 1;
    if (tt + 1 < NT) stage(cur ^ 1, (tt + 1) * KB);
    const char* Kt = smem + cur * KV_TILE;
    const char* Vt = smem + (2 + cur) * KV_TILE;

    // ---- S^T = K Q^T (K fragments shared by the QT query tiles) -----------
    // all eight K fragments are requested before the first MFMA, and the V^T
    // fragments right behind them, so that their LDS latency is covered by the
    // QK^T MFMAs and the softmax instead of stalling every MFMA pair
    bf16x8 kf[2][4], vf[2][4];
#pragma unroll
    for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        kf[kt2][ks] = *reinterpret_cast<const bf16x8*>(Kt + koff[kt2][ks]);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4)
        vf[dt][c4] = *reinterpret_cast<const bf16x8*>(Vt + voff[dt][c4]);
    f32x16 sacc[QT][2];
#pragma unroll
    for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int t = 0; t < QT; ++t)
          sacc[t][kt2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
            kf[kt2][ks], qf[t][ks], ks == 0 ? zero16 : sacc[t][kt2], 0, 0, 0);
    // keep all sixteen fragment reads ahead of the MFMAs (hipcc otherwise sinks
    // each K read to just before its MFMA pair: read -> wait -> 2 MFMAs)
    __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 8 * QT, 0);
    // ---- online softmax (per query column; keys in-lane + other half) ----
    // The running max is a SCALE, not a bound: p = exp2(s - m_run) is exact in
    // fp32 / bf16 whatever its magnitude, and O / l is invariant to it.  So the
    // tile max (32 VALU ops per query tile and key tile) is only computed for the
    // first key tile, and again when a tile's row sum leaves the safe range
    // (some score exceeds m_run by > 2^100 in the exp2 domain: never, in
    // practice); the re-evaluation then runs from the scores still in registers.
    bf16x8 pf[QT][4];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      // with_max: raise the running max to this tile's max first (and rescale)
      auto softmax_tile = [&](bool with_max) -> float {
        if (with_max) {
          // keeps the compiler from hoisting the 32 max ops out of the rare path
          asm volatile("" : "+v"(sacc[t][0][0]));
          float mx = sacc[t][0][0];
#pragma unroll
          for (int e = 1; e < 16; ++e) mx = fmaxf(mx, sacc[t][0][e]);
#pragma unroll
          for (int e = 0; e < 16; ++e) mx = fmaxf(mx, sacc[t][1][e]);
          mx = fmaxf(mx, __shfl_xor(mx, 32));
          const float m_new = fmaxf(m_run[t], mx * sc);
          const float alpha = __builtin_amdgcn_exp2f(m_run[t] - m_new);   // 0 on the first tile
          l_run[t] *= alpha;
#pragma unroll
          for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) oacc[t][dt][e] *= alpha;
          m_run[t] = m_new;
        }
        // p = exp2(s * sc - m): packed fma / add (two elements per VALU
        // instruction), one v_exp_f32 per element, one v_cvt_pk_bf16_f32 per pair
        f32x2 lsum2 = {0.f, 0.f};
        const f32x2 sc2 = {sc, sc}, nm2 = {-m_run[t], -m_run[t]};
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
          const int kt2 = c4 >> 1, s8 = (c4 & 1) * 8;
          unsigned pk[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            f32x2 x = {sacc[t][kt2][s8 + 2 * e], sacc[t][kt2][s8 + 2 * e + 1]};
            x = __builtin_elementwise_fma(x, sc2, nm2);
            f32x2 pp = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
            lsum2 += pp;
            const bf16x2 pb = __builtin_convertvector(pp, bf16x2);
            pk[e] = __builtin_bit_cast(unsigned, pb);
          }
          const uint4 pk4 = make_uint4(pk[0], pk[1], pk[2], pk[3]);
          pf[t][c4] = __builtin_bit_cast(bf16x8, pk4);
        }
        return lsum2[0] + lsum2[1];
      };
      float lsum = softmax_tile(false);
      // -inf scale (nothing seen yet: the first tile) gives inf / NaN, a score
      // 2^100 above the scale gives inf: both redo the tile with its max
      if (__builtin_expect(__any(!(lsum < 1e30f)), 0)) lsum = softmax_tile(true);
      l_run[t] += lsum;
    }
    // ---- O^T += V^T P^T (V^T fragments shared by the QT query tiles) -------
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) {
#pragma unroll
        for (int t = 0; t < QT; ++t)
          oacc[t][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[dt][c4], pf[t][c4], oacc[t][dt], 0, 0, 0);
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next tile's LDS-DMA landed
    __syncthreads();
  }

  // ---- normalise and store O[q, head*64 + d] ------------------------------
  // A lane holds 4 consecutive d of its query row in 8 places: stored straight
  // from registers that is sixteen 8-byte stores per tile, 16 bytes per row per
  // instruction.  The K/V stages are free now (every wave passed the last
  // barrier), so each wave transposes its [32 q][64 d] bf16 tile through 4 KiB
  // of LDS (16-byte chunk swizzled by row & 7) and writes whole 128-byte rows.
  char* ot = smem + wid * (QT * 4096);
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const float l_tot = l_run[t] + __shfl_xor(l_run[t], 32);
    const float inv = 1.f / l_tot;
    // training: log2-domain log-sum-exp of the scaled scores, read by the backward
    if (lse && hh == 0) lse[(int64_t)bh * L + q0 + t * 32 + r] = m_run[t] + __log2f(l_tot);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int c16 = dt * 4 + g4;           // 16-byte chunk: d = 8*c16 + 4*hh + (0..3)
        *reinterpret_cast<uint2*>(ot + t * 4096 + r * 128 + ((c16 ^ (r & 7)) * 16) + hh * 8) = make_uint2(
          pack_bf16x2(oacc[t][dt][4 * g4 + 0] * inv, oacc[t][dt][4 * g4 + 1] * inv),
          pack_bf16x2(oacc[t][dt][4 * g4 + 2] * inv, oacc[t][dt][4 * g4 + 3] * inv));
      }
  }
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    bf16_t* op = out + ((int64_t)seq * L + q0 + t * 32) * ((int64_t)H * HD) + head * HD;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 8 + (lane >> 3), ch = lane & 7;
      const uint4 v = *reinterpret_cast<const uint4*>(ot + t * 4096 + row * 128 + ((ch ^ (row & 7)) * 16));
      *reinterpret_cast<uint4*>(op + (int64_t)row * ((int64_t)H * HD) + ch * 8) = v;
    }
  }
}


// ---------------------------------------------------------------------------
// Software-pipelined variant (L % 256 == 0): same data layout, same MFMA plan,
// same LDS images.  At head_dim 64 the softmax costs more VALU issue time than
// the two matrix products cost MFMA time (per 32 x 32 score block: 8 MFMAs =
// 256 pipe cycles against ~350 cycles of VALU issue), and MFMA and VALU only
// overlap when they are interleaved in ONE wave's instruction stream.  The
// kernel above runs QK^T, softmax and PV of a tile one after the other (its
// PMC profile: MFMA busy 0.35 + VALU issue 0.54, i.e. no overlap).  Here a wave
// walks (query tile t, 32-key sub-block) blocks, t fastest, and step j issues
//     QK^T of block j+1   |   softmax of block j   |   PV of block j-1
// so that every MFMA has independent VALU work next to it.  Consecutive blocks
// belong to different query tiles: the softmax / rescale of tile t touch
// oacc[t] while the PV MFMAs in flight write oacc[1 - t].  K / V^T tiles sit in
// a 4-deep LDS ring (tile i+2 is fetched while i and i+1 are being read), one
// barrier per tile.  The running max is only raised when some row's block max
// is a SCALE, not a bound (p = exp2(s - m_run) is exact in fp32 / bf16 whatever
// its magnitude and O / l does not depend on it), so no block max is computed:
// only when a block's row sum leaves the safe range (first block: m_run = -inf
// gives inf; later: a score 2^100 above the scale) is the block redone with its
// max and the accumulators rescaled -- 16 of ~70 VALU instructions per block
// saved.  Measured (B = 64, L = 1024, random data, same process): 0.79 PF for the
// kernel above -> 0.88 without the tile max -> 0.93-0.96 pipelined.
constexpr int PIPE_STAGES = 4;
constexpr int VALU_PER_MFMA = 5;   // sched pin: VALU / exp instructions placed behind each MFMA

__device__ __forceinline__ float max_other_half(float x) {
  // lanes l and l ^ 32 hold the two key halves of a query column
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const unsigned u = __float_as_uint(x);
  const u32x2 r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

__global__ __launch_bounds__(AT, 2) void attn_fwd_pipe_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
    const bf16_t* __restrict__ vt, bf16_t* __restrict__ out, int H, int L,
    const int32_t* __restrict__ n_seq_dev, float* __restrict__ lse) {
  __shared__ __attribute__((aligned(16))) char smem[PIPE_STAGES * 2 * KV_TILE];  // (K, V^T) x 4
  constexpr int QBQ = QB * 2;
  const int qblocks = L / QBQ;
  int bid = blockIdx.x;
  if ((gridDim.x & 7) == 0 && ((gridDim.x >> 3) % qblocks) == 0) {
    const int xcd = bid & 7, j = bid >> 3;
    bid = ((j / qblocks) * 8 + xcd) * qblocks + (j % qblocks);
  }
  const int qb = bid % qblocks;
  const int bh = bid / qblocks;
  const int seq = bh / H, head = bh - seq * H;
  if (n_seq_dev && seq >= *n_seq_dev) return;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int q0 = qb * QBQ + wid * 64;
  const bf16_t* qp = q + ((int64_t)bh * L + q0 + r) * HD;
  const bf16_t* kp = k + (int64_t)bh * L * HD;
  const bf16_t* vp = vt + (int64_t)bh * HD * L;

  bf16x8 qf[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      qf[t][ks] = *reinterpret_cast<const bf16x8*>(qp + t * 32 * HD + ks * 16 + hh * 8);

  const int srow = lane >> 3;
  auto stage = [&](int st, int key0) {
    char* const kb = smem + st * (2 * KV_TILE);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = wid + i * 4;
      const int row = c * 8 + srow;
      const int logical = (lane & 7) ^ ((row >> 1) & 7);
      GLDS16(kp + (int64_t)(key0 + row) * HD + logical * 8, kb + c * 1024);
      GLDS16(vp + (int64_t)row * L + key0 + logical * 8, kb + KV_TILE + c * 1024);
    }
  };
  const int pr = (r & ~0xC) | ((r & 4) << 1) | ((r & 8) >> 1);
  // LDS byte offsets of this lane's fragments inside a stage: the swizzle term
  // (row >> 1) & 7 does not change with +32 rows, so the second key half / second
  // d block is the same offset + 4096 (an instruction immediate)
  int koff[4], voff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
    koff[ks] = pr * 128 + (((ks * 2 + hh) ^ ((pr >> 1) & 7)) * 16);
#pragma unroll
  for (int c4 = 0; c4 < 4; ++c4)
    voff[c4] = KV_TILE + r * 128 + ((((c4 >> 1) * 4 + (c4 & 1) * 2 + hh) ^ ((r >> 1) & 7)) * 16);

  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  f32x16 oacc[2][2] = {{zero16, zero16}, {zero16, zero16}};
  f32x16 S[2];
  bf16x8 P[2][2];
  bf16x8 kf[2][4], vf[2][4];
  float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
  const float sc = 0.125f * 1.4426950408889634f;

  auto load_k = [&](int kt2, int st) {
    const char* base = smem + st * (2 * KV_TILE);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kf[kt2][ks] = *reinterpret_cast<const bf16x8*>(base + koff[ks] + kt2 * 4096);
  };
  auto load_v = [&](int kt2, int st) {
    const char* base = smem + st * (2 * KV_TILE);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
        vf[dt][kt2 * 2 + s2] = *reinterpret_cast<const bf16x8*>(base + voff[kt2 * 2 + s2] + dt * 4096);
  };

  const int NT = L / KB;
  stage(0, 0);
  if (NT > 1) stage(1, KB);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  load_k(0, 0); load_k(1, 0); load_v(0, 0); load_v(1, 0);
  {
    const bf16x8 z = __builtin_bit_cast(bf16x8, make_uint4(0u, 0u, 0u, 0u));
    P[0][0] = z; P[0][1] = z; P[1][0] = z; P[1][1] = z;   // the first PV adds 0
  }
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
    S[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0][ks], qf[0][ks], ks == 0 ? zero16 : S[0], 0, 0, 0);

  // one pipeline step: t = query tile of the block whose softmax runs now; NK /
  // PK = key sub-block (0/1) of the next block's QK^T and of the previous
  // block's PV
  auto step = [&](auto t_c, auto nk_c, auto pk_c) {
    constexpr int T = decltype(t_c)::value, NK = decltype(nk_c)::value, PK = decltype(pk_c)::value;
    constexpr int O = 1 - T;
    // MFMA order: the links of the S[O] chain (each waits for its predecessor's
    // result) alternate with the four independent P.V products of the previous block
    auto pv = [&](int i) {
      const int s2 = i >> 1, dt = i & 1;
      oacc[O][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[dt][PK * 2 + s2], P[O][s2], oacc[O][dt], 0, 0, 0);
    };
    S[O] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[NK][0], qf[O][0], zero16, 0, 0, 0);
    pv(0);
    S[O] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[NK][1], qf[O][1], S[O], 0, 0, 0);
    pv(1);
    S[O] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[NK][2], qf[O][2], S[O], 0, 0, 0);
    pv(2);
    S[O] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[NK][3], qf[O][3], S[O], 0, 0, 0);
    pv(3);
    auto softmax_block = [&](bool with_max) -> float {
      if (with_max) {
        asm volatile("" : "+v"(S[T][0]));   // keep the max ops inside the rare path
        float mx = fmaxf(fmaxf(S[T][0], S[T][1]), S[T][2]);
#pragma unroll
        for (int e = 3; e < 15; e += 2) mx = fmaxf(fmaxf(mx, S[T][e]), S[T][e + 1]);
        mx = fmaxf(mx, S[T][15]);
        mx = max_other_half(mx) * sc;
        const float m_new = fmaxf(m_run[T], mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run[T] - m_new);
        l_run[T] *= alpha;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int e = 0; e < 16; ++e) oacc[T][dt][e] *= alpha;
        m_run[T] = m_new;
      }
      const float nm = -m_run[T];
      float ls = 0.f;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        unsigned pk[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(S[T][s2 * 8 + 2 * e], sc, nm));
          const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(S[T][s2 * 8 + 2 * e + 1], sc, nm));
          ls += p0; ls += p1;
          pk[e] = pack_bf16x2(p0, p1);
        }
        P[T][s2] = __builtin_bit_cast(bf16x8, make_uint4(pk[0], pk[1], pk[2], pk[3]));
      }
      return ls;
    };
    float ls = softmax_block(false);
    // one MFMA, then its share of the VALU work (6 MFMAs, ~64 VALU + 16 exp)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x402, VALU_PER_MFMA, 0);
    }
    if (__builtin_expect(__any(!(ls < 1e30f)), 0)) ls = softmax_block(true);
    l_run[T] += ls;
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  for (int tt = 0; tt < NT; ++tt) {
    const int st_cur = tt & (PIPE_STAGES - 1), st_next = (tt + 1) & (PIPE_STAGES - 1);
    // tile tt+1 has landed (fetched one tile ago); every wave is past the reads
    // of tile tt-2, whose stage receives tile tt+2
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tt + 2 < NT) stage((tt + 2) & (PIPE_STAGES - 1), (tt + 2) * KB);
    // b0: softmax (t0, k0) | QK (t1, k0) | PV (t1, k1 of the previous tile)
    step(I0{}, I0{}, I1{});
    load_k(0, st_next);        // kf[0] <- next tile (its last use was the QK just issued)
    load_v(1, st_cur);         // vf[.][2..3] <- this tile's second key half
    // b1: softmax (t1, k0) | QK (t0, k1) | PV (t0, k0)
    step(I1{}, I1{}, I0{});
    // b2: softmax (t0, k1) | QK (t1, k1) | PV (t1, k0)
    step(I0{}, I1{}, I0{});
    load_k(1, st_next);
    load_v(0, st_next);
    // b3: softmax (t1, k1) | QK (t0, k0 of the next tile) | PV (t0, k1)
    step(I1{}, I0{}, I1{});
  }
  // PV of the last block (t1, k1 of the last tile)
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
      oacc[1][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[dt][2 + s2], P[1][s2], oacc[1][dt], 0, 0, 0);
  __syncthreads();   // every wave is done with the ring: reuse it for the output transposition

  char* ot = smem + wid * (2 * 4096);
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const float l_tot = l_run[t] + __shfl_xor(l_run[t], 32);
    const float inv = 1.f / l_tot;
    if (lse && hh == 0) lse[(int64_t)bh * L + q0 + t * 32 + r] = m_run[t] + __log2f(l_tot);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int c16 = dt * 4 + g4;
        *reinterpret_cast<uint2*>(ot + t * 4096 + r * 128 + ((c16 ^ (r & 7)) * 16) + hh * 8) = make_uint2(
          pack_bf16x2(oacc[t][dt][4 * g4 + 0] * inv, oacc[t][dt][4 * g4 + 1] * inv),
          pack_bf16x2(oacc[t][dt][4 * g4 + 2] * inv, oacc[t][dt][4 * g4 + 3] * inv));
      }
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    bf16_t* op = out + ((int64_t)seq * L + q0 + t * 32) * ((int64_t)H * HD) + head * HD;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 8 + (lane >> 3), ch = lane & 7;
      const uint4 v = *reinterpret_cast<const uint4*>(ot + t * 4096 + row * 128 + ((ch ^ (row & 7)) * 16));
      *reinterpret_cast<uint4*>(op + (int64_t)row * ((int64_t)H * HD) + ch * 8) = v;
    }
  }
}

}  // namespace mdlm

using namespace mdlm;

static int attn_fwd_launch(const void* q, const void* k, const void* vt,
                           void* out, int n_seq, int H, int L,
                           const int32_t* n_seq_dev, float* lse, void* stream) {
  if (n_seq == 0) return 0;
  MDLM_CHECK_ARG(q && k && vt && out, "attn_fwd: null pointer");
  MDLM_CHECK_ARG(n_seq >= 0 && H > 0 && L > 0 && L % QB == 0, "attn_fwd: L=%d must be a multiple of %d", L, QB);
  MDLM_CHECK_ARG((((uintptr_t)q | (uintptr_t)k | (uintptr_t)vt | (uintptr_t)out) & 15) == 0, "attn_fwd: buffers must be 16-byte aligned");
  if (n_seq == 0) return 0;
  const int qt = (L % (2 * QB) == 0) ? 2 : 1;
  const int64_t blocks = (int64_t)n_seq * H * (L / (QB * qt));
  MDLM_CHECK_ARG(blocks < (1ll << 31), "attn_fwd: too many blocks");
  if (qt == 2)
    attn_fwd_pipe_kernel<<<(unsigned)blocks, AT, 0, (hipStream_t)stream>>>(
      (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)vt, (bf16_t*)out, H, L, n_seq_dev, lse);
  else
    attn_fwd_kernel<1><<<(unsigned)blocks, AT, 0, (hipStream_t)stream>>>(
      (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)vt, (bf16_t*)out, H, L, n_seq_dev, lse);
  MDLM_LAUNCH_CHECK();
  return 0;
}

extern "C" int mdlm_attn_fwd(const void* q, const void* k, const void* vt,
                             void* out, int n_seq, int H, int L,
                             const int32_t* n_seq_dev, void* stream) {
  return attn_fwd_launch(q, k, vt, out, n_seq, H, L, n_seq_dev, nullptr, stream);
}

extern "C" int mdlm_attn_fwd_lse(const void* q, const void* k, const void* vt,
                                 void* out, float* lse, int n_seq, int H, int L,
                                 void* stream) {
  MDLM_CHECK_ARG(lse || n_seq == 0, "attn_fwd_lse: null lse");
  return attn_fwd_launch(q, k, vt, out, n_seq, H, L, nullptr, lse, stream);
}
