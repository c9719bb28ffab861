// bf16 MFMA GEMM with fused epilogues for the DiT linears (gfx950).
//   C[M,N] = epilogue(A[M,K] @ W[N,K]^T)     (nn.Linear layout: both K-contiguous)
// Reference call sites: models/dit.py:256 (QKV), 277 (out-proj + gate/residual),
// 225-228 (MLP, GELU-tanh), 306/320 (vocab projection), 112-115 (rotary).
//
// Two tile configurations of one kernel template:
//   256x256x64, 512 threads = 8 waves (2 M x 4 N), wave tile 128x64  (M % 256 == 0)
//   128x128x64, 256 threads = 4 waves (2 M x 2 N), wave tile  64x64  (M % 128 == 0)
// MFMA 16x16x32 bf16, fp32 accumulators.  A and W tiles are staged global->LDS
// with 16-byte global_load_lds (LDS-DMA), double-buffered.  The LDS image is
// lane-linear with the 16-byte chunk index XOR-swizzled by (row & 7) on the
// SOURCE address and un-swizzled on the ds_read_b128 fragment reads
// (conflict-free for the 16x16x32 operand map).
//
// Operand order: the MFMA is issued as D = W_frag x A_frag, i.e. it computes
// 16x16 tiles of C^T: a lane then holds FOUR CONSECUTIVE OUTPUT COLUMNS of one
// output row (D row = n, D col = m).  That makes the epilogue cheap: bias /
// GELU / rotary are applied in registers, four values are packed and written
// with ONE ds_write_b64 (bf16) or ds_write_b128 (fp32) into a swizzled
// per-wave LDS tile, and read back as whole rows for 16-byte coalesced global
// stores.  Only the V slice of the QKV GEMM keeps the natural order (a lane
// then holds four consecutive ROWS = positions, which is what the transposed
// v^T[b,h,d,l] store wants).
// Block ids are remapped so that each XCD (private 4 MiB L2) walks groups of
// M-tiles x all N-tiles: the A panels stay L2-resident while W streams.
#include "gemm_common.h"

namespace mdlm {

// Per-lane column constants of one wave tile, fetched BEFORE the main loop so
// their latency is hidden (interior wave tiles only; `ok` is wave-uniform).
//   bf16 outputs: bias for columns gcol0 + n*16 + fq*4 + (0..3)   (register layout)
//   fp32 outputs: bias / gate for columns gcol0 + (lane&15)*4 + (0..3)  (row layout)
struct EpiConsts {
  float4 bias4[4];
  float4 b4, gt;
  bool ok;       // interior wave tile with 16-byte row stores: vector path
  bool gt_tile;  // gt holds the gate row of the whole wave tile (one sample)
};

// Row statistics of the vocabulary projection, taken from the accumulators while
// they are still in registers (SUBS max / sum out of the vocab-GEMM
// epilogue; reference diffusion.py:263-268 after models/dit.py:317-321).  For every
// row of the MS*16-row sub-tile and this wave's 64 columns:
//   m = max_v y_v,  s = sum_v exp(y_v - m),   y = bf16(acc + bias), exactly the
// values the epilogue stores; column lse_skip (the mask token, whose SUBS logit is
// shifted by -1e6, i.e. contributes exp(.) = 0) and columns >= N excluded.
// A lane holds 16 of a row's 64 values (4 per n); lanes fr, fr+16, fr+32, fr+48
// share the row: in-lane reduction, then two cross-lane steps.  Layout of the
// partials: [row / 256][group = column / 64][row % 256] float2 -- 16 consecutive
// rows per store instruction are contiguous (128 bytes); mdlm_lse_combine reduces
// the groups of a row.
template <int MT, int M0, int MS>
__device__ __forceinline__ void lse_partials(const f32x4 (&accf)[MT][4], int64_t grow0, int gcol0,
                                             const EpiConsts& ec, const GemmArgs& g, int lane) {
  const int fq = lane >> 4, fr = lane & 15;
  constexpr float L2E = 1.4426950408889634f;
  const bool plain = ec.ok && !(g.lse_skip >= gcol0 && g.lse_skip < gcol0 + 64);   // wave-uniform
  float bias_e[4][4];
  if (!plain) {
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = gcol0 + n * 16 + fq * 4 + j;
        bias_e[n][j] = (g.bias && col < g.N) ? g.bias[col] : 0.f;
      }
  }
#pragma unroll
  for (int m = 0; m < MS; ++m) {
    float y[16];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const f32x4 a = accf[M0 + m][n];
      if (plain) {
        const f32x2 lo = bf16_round2(a[0] + ec.bias4[n].x, a[1] + ec.bias4[n].y);
        const f32x2 hi = bf16_round2(a[2] + ec.bias4[n].z, a[3] + ec.bias4[n].w);
        y[n * 4 + 0] = lo.x; y[n * 4 + 1] = lo.y; y[n * 4 + 2] = hi.x; y[n * 4 + 3] = hi.y;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int col = gcol0 + n * 16 + fq * 4 + j;
          y[n * 4 + j] = (col < g.N && col != g.lse_skip) ? bf16_round(a[j] + bias_e[n][j]) : -INFINITY;
        }
      }
    }
    float mx = fmaxf(fmaxf(y[0], y[1]), y[2]);
#pragma unroll
    for (int k = 3; k < 15; k += 2) mx = fmaxf(fmaxf(mx, y[k]), y[k + 1]);
    mx = fmaxf(mx, y[15]);
    const float nm = (mx == -INFINITY) ? 0.f : -mx * L2E;   // all 16 excluded: s = 0
    float sm = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) sm += __builtin_amdgcn_exp2f(__builtin_fmaf(y[k], L2E, nm));
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
      const float mo = __shfl_xor(mx, o), so = __shfl_xor(sm, o);
      const float M = fmaxf(mx, mo);
      const float ea = (mx == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((mx - M) * L2E);
      const float eb = (mo == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((mo - M) * L2E);
      sm = sm * ea + so * eb;
      mx = M;
    }
    if (fq == 0) {
      const int64_t R = grow0 + m * 16 + fr;
      g.lse_part[((R >> 8) * g.lse_ng + (gcol0 >> 6)) * 256 + (R & 255)] = make_float2(mx, sm);
    }
  }
}

// Epilogue of MS*16 rows x 64 columns of a wave tile (m sub-tiles M0..M0+MS-1).
// SWAPPED accumulators (see header):
//   acc(m, n)[j] = C[grow0 + m*16 + fr][gcol0 + n*16 + fq*4 + j].
// ep = this wave's 8 KiB LDS region inside the stage buffer the main loop read
// last (the other one already holds the next tile's first K slab):
// [64 rows][64 bf16] or [32 rows][64 fp32].  rows_per_mod % 64 == 0 whenever
// rows are decomposed into (sample, position), so a sub-tile never straddles
// two samples.
template <int EPI, typename TC, int MT, int M0, int MS>
__device__ __forceinline__ void epilogue_sub(const f32x4 (&accf)[MT][4], char* ep,
                                             int64_t grow0, int gcol0, const EpiConsts& ec,
                                             const GemmArgs& g, int lane) {
  const int fq = lane >> 4, fr = lane & 15;
  if (gcol0 >= g.N) return;  // wave tile entirely past the N edge (wave-uniform)
#define acc(m, n) accf[M0 + (m)][n]
  constexpr bool BF16_OUT = sizeof(TC) == 2;

  if constexpr (EPI == MDLM_EPI_QKV_ROTARY) {
    static_assert(MS == 4, "rotary epilogue works on 64-row sub-tiles");
    const int cg = gcol0 >> 6;               // (three, head) column group
    const int three = cg / g.n_heads;
    const int h = cg - three * g.n_heads;
    const int64_t L = g.rows_per_mod;
    const int64_t b = grow0 / L;
    const int64_t pos0 = grow0 - b * L;
    bf16_t* const base = reinterpret_cast<bf16_t*>(g.C) + (int64_t)three * g.M * g.n_heads * 64;
    if (three == 2) {
      // NATURAL accumulators here: acc(m, n)[j] = C[m*16 + fq*4 + j][n*16 + fr];
      // v^T[b,h,d,l]: a lane holds 4 consecutive l of one d.  Transposed
      // through the LDS tile [64 d][64 l] (16-B chunk swizzled by d & 7) so the
      // global stores are whole 128-byte rows of v^T.
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          const int d = n * 16 + fr;
          const int c16 = m * 2 + (fq >> 1);
          *reinterpret_cast<uint2*>(ep + d * 128 + ((c16 ^ (d & 7)) * 16) + (fq & 1) * 8) =
            make_uint2(pack_bf16x2(acc(m, n)[0], acc(m, n)[1]),
                       pack_bf16x2(acc(m, n)[2], acc(m, n)[3]));
        }
      bf16_t* const vb = base + ((b * g.n_heads + h) * 64) * L + pos0;
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int d = it * 8 + (lane >> 3), ch = lane & 7;
        const uint4 v = *reinterpret_cast<const uint4*>(ep + d * 128 + ((ch ^ (d & 7)) * 16));
        store16<true>(vb + (int64_t)d * L + ch * 8, v, GEMM_DBG(g));
      }
      return;
    }
    // q / k: rotary in registers.  d = n*16 + fq*4 + j pairs with d + 32
    // (tiles n and n+2); Linear output is bf16 under autocast (dit.py:256,
    // 365); the rotary kernel reads bf16, rotates in fp32, writes bf16
    // (dit.py:112-115, 264).  All 16 table loads are issued first.
    float4 cs[4][2], sn[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const int64_t o = (pos0 + m * 16 + fr) * 32 + n * 16 + fq * 4;
        cs[m][n] = *reinterpret_cast<const float4*>(g.rot_cos + o);
        sn[m][n] = *reinterpret_cast<const float4*>(g.rot_sin + o);
      }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int row = m * 16 + fr;
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        // packed fp32 math on column pairs (0,1) and (2,3)
        const f32x2 c01 = {cs[m][n].x, cs[m][n].y}, c23 = {cs[m][n].z, cs[m][n].w};
        const f32x2 s01 = {sn[m][n].x, sn[m][n].y}, s23 = {sn[m][n].z, sn[m][n].w};
        const f32x2 a01 = bf16_round2(acc(m, n)[0], acc(m, n)[1]);
        const f32x2 a23 = bf16_round2(acc(m, n)[2], acc(m, n)[3]);
        const f32x2 b01 = bf16_round2(acc(m, n + 2)[0], acc(m, n + 2)[1]);
        const f32x2 b23 = bf16_round2(acc(m, n + 2)[2], acc(m, n + 2)[3]);
        const f32x2 l01 = __builtin_elementwise_fma(a01, c01, -(b01 * s01));
        const f32x2 l23 = __builtin_elementwise_fma(a23, c23, -(b23 * s23));
        const f32x2 h01 = __builtin_elementwise_fma(a01, s01, b01 * c01);
        const f32x2 h23 = __builtin_elementwise_fma(a23, s23, b23 * c23);
        const uint2 lo = make_uint2(pack_bf16x2(l01.x, l01.y), pack_bf16x2(l23.x, l23.y));
        const uint2 hi = make_uint2(pack_bf16x2(h01.x, h01.y), pack_bf16x2(h23.x, h23.y));
        // bf16 tile [64][64], 16-B chunk index swizzled by (row & 7)
        const int c_lo = n * 2 + (fq >> 1), c_hi = c_lo + 4;
        *reinterpret_cast<uint2*>(ep + row * 128 + ((c_lo ^ (row & 7)) * 16) + (fq & 1) * 8) = lo;
        *reinterpret_cast<uint2*>(ep + row * 128 + ((c_hi ^ (row & 7)) * 16) + (fq & 1) * 8) = hi;
      }
    }
    bf16_t* const qb = base + ((b * g.n_heads + h) * L + pos0) * 64;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = it * 8 + (lane >> 3), ch = lane & 7;
      const uint4 v = *reinterpret_cast<const uint4*>(ep + row * 128 + ((ch ^ (row & 7)) * 16));
      store16<true>(qb + row * 64 + ch * 8, v, GEMM_DBG(g));
    }
    return;
  } else {
    // output row of sub-tile row 0 (scatter of compacted samples, STORE only)
    int64_t orow0 = grow0;
    if constexpr (EPI == MDLM_EPI_STORE) {
      if (g.out_sample_map) {
        // Rows past the live count of a partly live tile (samples shorter than the
        // tile) belong to list entries past n_active: stale sample ids, whose cached
        // rows must not be touched.  (rows_per_mod % 64 == 0: a sub-tile is all live or all dead.)
        if (g.m_rows_dev && grow0 >= (int64_t)*g.m_rows_dev) return;
        const int64_t sidx = grow0 / g.rows_per_mod;
        orow0 = (int64_t)g.out_sample_map[sidx] * g.rows_per_mod + (grow0 - sidx * g.rows_per_mod);
      }
    }
    const float* gate_row = nullptr;
    if constexpr (EPI == MDLM_EPI_GATE_RESIDUAL) {
      // gate row is constant over the sub-tile (rows_per_mod % 64 == 0)
      const int64_t mrow = g.n_mod == 1 ? 0 : (grow0 / g.rows_per_mod) % g.n_mod;
      gate_row = g.gate + mrow * g.mod_ld;
    }
    TC* const Cp0 = reinterpret_cast<TC*>(g.C) + gcol0;
    // destination row of sub-tile row `row` (per-row scatter list: STORE only; -1 = skip)
    auto dst_row = [&](int row) -> int64_t {
      if constexpr (EPI == MDLM_EPI_STORE) {
        if (g.out_row_map) return (int64_t)g.out_row_map[grow0 + row];
      }
      return orow0 + row;
    };
    if (ec.ok) {
      if constexpr (BF16_OUT) {
        // destination rows first (scatter list entries are loads: all in
        // flight together, not one load -> wait -> store per row group)
        int64_t drs[MS * 2];
        bool mapped = false;
        if constexpr (EPI == MDLM_EPI_STORE) mapped = g.out_row_map != nullptr;
        if (mapped) {   // one wave-uniform branch around all the loads
          const int32_t* mp = g.out_row_map + grow0 + (lane >> 3);
#pragma unroll
          for (int it = 0; it < MS * 2; ++it) drs[it] = (int64_t)mp[it * 8];
        } else {
#pragma unroll
          for (int it = 0; it < MS * 2; ++it) drs[it] = orow0 + it * 8 + (lane >> 3);
        }
        if constexpr (EPI == EPI_BIAS_GELU_SAVE) {
          // first pass through the LDS tile: the pre-activation acc + bias -> aux
          // (one wave's LDS accesses execute in order: no barrier between the passes)
          TC* const Ap0 = reinterpret_cast<TC*>(g.aux) + gcol0;
#pragma unroll
          for (int m = 0; m < MS; ++m) {
            const int row = m * 16 + fr;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
              const int c16 = n * 2 + (fq >> 1);
              *reinterpret_cast<uint2*>(ep + row * 128 + ((c16 ^ (row & 7)) * 16) + (fq & 1) * 8) =
                make_uint2(pack_bf16x2(acc(m, n)[0] + ec.bias4[n].x, acc(m, n)[1] + ec.bias4[n].y),
                           pack_bf16x2(acc(m, n)[2] + ec.bias4[n].z, acc(m, n)[3] + ec.bias4[n].w));
            }
          }
#pragma unroll
          for (int it = 0; it < MS * 2; ++it) {
            const int row = it * 8 + (lane >> 3), ch = lane & 7;
            const uint4 v = *reinterpret_cast<const uint4*>(ep + row * 128 + ((ch ^ (row & 7)) * 16));
            store16<true>(Ap0 + drs[it] * g.ldc + ch * 8, v, GEMM_DBG(g));
          }
        }
        // pre-activation rows of the GELU-gradient epilogue, all loads in flight
        uint4 pre[MS * 2];
        if constexpr (EPI == EPI_MUL_GELU_GRAD) {
          const TC* const Ap0 = reinterpret_cast<const TC*>(g.aux) + gcol0;
#pragma unroll
          for (int it = 0; it < MS * 2; ++it)
            pre[it] = *reinterpret_cast<const uint4*>(Ap0 + drs[it] * g.ldc + (lane & 7) * 8);
        }
        // registers -> (bias, GELU) -> packed bf16 -> swizzled LDS tile [MS*16][64]
#pragma unroll
        for (int m = 0; m < MS; ++m) {
          const int row = m * 16 + fr;
#pragma unroll
          for (int n = 0; n < 4; ++n) {
            f32x2 lo = {acc(m, n)[0] + ec.bias4[n].x, acc(m, n)[1] + ec.bias4[n].y};
            f32x2 hi = {acc(m, n)[2] + ec.bias4[n].z, acc(m, n)[3] + ec.bias4[n].w};
            if constexpr (EPI == MDLM_EPI_BIAS_GELU || EPI == EPI_BIAS_GELU_SAVE) { lo = gelu_tanh2(lo); hi = gelu_tanh2(hi); }
            const int c16 = n * 2 + (fq >> 1);
            *reinterpret_cast<uint2*>(ep + row * 128 + ((c16 ^ (row & 7)) * 16) + (fq & 1) * 8) =
              make_uint2(pack_bf16x2(lo.x, lo.y), pack_bf16x2(hi.x, hi.y));
          }
        }
#pragma unroll
        for (int it = 0; it < MS * 2; ++it) {
          const int row = it * 8 + (lane >> 3), ch = lane & 7;
          uint4 v = *reinterpret_cast<const uint4*>(ep + row * 128 + ((ch ^ (row & 7)) * 16));
          if constexpr (EPI == EPI_MUL_GELU_GRAD) {
            // the bf16-rounded dX times gelu'(pre): the arithmetic of the separate
            // GELU-backward pass (train_ops.hip), without its HBM round trip
            const unsigned dv[4] = {v.x, v.y, v.z, v.w};
            const unsigned pv[4] = {pre[it].x, pre[it].y, pre[it].z, pre[it].w};
            unsigned o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const float d0 = __uint_as_float(dv[k] << 16) * gelu_tanh_grad_fast(__uint_as_float(pv[k] << 16));
              const float d1 = __uint_as_float(dv[k] & 0xffff0000u) * gelu_tanh_grad_fast(__uint_as_float(pv[k] & 0xffff0000u));
              o[k] = pack_bf16x2(d0, d1);
            }
            v = make_uint4(o[0], o[1], o[2], o[3]);
          }
          const int64_t dr = drs[it];
          if (dr >= 0) store16<true>(Cp0 + dr * g.ldc + ch * 8, v, GEMM_DBG(g));
        }
      } else {
        // fp32 tile [MS*16][64], 16-B chunk index swizzled by (row & 15).
        const int ch = lane & 15;
        const int gc = gcol0 + ch * 4;
        // The residual rows are fetched with all loads in flight, not one
        // load -> wait -> store per row.
        float4 rs[MS * 4];
        if constexpr (EPI == MDLM_EPI_GATE_RESIDUAL) {
          const float* rp = g.residual + (grow0 + (lane >> 4)) * g.ldc + gc;
#pragma unroll
          for (int it = 0; it < MS * 4; ++it)
            rs[it] = *reinterpret_cast<const float4*>(rp + (int64_t)(it * 4) * g.ldc);
        }
#pragma unroll
        for (int m = 0; m < MS; ++m) {
          const int row = m * 16 + fr;
#pragma unroll
          for (int n = 0; n < 4; ++n) {
            const int c16 = n * 4 + fq;
            *reinterpret_cast<f32x4*>(ep + row * 256 + ((c16 ^ (row & 15)) * 16)) = acc(m, n);
          }
        }
        const float4 b4 = ec.b4;
        float4 gt = ec.gt;
        if constexpr (EPI == MDLM_EPI_GATE_RESIDUAL) {
          if (!ec.gt_tile) gt = *reinterpret_cast<const float4*>(gate_row + gc);
        }
#pragma unroll
        for (int it = 0; it < MS * 4; ++it) {
          const int row = it * 4 + (lane >> 4);
          float4 v = *reinterpret_cast<const float4*>(ep + row * 256 + ((ch ^ (row & 15)) * 16));
          v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
          if constexpr (EPI == MDLM_EPI_GATE_RESIDUAL) {
            // residual + gate * x (bias_dropout_add_scale, eval), dit.py:28-34
            v.x = rs[it].x + gt.x * v.x; v.y = rs[it].y + gt.y * v.y;
            v.z = rs[it].z + gt.z * v.z; v.w = rs[it].w + gt.w * v.w;
          }
          const int64_t dr = dst_row(row);
          if (dr >= 0) store16<false>(Cp0 + dr * g.ldc + ch * 4, make_uint4(__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)), GEMM_DBG(g));
        }
      }
    } else {
      // N edge / unaligned ldc: element-wise masked stores straight from registers
#pragma unroll
      for (int m = 0; m < MS; ++m) {
        const int row = m * 16 + fr;
        const int64_t dr = dst_row(row);
        if (dr < 0) continue;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const int cc = n * 16 + fq * 4 + jj;
            if (gcol0 + cc < g.N) {
              float o = acc(m, n)[jj] + (g.bias ? g.bias[gcol0 + cc] : 0.f);
              if constexpr (EPI == EPI_BIAS_GELU_SAVE) {
                elem<TC>::store(reinterpret_cast<TC*>(g.aux) + gcol0 + dr * g.ldc + cc, o);
                o = gelu_tanh(o);
              }
              if constexpr (EPI == EPI_MUL_GELU_GRAD)
                o = bf16_round(o) * gelu_tanh_grad_fast(elem<TC>::load(reinterpret_cast<const TC*>(g.aux) + gcol0 + dr * g.ldc + cc));
              if constexpr (EPI == MDLM_EPI_BIAS_GELU) o = gelu_tanh(o);
              if constexpr (EPI == MDLM_EPI_GATE_RESIDUAL)
                o = g.residual[(grow0 + row) * g.ldc + gcol0 + cc] + gate_row[gcol0 + cc] * o;
              elem<TC>::store(Cp0 + dr * g.ldc + cc, o);
            }
          }
        }
      }
    }
    // row statistics AFTER the sub-tile's global stores have been issued: the
    // VALU work then runs while the stores drain
    if constexpr (EPI == MDLM_EPI_STORE && BF16_OUT) {
      if (g.lse_part) lse_partials<MT, M0, MS>(accf, grow0, gcol0, ec, g, lane);   // wave-uniform
    }
  }
#undef acc
}

// WM x WN waves; each wave MT x 4 MFMA tiles (MT*16 rows x 64 columns).
template <int EPI, typename TC, int WM, int WN, int MT>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN) / 4)
void gemm_bf16_kernel(GemmArgs g) {
  constexpr int NW = WM * WN;
  constexpr int TBM = WM * MT * 16, TBN = WN * 64;
  constexpr int A_BYTES = TBM * BK * 2, B_BYTES = TBN * BK * 2;
  constexpr int STAGE = A_BYTES + B_BYTES;   // one K slab of A and W; two stages
  constexpr int A_CH = (TBM / 8) / NW, B_CH = (TBN / 8) / NW;  // 1 KiB chunks per wave
  constexpr int GROUP_M = 1024 / TBM;  // 1024 rows of A (1.5 MiB at K=768) per group
  constexpr bool BF16_OUT = sizeof(TC) == 2;
  static_assert(STAGE / NW == 8192, "epilogue regions are 8 KiB per wave");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid / WN, wc = wid - wr * WN;
  // ---- XCD-aware, persistent, balanced tile walk --------------------------
  // The valid tiles (M tiles below the live row count x all N tiles) form one
  // sequence: groups of 1024 A rows x all N tiles, N outer inside a group, so
  // neighbours share A panels.  It is cut into 8 equal contiguous ranges, one
  // per XCD (workgroup b runs on XCD b & 7: its range stays in that XCD's
  // L2), and the gridDim/8 workgroups of an XCD stride through their range.
  // The cut is made on the live tile count, so the XCDs finish within one tile
  // of each other whatever the row count (a per-sample partition left up to
  // 7/8 of a sample's tiles of imbalance).  The K slabs of consecutive tiles
  // form ONE double-buffered stream: the last K step of tile i prefetches the
  // first slab of tile i+1, whose load latency and the stores of tile i overlap.
  const int bid = blockIdx.x;
  const int xcd = bid & 7;
  const int slots = gridDim.x >> 3;
  const int per_group = GROUP_M * g.tiles_n;
  const int64_t m_lim = g.m_rows_dev ? (int64_t)*g.m_rows_dev : g.M;
  const int64_t vm_live = (m_lim + TBM - 1) / TBM;
  const int vm = vm_live < g.tiles_m ? (int)vm_live : g.tiles_m;  // valid M tiles
  constexpr bool SPLITK = EPI == MDLM_EPI_STORE && !BF16_OUT;
  const int per_split = vm * g.tiles_n;
  const int total = SPLITK ? per_split * g.splits : per_split;
  const int per_xcd = (total + 7) >> 3;
  const int s_begin = xcd * per_xcd;
  const int s_end = total < s_begin + per_xcd ? total : s_begin + per_xcd;
  // Tail of an XCD's range.  Its `cnt` tiles run in rounds of `slots`
  // workgroups; when the last round holds at most slots/2 tiles, each of them is
  // cut into two half tiles of TBM/2 rows (same N, same K order per output row, so
  // the results are bit-identical): the partial round then costs about half a
  // round instead of a whole one.  (In the sampling loop the row count changes
  // every step: 12 n tiles of the N = 768 GEMMs on 256 workgroups, n = 40 +- 6.)
  // Only where it pays (same-process sweep over 30-54 active samples, scratch/sweep_half.py):
  // gate/residual -4.5 % / -2.6 % (K = 768 / 3072), QKV -1.9 %; the GELU and plain-store
  // instantiations sit at the 256-register limit and the second tile body spills them (+5 %).
#ifndef MDLM_GEMM_HALF
#define MDLM_GEMM_HALF 1
#endif
  constexpr bool HALF_OK = MDLM_GEMM_HALF && !SPLITK && MT == 8 &&
                           (EPI == MDLM_EPI_GATE_RESIDUAL || EPI == MDLM_EPI_QKV_ROTARY);
  const int cnt = s_end > s_begin ? s_end - s_begin : 0;
  const int full_r = cnt / slots;
  const int rx = cnt - full_r * slots;
  const bool split_tail = HALF_OK && !g.whole_tiles && rx > 0 && 2 * rx <= slots;
  const int n_full = split_tail ? full_r * slots : cnt;   // virtual indices below: whole tiles
  const int n_virt = split_tail ? n_full + 2 * rx : cnt;
  // tile at virtual index j of this XCD's range (-1: past its end); hf = -1 for a
  // whole tile, 0 / 1 for its upper / lower half
  auto next_tile = [&](int j, int& tm, int& tn, int& sp, int& hf) -> int {
    if (j >= n_virt) return -1;
    int s = s_begin + j;
    hf = -1;
    if (j >= n_full) {
      const int e = j - n_full;
      s = s_begin + n_full + (e >> 1);
      hf = e & 1;
    }
    sp = 0;
    if constexpr (SPLITK) {   // split outermost: neighbours share the K range
      sp = s / per_split;
      s -= sp * per_split;
    }
    const int gl = s / per_group;
    const int rem = s - gl * per_group;
    const int left = vm - gl * GROUP_M;
    const int gm = left < GROUP_M ? left : GROUP_M;  // the last group may be partial
    tn = rem / gm;
    tm = gl * GROUP_M + (rem - tn * gm);
    return j;
  };

  // ---- staging addresses -------------------------------------------------
  // chunk c = i*NW + wid covers tile rows c*8 .. c*8+7; lane -> (row lane>>3,
  // physical 16-B chunk lane&7) holding logical chunk (lane&7)^(lane>>3).
  const int srow = lane >> 3;
  const int skc = ((lane & 7) ^ srow) * 8;  // element offset inside the 64-wide K slab
  const bf16_t* a_src[A_CH];
  const bf16_t* b_src[B_CH];
  auto set_src = [&](int tm, int tn, int sp, int hf) {
    // a half tile stages the TBM rows from ITS first row (it reads the first
    // TBM/2 of them); rows past the matrix are clamped, never read
    const int64_t m0 = (int64_t)tm * TBM + (hf == 1 ? TBM / 2 : 0);
    const int n0 = tn * TBN;
    const int64_t kofs = (SPLITK ? (int64_t)sp * g.K : 0) + skc;
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      int64_t arow = m0 + (i * NW + wid) * 8 + srow;
      arow = arow < g.M ? arow : g.M - 1;
      a_src[i] = g.A + arow * g.lda + kofs;
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
      int brow = n0 + (i * NW + wid) * 8 + srow;
      brow = brow < g.N ? brow : g.N - 1;  // clamp the N edge (stores are masked)
      b_src[i] = g.W + (int64_t)brow * g.ldw + kofs;
    }
  };
  auto stage = [&](int buf, int k0) {
    char* const sb = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < A_CH; ++i)
      GLDS16(a_src[i] + k0, sb + (i * NW + wid) * 1024);
#pragma unroll
    for (int i = 0; i < B_CH; ++i)
      GLDS16(b_src[i] + k0, sb + A_BYTES + (i * NW + wid) * 1024);
  };

  // ---- fragment read offsets ---------------------------------------------
  // frag (sub-tile m, k-step ks): row = base + m*16 + (lane&15), logical chunk
  // = ks*4 + (lane>>4), physical = logical ^ (row&7) = logical ^ (lane&7)
  int foff[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
    foff[ks] = (lane & 15) * 128 + (((ks * 4 + (lane >> 4)) ^ (lane & 7)) * 16);
  const int b_base = A_BYTES + wc * 64 * 128;
  const int KT = g.K / BK;

  int tm, tn, sp, hf;
  int j = next_tile(bid >> 3, tm, tn, sp, hf);
  if (j < 0) return;
  set_src(tm, tn, sp, hf);
  stage(0, 0);
  int buf = 0;          // stage holding the slab the next K step reads
  // the LDS-DMA of this wave must have landed before any wave reads the stage:
  // explicit, not left to the compiler's waitcnt insertion
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  while (true) {
    int tm_next, tn_next, sp_next, hf_next;
    const int j_next = next_tile(j + slots, tm_next, tn_next, sp_next, hf_next);
    const int64_t m0 = (int64_t)tm * TBM + (hf == 1 ? TBM / 2 : 0);
    const int n0 = tn * TBN;
    const int gcol0 = n0 + wc * 64;

    // One tile: MTE = MT for a whole tile, MT / 2 for a half tile (each wave then
    // owns MTE * 16 rows; the staged A block starts at the half tile's first row)
    auto tile_body = [&](auto mte_c) {
    constexpr int MTE = decltype(mte_c)::value;
    const int a_base = wr * (MTE * 16) * 128;

    // ---- column constants of this wave tile (latency hidden by the K loop) --
    EpiConsts ec;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    ec.ok = g.vec_ok && (gcol0 + 63 < g.N);  // wave-uniform
    ec.gt_tile = false;
    ec.b4 = zero4; ec.gt = zero4;
#pragma unroll
    for (int n = 0; n < 4; ++n) ec.bias4[n] = zero4;
    if constexpr (EPI != MDLM_EPI_QKV_ROTARY) {
      if (ec.ok) {
        if constexpr (BF16_OUT) {
          if (g.bias) {
#pragma unroll
            for (int n = 0; n < 4; ++n)
              ec.bias4[n] = *reinterpret_cast<const float4*>(g.bias + gcol0 + n * 16 + (lane >> 4) * 4);
          }
        } else {
          const int gc = gcol0 + (lane & 15) * 4;
          if (g.bias) ec.b4 = *reinterpret_cast<const float4*>(g.bias + gc);
          if constexpr (EPI == MDLM_EPI_GATE_RESIDUAL) {
            // one gate row for the whole wave tile unless samples are shorter than it
            ec.gt_tile = g.n_mod == 1 || g.rows_per_mod % (MTE * 16) == 0;
            if (ec.gt_tile) {
              const int64_t growW = m0 + wr * (MTE * 16);
              const int64_t mrow = g.n_mod == 1 ? 0 : (growW / g.rows_per_mod) % g.n_mod;
              ec.gt = *reinterpret_cast<const float4*>(g.gate + mrow * g.mod_ld + gc);
            }
          }
        }
      }
    }

    f32x4 acc[MTE][4];
#pragma unroll
    for (int m = 0; m < MTE; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // natural operand order only for the V slice of the QKV GEMM (wave-uniform)
    bool natural = false;
    if constexpr (EPI == MDLM_EPI_QKV_ROTARY)
      natural = (gcol0 >> 6) / g.n_heads == 2;

    auto mainloop = [&](auto nat) {
      constexpr bool NAT = decltype(nat)::value;
      for (int kt = 0; kt < KT; ++kt) {
        if (kt + 1 < KT) {
          stage(buf ^ 1, (kt + 1) * BK);
        } else if (j_next >= 0) {
          set_src(tm_next, tn_next, sp_next, hf_next);   // first slab of the next tile
          stage(buf ^ 1, 0);
        }
        const char* At = smem + buf * STAGE + a_base;
        const char* Bt = smem + buf * STAGE + b_base;
        bf16x8 af[2][MTE], bfr[2][4];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
          for (int n = 0; n < 4; ++n)
            bfr[ks][n] = *reinterpret_cast<const bf16x8*>(Bt + n * 2048 + foff[ks]);
#pragma unroll
          for (int m = 0; m < MTE; ++m)
            af[ks][m] = *reinterpret_cast<const bf16x8*>(At + m * 2048 + foff[ks]);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int m = 0; m < MTE; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) {
              if constexpr (NAT)
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks][m], bfr[ks][n], acc[m][n], 0, 0, 0);
              else
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ks][n], af[ks][m], acc[m][n], 0, 0, 0);
            }
        // fragment reads run ahead of the MFMA groups that consume them: R0
        // reads up front, then one read behind every group of 4 MFMAs (same-
        // process A/B: 16 up front is 2-3 % faster for the bf16-output kernels,
        // 12 for the fp32 gate/residual one, whose K = 3072 shape loses 4 % at 16)
        if constexpr (MTE == 8) {
          constexpr int R0 = BF16_OUT ? 16 : 12;
          __builtin_amdgcn_sched_group_barrier(0x100, R0, 0);
#pragma unroll
          for (int i = 0; i < 24 - R0; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x008, 64 - 4 * (24 - R0), 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next slab's LDS-DMA landed
        __syncthreads();
        buf ^= 1;
      }
    };
    if constexpr (EPI == MDLM_EPI_QKV_ROTARY) {
      if (natural) mainloop(std::true_type{}); else mainloop(std::false_type{});
    } else {
      mainloop(std::false_type{});
    }

    // ---- epilogue through this wave's 8 KiB of the stage read last ----------
    // (dbg bit 2, timing experiments only: skip it)
    if (!(GEMM_DBG(g) & 2)) {
      // launder the lane id: the epilogue's per-lane addresses are then
      // recomputed per tile instead of being hoisted out of the tile loop and
      // spilled (their reloads would sit between the stores)
      int lane_e = lane;
      asm volatile("" : "+v"(lane_e));
      char* ep = smem + (buf ^ 1) * STAGE + wid * 8192;
      const int64_t growW = m0 + wr * (MTE * 16);
      if constexpr (BF16_OUT) {
        epilogue_sub<EPI, TC, MTE, 0, 4>(acc, ep, growW, gcol0, ec, g, lane_e);
        if constexpr (MTE == 8) epilogue_sub<EPI, TC, MTE, 4, 4>(acc, ep, growW + 64, gcol0, ec, g, lane_e);
      } else if constexpr (SPLITK) {
        GemmArgs gs = g;   // this split's partial-sum plane
        gs.C = reinterpret_cast<float*>(g.C) + (int64_t)sp * g.c_split_stride;
        epilogue_sub<EPI, TC, MTE, 0, 2>(acc, ep, growW, gcol0, ec, gs, lane_e);
        epilogue_sub<EPI, TC, MTE, 2, 2>(acc, ep, growW + 32, gcol0, ec, gs, lane_e);
        if constexpr (MTE == 8) {
          epilogue_sub<EPI, TC, MTE, 4, 2>(acc, ep, growW + 64, gcol0, ec, gs, lane_e);
          epilogue_sub<EPI, TC, MTE, 6, 2>(acc, ep, growW + 96, gcol0, ec, gs, lane_e);
        }
      } else {
        epilogue_sub<EPI, TC, MTE, 0, 2>(acc, ep, growW, gcol0, ec, g, lane_e);
        epilogue_sub<EPI, TC, MTE, 2, 2>(acc, ep, growW + 32, gcol0, ec, g, lane_e);
        if constexpr (MTE == 8) {
          epilogue_sub<EPI, TC, MTE, 4, 2>(acc, ep, growW + 64, gcol0, ec, g, lane_e);
          epilogue_sub<EPI, TC, MTE, 6, 2>(acc, ep, growW + 96, gcol0, ec, g, lane_e);
        }
      }
    } else if (acc[0][0][0] == 123.456f) {
      reinterpret_cast<float*>(g.C)[0] = 1.f;
    }
    };   // tile_body
    if constexpr (HALF_OK) {
      if (hf >= 0) tile_body(std::integral_constant<int, MT / 2>{});   // wave-uniform
      else tile_body(std::integral_constant<int, MT>{});
    } else {
      tile_body(std::integral_constant<int, MT>{});
    }
    if (j_next < 0) break;
    j = j_next; tm = tm_next; tn = tn_next; sp = sp_next; hf = hf_next;
    // every wave must be done with its epilogue region before the next K
    // step's DMA overwrites that stage
    __syncthreads();
  }
}

template <int EPI, typename TC, int WM, int WN, int MT>
static int launch_cfg(GemmArgs g, hipStream_t s) {
  constexpr int TBM = WM * MT * 16, TBN = WN * 64;
  constexpr int LDS = 2 * (TBM + TBN) * BK * 2;
  static bool attr_done[MAX_DEVICES] = {};
  int dev = 0;
  if (int rc = ensure_dynamic_lds(&gemm_bf16_kernel<EPI, TC, WM, WN, MT>, LDS, attr_done, "gemm", &dev)) return rc;
  g.tiles_m = (int)(g.M / TBM);
  g.tiles_n = (g.N + TBN - 1) / TBN;
  int64_t grid = (((int64_t)g.tiles_m * g.tiles_n * (g.splits > 1 ? g.splits : 1) + 7) / 8) * 8;
  if (grid >= (1ll << 31)) { set_error("gemm: grid too large"); return -1; }
  // persistent launch: one workgroup per CU for the 128 KiB-LDS tile, two for
  // the 64 KiB one
  int64_t resident = (LDS > 80 * 1024) ? device_cus(dev) : 2 * device_cus(dev);
#ifdef MDLM_GEMM_ABLATION
  if (g.dbg & 8) resident /= 2;   // experiment: half of the CUs
#endif
  if (grid > resident) grid = resident;
  gemm_bf16_kernel<EPI, TC, WM, WN, MT><<<(unsigned)grid, 64 * WM * WN, LDS, s>>>(g);
  MDLM_LAUNCH_CHECK();
  return 0;
}

// tile: 0 = 256x256 when M % 256 == 0 and N >= 256, else 128x128; 1 = 128x128
template <int EPI, typename TC>
static int launch_gemm(const GemmArgs& g, hipStream_t s, int tile = 0) {
  if (g.M % 256 == 0 && g.N >= 256 && tile != 1)
    return launch_cfg<EPI, TC, 2, 4, 8>(g, s);
  return launch_cfg<EPI, TC, 2, 2, 4>(g, s);
}

}  // namespace mdlm

namespace mdlm {
int gemm_pp_dispatch(const GemmArgs& g, int epilogue, int dtype_c, hipStream_t s);   // gemm_pp.hip
}

using namespace mdlm;

extern "C" int mdlm_gemm_bf16(const void* A, int64_t lda, const void* W,
                              const float* bias, void* C, int64_t ldc,
                              int dtype_c, int64_t M, int N, int K, int epilogue,
                              const float* gate, const float* residual,
                              int64_t rows_per_mod, int n_mod, int64_t mod_ld,
                              const float* rot_cos, const float* rot_sin,
                              int n_heads, const int32_t* m_rows_dev,
                              const int32_t* out_sample_map,
                              const int32_t* out_row_map, float* lse_partials_out,
                              int lse_skip_col, int tile, void* stream) {
  if (M == 0) return 0;
  MDLM_CHECK_ARG(A && W && C, "gemm_bf16: null pointer");
  MDLM_CHECK_ARG((tile & ~63) == 0, "gemm_bf16: tile must be 0 (auto), 1 (128x128), optionally | 16 (whole tiles only) | 32 (two-accumulator-set kernel where it applies)");
  MDLM_CHECK_ARG(M >= 0 && N > 0 && K > 0, "gemm_bf16: bad shape");
  MDLM_CHECK_ARG(M % 128 == 0, "gemm_bf16: M=%lld must be a multiple of 128", (long long)M);
  MDLM_CHECK_ARG(K % BK == 0, "gemm_bf16: K=%d must be a multiple of %d", K, BK);
  MDLM_CHECK_ARG(lda >= K && lda % 8 == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0,
                 "gemm_bf16: A/W must be 16-byte aligned with lda %% 8 == 0");
  MDLM_CHECK_ARG(M * (N > K ? N : K) < (1ll << 40), "gemm_bf16: too large");
  MDLM_CHECK_ARG(!bias || ((uintptr_t)bias & 15) == 0, "gemm_bf16: bias must be 16-byte aligned");
  if (M == 0) return 0;
  GemmArgs g;
  g.A = (const bf16_t*)A; g.lda = lda; g.W = (const bf16_t*)W; g.bias = bias;
  g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.gate = gate; g.residual = residual;
  g.rows_per_mod = rows_per_mod > 0 ? rows_per_mod : 1; g.n_mod = n_mod > 0 ? n_mod : 1;
  g.mod_ld = mod_ld; g.out_sample_map = out_sample_map; g.out_row_map = out_row_map;
  g.rot_cos = rot_cos; g.rot_sin = rot_sin; g.n_heads = n_heads; g.m_rows_dev = m_rows_dev;
  g.tiles_m = 0; g.tiles_n = 0; g.dbg = tile & 14;   // ablation builds only
  g.whole_tiles = (tile >> 4) & 1;
  const bool try_pp = (tile & 32) != 0;
  tile &= 1;
  g.ldw = K; g.splits = 1; g.c_split_stride = 0; g.aux = nullptr;
  g.lse_part = reinterpret_cast<float2*>(lse_partials_out); g.lse_skip = lse_skip_col;
  g.lse_ng = (N + 63) / 64;
  MDLM_CHECK_ARG(!lse_partials_out || (epilogue == MDLM_EPI_STORE && dtype_c == MDLM_BF16 &&
                                       ((uintptr_t)lse_partials_out & 7) == 0 && M % 256 == 0),
                 "gemm_bf16: lse partials need the bf16 STORE epilogue, M %% 256 == 0 and an 8-byte aligned buffer");
  hipStream_t s = (hipStream_t)stream;
  // 16-byte row stores: 8 bf16 or 4 fp32 per lane
  const int64_t per16 = dtype_c == MDLM_F32 ? 4 : 8;
  g.vec_ok = (((uintptr_t)C & 15) == 0) && (ldc % per16 == 0) ? 1 : 0;
  switch (epilogue) {
    case MDLM_EPI_STORE:
      MDLM_CHECK_ARG(ldc >= N, "gemm_bf16: ldc < N");
      MDLM_CHECK_ARG(dtype_c == MDLM_F32 || dtype_c == MDLM_BF16, "gemm_bf16: bad dtype_c");
      MDLM_CHECK_ARG(!out_sample_map || g.rows_per_mod % 64 == 0, "gemm_bf16: out_sample_map needs rows_per_mod %% 64 == 0");
      MDLM_CHECK_ARG(!(out_sample_map && out_row_map), "gemm_bf16: out_sample_map and out_row_map are exclusive");
      if (dtype_c == MDLM_F32) return launch_gemm<MDLM_EPI_STORE, float>(g, s, tile);
      return launch_gemm<MDLM_EPI_STORE, bf16_t>(g, s, tile);
    case MDLM_EPI_BIAS_GELU:
      MDLM_CHECK_ARG(ldc >= N && dtype_c == MDLM_BF16, "gemm_bf16: BIAS_GELU writes bf16 with ldc >= N");
      if (try_pp) {
        const int rc = gemm_pp_dispatch(g, epilogue, dtype_c, s);
        if (rc != -2) return rc;
      }
      return launch_gemm<MDLM_EPI_BIAS_GELU, bf16_t>(g, s, tile);
    case MDLM_EPI_GATE_RESIDUAL:
      MDLM_CHECK_ARG(gate && residual && dtype_c == MDLM_F32, "gemm_bf16: GATE_RESIDUAL needs gate, residual and fp32 C");
      MDLM_CHECK_ARG(ldc >= N && N % 4 == 0 && ldc % 4 == 0 && ((uintptr_t)residual & 15) == 0 && ((uintptr_t)gate & 15) == 0 && g.vec_ok,
                     "gemm_bf16: GATE_RESIDUAL needs N, ldc multiples of 4 and 16-byte aligned buffers");
      MDLM_CHECK_ARG(mod_ld % 4 == 0 && (g.n_mod == 1 || (mod_ld >= N && g.rows_per_mod % 64 == 0)),
                     "gemm_bf16: bad gate leading dimension / rows_per_mod %% 64 != 0");
      if (try_pp) {
        const int rc = gemm_pp_dispatch(g, epilogue, dtype_c, s);
        if (rc != -2) return rc;
      }
      return launch_gemm<MDLM_EPI_GATE_RESIDUAL, float>(g, s, tile);
    case MDLM_EPI_QKV_ROTARY:
      MDLM_CHECK_ARG(rot_cos && rot_sin && n_heads > 0 && N == 3 * n_heads * 64,
                     "gemm_bf16: QKV_ROTARY needs head_dim 64 (N == 3*H*64) and rotary tables");
      MDLM_CHECK_ARG(rows_per_mod > 0 && rows_per_mod % 64 == 0 && M % rows_per_mod == 0,
                     "gemm_bf16: QKV_ROTARY needs M %% L == 0 and L %% 64 == 0");
      MDLM_CHECK_ARG(dtype_c == MDLM_BF16 && ((uintptr_t)C & 15) == 0, "gemm_bf16: QKV_ROTARY writes 16-byte aligned bf16");
      return launch_gemm<MDLM_EPI_QKV_ROTARY, bf16_t>(g, s, tile);
    default:
      MDLM_CHECK_ARG(false, "gemm_bf16: unknown epilogue %d", epilogue);
  }
  return -1;
}


// ---- split-K: weight-gradient GEMMs of the training step --------------------
// dW[N_out, K_in] = dY^T X reduces over the TOKEN dimension (65,536 at the LM1B
// batch) into a small output (768x768 = 9 tiles): without a split only a few
// of the 256 CUs would work.  Split s writes the fp32 partial sum over its
// token range into partials[s]; mdlm_splitk_reduce adds them in index order.
namespace mdlm {
__global__ __launch_bounds__(256) void splitk_reduce_kernel(
    const float4* __restrict__ part, float4* __restrict__ out, int64_t n4, int splits,
    int64_t stride4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4 a = part[i];
  for (int s = 1; s < splits; ++s) {
    const float4 b = part[i + s * stride4];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  out[i] = a;
}
}  // namespace mdlm

extern "C" int mdlm_gemm_bf16_splitk(const void* A, int64_t lda, const void* W, int64_t ldw,
                                     float* partials, int64_t M, int N, int64_t K_total,
                                     int splits, void* stream) {
  if (M == 0) return 0;
  MDLM_CHECK_ARG(A && W && partials, "gemm_bf16_splitk: null pointer");
  MDLM_CHECK_ARG(M > 0 && N > 0 && K_total > 0 && splits >= 1 && splits <= 1024, "gemm_bf16_splitk: bad shape");
  MDLM_CHECK_ARG(M % 128 == 0, "gemm_bf16_splitk: M=%lld must be a multiple of 128", (long long)M);
  MDLM_CHECK_ARG(N % 4 == 0, "gemm_bf16_splitk: N must be a multiple of 4");
  MDLM_CHECK_ARG(K_total % ((int64_t)splits * BK) == 0 && K_total / splits < (1ll << 31),
                 "gemm_bf16_splitk: K=%lld must be a multiple of splits*%d", (long long)K_total, BK);
  MDLM_CHECK_ARG(lda >= K_total && ldw >= K_total && lda % 8 == 0 && ldw % 8 == 0 &&
                 ((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)partials & 15) == 0,
                 "gemm_bf16_splitk: operands must be 16-byte aligned with leading dimensions %% 8 == 0");
  GemmArgs g;
  g.A = (const bf16_t*)A; g.lda = lda; g.W = (const bf16_t*)W; g.ldw = ldw; g.bias = nullptr;
  g.C = partials; g.ldc = N; g.M = M; g.N = N; g.K = (int)(K_total / splits);
  g.gate = nullptr; g.residual = nullptr; g.rows_per_mod = 1; g.n_mod = 1; g.mod_ld = 0;
  g.out_sample_map = nullptr; g.out_row_map = nullptr; g.rot_cos = nullptr; g.rot_sin = nullptr;
  g.n_heads = 0; g.m_rows_dev = nullptr; g.tiles_m = 0; g.tiles_n = 0; g.dbg = 0; g.whole_tiles = 0;
  g.splits = splits; g.c_split_stride = M * (int64_t)N; g.vec_ok = 1; g.aux = nullptr;
  g.lse_part = nullptr; g.lse_skip = -1; g.lse_ng = 0;
  return launch_gemm<MDLM_EPI_STORE, float>(g, (hipStream_t)stream);
}

extern "C" int mdlm_splitk_reduce(const float* partials, float* out, int64_t n, int splits,
                                  int64_t split_stride, void* stream) {
  if (n == 0) return 0;
  MDLM_CHECK_ARG(partials && out && n > 0 && splits >= 1, "splitk_reduce: bad args");
  MDLM_CHECK_ARG(n % 4 == 0 && split_stride % 4 == 0 && ((uintptr_t)partials & 15) == 0 && ((uintptr_t)out & 15) == 0,
                 "splitk_reduce: n, split_stride multiples of 4 and 16-byte aligned buffers");
  const int64_t n4 = n / 4;
  MDLM_CHECK_ARG((n4 + 255) / 256 < (1ll << 31), "splitk_reduce: too large");
  splitk_reduce_kernel<<<(unsigned)((n4 + 255) / 256), 256, 0, (hipStream_t)stream>>>(
    (const float4*)partials, (float4*)out, n4, splits, split_stride / 4);
  MDLM_LAUNCH_CHECK();
  return 0;
}


// ---- GELU fused into the MLP GEMMs of the training step ----------------------
extern "C" int mdlm_gemm_bf16_gelu(const void* A, int64_t lda, const void* W, const float* bias,
                                   void* C, void* aux, int64_t ldc, int64_t M, int N, int K,
                                   int mode, void* stream) {
  if (M == 0) return 0;
  MDLM_CHECK_ARG(A && W && C && aux, "gemm_bf16_gelu: null pointer");
  MDLM_CHECK_ARG(mode == 0 || mode == 1, "gemm_bf16_gelu: mode must be 0 (save) or 1 (gradient)");
  MDLM_CHECK_ARG(M > 0 && N > 0 && K > 0 && M % 128 == 0 && K % BK == 0, "gemm_bf16_gelu: M %% 128, K %% 64");
  MDLM_CHECK_ARG(lda >= K && lda % 8 == 0 && ldc >= N && ldc % 8 == 0 &&
                 (((uintptr_t)A | (uintptr_t)W | (uintptr_t)C | (uintptr_t)aux) & 15) == 0,
                 "gemm_bf16_gelu: 16-byte aligned operands, leading dimensions %% 8 == 0");
  MDLM_CHECK_ARG(!bias || ((uintptr_t)bias & 15) == 0, "gemm_bf16_gelu: bias must be 16-byte aligned");
  MDLM_CHECK_ARG(M * (int64_t)(N > K ? N : K) < (1ll << 40), "gemm_bf16_gelu: too large");
  GemmArgs g;
  g.A = (const bf16_t*)A; g.lda = lda; g.W = (const bf16_t*)W; g.ldw = K; g.bias = bias;
  g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.gate = nullptr; g.residual = nullptr;
  g.rows_per_mod = 1; g.n_mod = 1; g.mod_ld = 0; g.out_sample_map = nullptr; g.out_row_map = nullptr;
  g.rot_cos = nullptr; g.rot_sin = nullptr; g.n_heads = 0; g.m_rows_dev = nullptr;
  g.tiles_m = 0; g.tiles_n = 0; g.dbg = 0; g.whole_tiles = 0; g.splits = 1; g.c_split_stride = 0; g.vec_ok = 1;
  g.aux = aux;
  g.lse_part = nullptr; g.lse_skip = -1; g.lse_ng = 0;
  if (mode == 0) return launch_gemm<EPI_BIAS_GELU_SAVE, bf16_t>(g, (hipStream_t)stream);
  return launch_gemm<EPI_MUL_GELU_GRAD, bf16_t>(g, (hipStream_t)stream);
}


// ---- row lse from the vocabulary GEMM's partials ------------------------------
// One workgroup = 64 rows x 16 group slices: thread (slice, row) folds its share
// of the row's ceil(N/64) (max, sum) pairs, the 16 slices meet in LDS, and
// lse[out_row_map[row]] = max + ln(sum) (natural log, what mdlm_subs_ddpm_update
// reads).  Loads are 512 contiguous bytes per wave.
namespace mdlm {
__global__ __launch_bounds__(1024) void lse_combine_kernel(
    const float2* __restrict__ part, int ng, const int32_t* __restrict__ out_row_map,
    const int32_t* __restrict__ n_rows_dev, int64_t rows, float* __restrict__ lse) {
  __shared__ float sm_m[16][64], sm_s[16][64];
  constexpr float L2E = 1.4426950408889634f;
  const int rl = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int64_t R = (int64_t)blockIdx.x * 64 + rl;
  const int64_t live = n_rows_dev ? (int64_t)*n_rows_dev : rows;
  if ((int64_t)blockIdx.x * 64 >= live) return;   // block-uniform
  const float2* p = part + (R >> 8) * (int64_t)ng * 256 + (R & 255);
  float m = -INFINITY, s = 0.f;
  for (int gq = sl; gq < ng; gq += 16) {
    const float2 v = p[(int64_t)gq * 256];
    const float M = fmaxf(m, v.x);
    const float ea = (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((m - M) * L2E);
    const float eb = (v.x == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((v.x - M) * L2E);
    s = s * ea + v.y * eb;
    m = M;
  }
  sm_m[sl][rl] = m; sm_s[sl][rl] = s;
  __syncthreads();
  if (sl == 0 && R < live && R < rows) {
    float M = sm_m[0][rl];
#pragma unroll
    for (int k = 1; k < 16; ++k) M = fmaxf(M, sm_m[k][rl]);
    float S = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k)
      S += (sm_m[k][rl] == -INFINITY) ? 0.f : sm_s[k][rl] * __builtin_amdgcn_exp2f((sm_m[k][rl] - M) * L2E);
    const int64_t dst = out_row_map ? (int64_t)out_row_map[R] : R;
    if (dst >= 0) lse[dst] = M + logf(S);
  }
}
}  // namespace mdlm

extern "C" int64_t mdlm_gemm_lse_partials_workspace_bytes(int64_t M, int N) {
  if (M < 0 || N <= 0 || M % 256) return -1;
  return M * (int64_t)((N + 63) / 64) * 8;
}

extern "C" int mdlm_lse_combine(const float* partials, int64_t M, int N,
                                const int32_t* out_row_map, const int32_t* n_rows_dev,
                                float* lse, void* stream) {
  if (M == 0) return 0;
  MDLM_CHECK_ARG(partials && lse, "lse_combine: null pointer");
  MDLM_CHECK_ARG(M > 0 && M % 256 == 0 && N > 0 && M / 64 < (1ll << 31), "lse_combine: M must be a positive multiple of 256");
  MDLM_CHECK_ARG(((uintptr_t)partials & 7) == 0, "lse_combine: partials must be 8-byte aligned");
  lse_combine_kernel<<<(unsigned)(M / 64), 1024, 0, (hipStream_t)stream>>>(
    reinterpret_cast<const float2*>(partials), (N + 63) / 64, out_row_map, n_rows_dev, M, lse);
  MDLM_LAUNCH_CHECK();
  return 0;
}
