// Training-step kernels: the backward halves of the DiT block's
// row-wise ops, the SUBS-NLL gradient, dropout inside bias_dropout_add_scale,
// column reductions for bias / adaLN gradients, bf16 transposes that put the
// backward GEMM operands K-contiguous for mdlm_gemm_bf16, global-norm clip and
// the fused AdamW + EMA + bf16-weight-copy update.  All HBM-bound.
//
// Reference: models/dit.py:20-34 (bias_dropout_add_scale, training=True),
// 126-134 + 74-78 (LayerNorm + modulate), 220-228 (GELU-tanh MLP),
// diffusion.py:261-277 + 884-916 (SUBS log-probs -> NELBO), 449-459 (AdamW),
// models/ema.py:29-49, configs/config.yaml:69 (gradient_clip_val 1.0).  The
// reference gets every backward from torch autograd.
#include "common.h"
#include <mutex>
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace mdlm {

constexpr int TR_NV = 4;  // float4 per lane per row: D <= 1024

__device__ __forceinline__ float4 bf16x4_to_f32(uint2 v) {
  return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u),
                     __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
}

// ----------------------------------------------------- LayerNorm+modulate bwd
// h = ((x - mu) * rstd * w) * (1 + scale) + shift.  Per row: dn = dh * w * (1 +
// scale); dx = rstd * (dn - mean(dn) - n * mean(dn * n)), added to dx_inout
// (the residual-stream gradient).  Column partial sums over the 32 rows of the
// workgroup: [0] dw = dh*n*(1+scale), [1] dscale = dh*n*w, [2] dshift = dh.
constexpr int LNB_ROWS = 32;
// NV = float4 columns per lane: 3 for D <= 768 (DiT-small: no idle register sets), else 4
template <int NV>
__global__ __launch_bounds__(256) void ln_modulate_bwd_kernel(
    const bf16_t* __restrict__ dh, const float* __restrict__ x,
    const float* __restrict__ ln_w, const float* __restrict__ scale,
    float* __restrict__ dx_inout, float* __restrict__ partial, int64_t rows,
    int D, int64_t rows_per_mod, int n_mod, int64_t mod_ld, int accumulate) {
  extern __shared__ float lnb_smem[];  // [4 waves][3][D]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nv = D / 4;
  const int64_t row0 = (int64_t)blockIdx.x * LNB_ROWS;
  const int64_t mrow = (row0 / rows_per_mod) % n_mod;
  const float4* w4 = reinterpret_cast<const float4*>(ln_w);
  const float4* sc4 = reinterpret_cast<const float4*>(scale + mrow * mod_ld);
  float4 w[NV], g1[NV], pw[NV], ps[NV], pb[NV];
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + i * 64;
    w[i] = c < nv ? w4[c] : z4;
    const float4 s = c < nv ? sc4[c] : z4;
    g1[i] = make_float4(1.f + s.x, 1.f + s.y, 1.f + s.z, 1.f + s.w);
    pw[i] = z4; ps[i] = z4; pb[i] = z4;
  }
  for (int rr = wid; rr < LNB_ROWS; rr += 4) {
    const int64_t r = row0 + rr;
    if (r >= rows) break;
    const float4* xr = reinterpret_cast<const float4*>(x + r * D);
    const uint2* dr = reinterpret_cast<const uint2*>(dh + r * D);
    float4 v[NV], d[NV], acc_in[NV];
    float4* o = reinterpret_cast<float4*>(dx_inout + r * D);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + i * 64;
      v[i] = c < nv ? xr[c] : z4;
      d[i] = c < nv ? bf16x4_to_f32(dr[c]) : z4;
      // the running residual gradient is fetched with the row, not after the
      // three wave reductions that precede its use
      acc_in[i] = (accumulate && c < nv) ? o[c] : z4;
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + i * 64;
      if (c < nv) {
        v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
        q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + 1e-5f);
    float s1 = 0.f, s2 = 0.f;
    float4 dn[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      v[i].x *= rstd; v[i].y *= rstd; v[i].z *= rstd; v[i].w *= rstd;  // n
      dn[i] = make_float4(d[i].x * w[i].x * g1[i].x, d[i].y * w[i].y * g1[i].y,
                          d[i].z * w[i].z * g1[i].z, d[i].w * w[i].w * g1[i].w);
      s1 += (dn[i].x + dn[i].y) + (dn[i].z + dn[i].w);
      s2 += (dn[i].x * v[i].x + dn[i].y * v[i].y) + (dn[i].z * v[i].z + dn[i].w * v[i].w);
      pw[i].x += d[i].x * v[i].x * g1[i].x; pw[i].y += d[i].y * v[i].y * g1[i].y;
      pw[i].z += d[i].z * v[i].z * g1[i].z; pw[i].w += d[i].w * v[i].w * g1[i].w;
      ps[i].x += d[i].x * v[i].x * w[i].x; ps[i].y += d[i].y * v[i].y * w[i].y;
      ps[i].z += d[i].z * v[i].z * w[i].z; ps[i].w += d[i].w * v[i].w * w[i].w;
      pb[i].x += d[i].x; pb[i].y += d[i].y; pb[i].z += d[i].z; pb[i].w += d[i].w;
    }
    s1 = wave_sum(s1) / (float)D;
    s2 = wave_sum(s2) / (float)D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + i * 64;
      if (c < nv) {
        float4 r4 = make_float4(rstd * (dn[i].x - s1 - v[i].x * s2), rstd * (dn[i].y - s1 - v[i].y * s2),
                                rstd * (dn[i].z - s1 - v[i].z * s2), rstd * (dn[i].w - s1 - v[i].w * s2));
        const float4 a = acc_in[i];
        r4.x += a.x; r4.y += a.y; r4.z += a.z; r4.w += a.w;
        o[c] = r4;
      }
    }
  }
  float4* sm = reinterpret_cast<float4*>(lnb_smem);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      sm[(wid * 3 + 0) * nv + c] = pw[i];
      sm[(wid * 3 + 1) * nv + c] = ps[i];
      sm[(wid * 3 + 2) * nv + c] = pb[i];
    }
  }
  __syncthreads();
  float4* po = reinterpret_cast<float4*>(partial + (int64_t)blockIdx.x * 3 * D);
  for (int i = threadIdx.x; i < 3 * nv; i += 256) {
    const float4 a = sm[i], b = sm[3 * nv + i], c = sm[6 * nv + i], e = sm[9 * nv + i];
    po[i] = make_float4((a.x + b.x) + (c.x + e.x), (a.y + b.y) + (c.y + e.y),
                        (a.z + b.z) + (c.z + e.z), (a.w + b.w) + (c.w + e.w));
  }
}

// out[o][c] = sum_{i < per_out} partial[o * per_out + i][c]   (fixed order:
// 16 interleaved row groups summed serially, then added up in group order)
__global__ __launch_bounds__(1024) void colsum_reduce_kernel(const float* __restrict__ partial,
                                                             float* __restrict__ out,
                                                             int64_t C, int per_out) {
  __shared__ float red[16][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int64_t c = (int64_t)blockIdx.x * 64 + tx;
  const int64_t o = blockIdx.y;
  float s = 0.f;
  if (c < C) {
    const float* p = partial + o * per_out * C + c;
#pragma unroll 8
    for (int i = ty; i < per_out; i += 16) s += p[(int64_t)i * C];
  }
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && c < C) {
    float t = red[0][tx];
#pragma unroll
    for (int k = 1; k < 16; ++k) t += red[k][tx];
    out[o * C + c] = t;
  }
}

// column sums of a bf16 matrix, stage 1: 128 rows per workgroup row-block
constexpr int CS_ROWS = 128;
__global__ __launch_bounds__(256) void colsum_bf16_partial_kernel(
    const bf16_t* __restrict__ x, int64_t ld, float* __restrict__ partial, int64_t rows, int N) {
  const int c2 = blockIdx.x * 256 + threadIdx.x;  // column pair
  if (c2 * 2 >= N) return;
  const int64_t r0 = (int64_t)blockIdx.y * CS_ROWS;
  const int64_t r1 = r0 + CS_ROWS < rows ? r0 + CS_ROWS : rows;
  float a = 0.f, b = 0.f;
  for (int64_t r = r0; r < r1; ++r) {
    const unsigned v = *reinterpret_cast<const unsigned*>(x + r * ld + c2 * 2);
    a += __uint_as_float(v << 16);
    b += __uint_as_float(v & 0xffff0000u);
  }
  float* po = partial + (int64_t)blockIdx.y * N + c2 * 2;
  po[0] = a;
  if (c2 * 2 + 1 < N) po[1] = b;
}

// ------------------------------------------- bias_dropout_add_scale, training
// out = residual + gate * dropout(y, p)  (dit.py:20-34; y already has its bias).
// The keep mask of element (r, c) is Philox(seed, offset + r*D/4 + c/4)[c%4]
// >= p, i.e. the layout of mdlm_philox_uniform(rows, D): backward regenerates
// it instead of storing it.
__device__ __forceinline__ float4 keep_scale(uint64_t seed, uint64_t ctr, float p, float inv_keep) {
  if (p <= 0.f) return make_float4(1.f, 1.f, 1.f, 1.f);
  const uint4 rn = Philox::gen(seed, ctr);
  return make_float4(Philox::u01(rn.x) >= p ? inv_keep : 0.f, Philox::u01(rn.y) >= p ? inv_keep : 0.f,
                     Philox::u01(rn.z) >= p ? inv_keep : 0.f, Philox::u01(rn.w) >= p ? inv_keep : 0.f);
}

__global__ __launch_bounds__(256) void dropout_gate_residual_fwd_kernel(
    const bf16_t* __restrict__ y, const float* __restrict__ gate, const float* __restrict__ residual,
    float* __restrict__ out, int64_t rows, int D, int64_t rows_per_mod, int n_mod, int64_t mod_ld,
    float p, uint64_t seed, uint64_t offset) {
  const int nv = D / 4;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * nv) return;
  const int64_t r = i / nv;
  const int c = (int)(i - r * nv);
  const int64_t mrow = (r / rows_per_mod) % n_mod;
  const float4 yv = bf16x4_to_f32(reinterpret_cast<const uint2*>(y)[i]);
  const float4 g = reinterpret_cast<const float4*>(gate + mrow * mod_ld)[c];
  const float4 rs = reinterpret_cast<const float4*>(residual)[i];
  const float4 k = keep_scale(seed, offset + (uint64_t)i, p, 1.f / (1.f - p));
  reinterpret_cast<float4*>(out)[i] = make_float4(rs.x + g.x * (yv.x * k.x), rs.y + g.y * (yv.y * k.y),
                                                  rs.z + g.z * (yv.z * k.z), rs.w + g.w * (yv.w * k.w));
}

// dy = dout * gate * keep/(1-p) (bf16); partial[blk][D] = sum_rows dout * dropout(y)
constexpr int DGR_ROWS = 32;
__global__ __launch_bounds__(256) void dropout_gate_residual_bwd_kernel(
    const float* __restrict__ dout, const bf16_t* __restrict__ y, const float* __restrict__ gate,
    bf16_t* __restrict__ dy, float* __restrict__ partial, int64_t rows, int D,
    int64_t rows_per_mod, int n_mod, int64_t mod_ld, float p, uint64_t seed, uint64_t offset) {
  const int nv = D / 4;
  const int c = threadIdx.x;
  if (c >= nv) return;
  const int64_t row0 = (int64_t)blockIdx.x * DGR_ROWS;
  const int64_t mrow = (row0 / rows_per_mod) % n_mod;
  const float4 g = reinterpret_cast<const float4*>(gate + mrow * mod_ld)[c];
  const float inv_keep = 1.f / (1.f - p);
  float4 pg = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int rr = 0; rr < DGR_ROWS; ++rr) {
    const int64_t r = row0 + rr;
    if (r >= rows) break;
    const int64_t i = r * nv + c;
    const float4 d = reinterpret_cast<const float4*>(dout)[i];
    const float4 yv = bf16x4_to_f32(reinterpret_cast<const uint2*>(y)[i]);
    const float4 k = keep_scale(seed, offset + (uint64_t)i, p, inv_keep);
    reinterpret_cast<uint2*>(dy)[i] = make_uint2(pack_bf16x2(d.x * g.x * k.x, d.y * g.y * k.y),
                                                 pack_bf16x2(d.z * g.z * k.z, d.w * g.w * k.w));
    pg.x += d.x * (yv.x * k.x); pg.y += d.y * (yv.y * k.y);
    pg.z += d.z * (yv.z * k.z); pg.w += d.w * (yv.w * k.w);
  }
  reinterpret_cast<float4*>(partial + (int64_t)blockIdx.x * D)[c] = pg;
}

// ------------------------------------------------------------------ GELU(tanh)
__device__ __forceinline__ float gelu_tanh_f(float x) {
  const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
  return 0.5f * x * (1.f + tanhf(u));
}
__device__ __forceinline__ float gelu_tanh_grad(float x) {
  const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
  const float th = tanhf(u);
  return 0.5f * (1.f + th) + 0.5f * x * (1.f - th * th) * 0.7978845608028654f * (1.f + 3.f * 0.044715f * x * x);
}
__device__ __forceinline__ void unpack8(uint4 v, float* f) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const uint4* __restrict__ pre, uint4* __restrict__ act, int64_t n8) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  float f[8];
  unpack8(pre[i], f);
#pragma unroll
  for (int k = 0; k < 8; ++k) f[k] = gelu_tanh_f(f[k]);
  act[i] = make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
}
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const uint4* __restrict__ dact, const uint4* __restrict__ pre,
                                                       uint4* __restrict__ dpre, int64_t n8) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  float f[8], d[8];
  unpack8(pre[i], f);
  unpack8(dact[i], d);
#pragma unroll
  for (int k = 0; k < 8; ++k) d[k] *= gelu_tanh_grad(f[k]);
  dpre[i] = make_uint4(pack_bf16x2(d[0], d[1]), pack_bf16x2(d[2], d[3]), pack_bf16x2(d[4], d[5]), pack_bf16x2(d[6], d[7]));
}

// ------------------------------------------------------ batched bf16 transpose
// out[b][c][r] = in[b][r][c]; 64x64 tiles through LDS
__global__ __launch_bounds__(256) void transpose_bf16_kernel(
    const bf16_t* __restrict__ in, int64_t ld_in, int64_t bs_in, bf16_t* __restrict__ out,
    int64_t ld_out, int64_t bs_out, int R, int C) {
  __shared__ bf16_t tile[64][66];
  const bf16_t* ib = in + (int64_t)blockIdx.z * bs_in;
  bf16_t* ob = out + (int64_t)blockIdx.z * bs_out;
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < R && c < C) ? ib[(int64_t)r * ld_in + c] : (bf16_t)0;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < C && r < R) ob[(int64_t)c * ld_out + r] = tile[tx][i];
  }
}

// Fast path (R, C, leading dimensions multiples of 8, 16-byte aligned): 16-byte
// loads of 8 columns, scattered into a TRANSPOSED LDS tile [64 c][64 r] whose
// 16-byte chunk index is XOR-swizzled with (c >> 3) (the eight chunks a wave
// writes per store instruction then fall into eight different bank groups),
// read back as whole 16-byte chunks: eight lanes store 128 contiguous bytes
// of one output row.
//
// colsum_partial (optional, batch 1): partial[blockIdx.y][c] = sum of the tile's
// 64 rows of column c in fp32 (the bias gradient of the Linear whose dY is being
// transposed), added up by mdlm_colsum_reduce: the separate column-sum pass over
// dY is then not needed.
__global__ __launch_bounds__(256) void transpose_bf16_vec_kernel(
    const bf16_t* __restrict__ in, int64_t ld_in, int64_t bs_in, bf16_t* __restrict__ out,
    int64_t ld_out, int64_t bs_out, int R, int C, float* __restrict__ colsum_partial) {
  __shared__ __attribute__((aligned(16))) bf16_t tile[64 * 64];
  const bf16_t* ib = in + (int64_t)blockIdx.z * bs_in;
  bf16_t* ob = out + (int64_t)blockIdx.z * bs_out;
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int k = lane & 7;
  uint4 v[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int r = r0 + w * 16 + it * 8 + (lane >> 3), c = c0 + k * 8;
    v[it] = (r < R && c < C) ? *reinterpret_cast<const uint4*>(ib + (int64_t)r * ld_in + c)
                             : make_uint4(0, 0, 0, 0);
  }
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int rl = w * 16 + it * 8 + (lane >> 3);
    const unsigned u[4] = {v[it].x, v[it].y, v[it].z, v[it].w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int cl = k * 8 + j;
      const bf16_t e = (bf16_t)(j & 1 ? u[j >> 1] >> 16 : u[j >> 1] & 0xffffu);
      tile[cl * 64 + (((rl >> 3) ^ k) << 3) + (rl & 7)] = e;
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int cl = w * 16 + it * 8 + (lane >> 3), rc = lane & 7;
    const int c = c0 + cl, r = r0 + rc * 8;
    const uint4 o = *reinterpret_cast<const uint4*>(tile + cl * 64 + ((rc ^ ((cl >> 3) & 7)) << 3));
    if (c < C && r < R) *reinterpret_cast<uint4*>(ob + (int64_t)c * ld_out + r) = o;
    if (colsum_partial) {   // rows past R were loaded as zeros
      const unsigned u[4] = {o.x, o.y, o.z, o.w};
      float sm = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        sm += __uint_as_float(u[j] << 16) + __uint_as_float(u[j] & 0xffff0000u);
      sm += __shfl_xor(sm, 1); sm += __shfl_xor(sm, 2); sm += __shfl_xor(sm, 4);
      if (rc == 0 && c < C) colsum_partial[(int64_t)blockIdx.y * C + c] = sm;
    }
  }
}

// ----------------------------------------- d(qkv) from dq, dk, dv (head-major)
// Inverse of the QKV epilogue: un-rotate dq / dk (dit.py:112-115 transposed:
// dx1 = do1*c + do2*s, dx2 = -do1*s + do2*c) and lay the three gradients out
// as rows of the (M, 3*H*64) matrix the QKV GEMM produced ('three h d').
__global__ __launch_bounds__(256) void qkv_grad_pack_kernel(
    const bf16_t* __restrict__ dq, const bf16_t* __restrict__ dk, const bf16_t* __restrict__ dv,
    const float* __restrict__ cosb, const float* __restrict__ sinb, bf16_t* __restrict__ dqkv,
    int64_t M, int H, int L) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= M * H * 4) return;
  const int j = (int)(i & 3);
  const int64_t mh = i >> 2;
  const int h = (int)(mh % H);
  const int64_t m = mh / H;
  const int64_t seq = m / L;
  const int l = (int)(m - seq * L);
  const int64_t src = ((seq * H + h) * L + l) * 64;
  const int64_t D3 = (int64_t)3 * H * 64;
  bf16_t* dst = dqkv + m * D3 + h * 64;
  float c[8], s[8];
  *reinterpret_cast<float4*>(c) = *reinterpret_cast<const float4*>(cosb + l * 32 + j * 8);
  *reinterpret_cast<float4*>(c + 4) = *reinterpret_cast<const float4*>(cosb + l * 32 + j * 8 + 4);
  *reinterpret_cast<float4*>(s) = *reinterpret_cast<const float4*>(sinb + l * 32 + j * 8);
  *reinterpret_cast<float4*>(s + 4) = *reinterpret_cast<const float4*>(sinb + l * 32 + j * 8 + 4);
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    const bf16_t* g = (which == 0 ? dq : dk) + src;
    float a[8], b[8], o1[8], o2[8];
    unpack8(*reinterpret_cast<const uint4*>(g + j * 8), a);
    unpack8(*reinterpret_cast<const uint4*>(g + 32 + j * 8), b);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      o1[k] = a[k] * c[k] + b[k] * s[k];
      o2[k] = b[k] * c[k] - a[k] * s[k];
    }
    bf16_t* d = dst + (int64_t)which * H * 64;
    *reinterpret_cast<uint4*>(d + j * 8) = make_uint4(pack_bf16x2(o1[0], o1[1]), pack_bf16x2(o1[2], o1[3]),
                                                      pack_bf16x2(o1[4], o1[5]), pack_bf16x2(o1[6], o1[7]));
    *reinterpret_cast<uint4*>(d + 32 + j * 8) = make_uint4(pack_bf16x2(o2[0], o2[1]), pack_bf16x2(o2[2], o2[3]),
                                                           pack_bf16x2(o2[4], o2[5]), pack_bf16x2(o2[6], o2[7]));
  }
  bf16_t* d = dst + (int64_t)2 * H * 64;
  *reinterpret_cast<uint4*>(d + j * 8) = *reinterpret_cast<const uint4*>(dv + src + j * 8);
  *reinterpret_cast<uint4*>(d + 32 + j * 8) = *reinterpret_cast<const uint4*>(dv + src + 32 + j * 8);
}

// ---------------------------------------------------------- embedding gradient
// dtable[idx[r]] += dx[r].  About half of the rows of a training batch hold
// the mask token (q_xt): its table row would take ~rows/2 atomic adds per
// element.  Rows whose index is `hot` are therefore summed per 128-row block
// into hot_partial[block][D] (fixed order; mdlm_colsum_reduce then writes the
// hot row), every other row goes through fp32 atomics.
constexpr int EG_ROWS = 128;
__global__ __launch_bounds__(256) void embed_grad_kernel(const float* __restrict__ dx, const int64_t* __restrict__ idx,
                                                         float* __restrict__ dtable, int64_t rows, int D,
                                                         int vocab, int hot, float* __restrict__ hot_partial) {
  __shared__ int64_t sidx[EG_ROWS];
  const int nv = D / 4;
  const int64_t r0 = (int64_t)blockIdx.x * EG_ROWS;
  const int nr = (int)(rows - r0 < EG_ROWS ? rows - r0 : EG_ROWS);
  if (threadIdx.x < nr) sidx[threadIdx.x] = idx[r0 + threadIdx.x];
  __syncthreads();
  for (int c = threadIdx.x; c < nv; c += 256) {
    float4 h = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = 0; i < nr; ++i) {
      const float4 v = reinterpret_cast<const float4*>(dx + (r0 + i) * D)[c];
      const int64_t t = sidx[i];
      if (t == hot) {
        h.x += v.x; h.y += v.y; h.z += v.z; h.w += v.w;
      } else if (t >= 0 && t < vocab) {   // never write outside the table (ids are validated on the host)
        float* d = dtable + t * D + c * 4;
        unsafeAtomicAdd(d + 0, v.x); unsafeAtomicAdd(d + 1, v.y);
        unsafeAtomicAdd(d + 2, v.z); unsafeAtomicAdd(d + 3, v.w);
      }
    }
    if (hot_partial) reinterpret_cast<float4*>(hot_partial + (int64_t)blockIdx.x * D)[c] = h;
  }
}

// ---------------------------------------------------------------- SUBS-NLL bwd
// loss = sum_rows w_r * -(logit[x0] - lse_{v != mask}) for rows with xt == mask
// (diffusion.py:261-277, 884-916): dlogit[v] = w_r * (softmax_v - [v == x0]),
// 0 in the mask column, 0 for unmasked rows and in the padding columns.
template <typename T>
__global__ __launch_bounds__(256) void subs_nll_bwd_kernel(
    const T* __restrict__ logits, int64_t ld, const int64_t* __restrict__ xt,
    const int64_t* __restrict__ x0, const float* __restrict__ row_w, bf16_t* __restrict__ dl,
    int64_t ldo, int V, int mask) {
  __shared__ float red[8];
  const int64_t r = blockIdx.x;
  const int tid = threadIdx.x;
  bf16_t* o = dl + r * ldo;
  const float w = row_w[r];
  if (xt[r] != mask || w == 0.f) {
    for (int64_t c = (int64_t)tid * 8; c < ldo; c += 256 * 8)
      *reinterpret_cast<uint4*>(o + c) = make_uint4(0, 0, 0, 0);
    return;
  }
  const T* lr = logits + r * ld;
  float mx = -INFINITY;
  for (int c = tid; c < V; c += 256)
    if (c != mask) mx = fmaxf(mx, elem<T>::load(lr + c));
  mx = wave_max(mx);
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float s = 0.f;
  for (int c = tid; c < V; c += 256)
    if (c != mask) s += __expf(elem<T>::load(lr + c) - mx);
  s = wave_sum(s);
  if ((tid & 63) == 0) red[4 + (tid >> 6)] = s;
  __syncthreads();
  s = (red[4] + red[5]) + (red[6] + red[7]);
  const float ws = w / s;
  const int tgt = (int)x0[r];
  for (int64_t c0 = (int64_t)tid * 8; c0 < ldo; c0 += 256 * 8) {
    float f[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = (int)c0 + k;
      float g = 0.f;
      if (c < V && c != mask) {
        g = ws * __expf(elem<T>::load(lr + c) - mx);
        if (c == tgt) g -= w;
      }
      f[k] = g;
    }
    *reinterpret_cast<uint4*>(o + c0) = make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]),
                                                   pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
  }
}

// bf16 rows with 16-byte aligned, 8-element padded leading dimensions: one
// online (max, sum) pass over 16-byte loads, then the row again (an L2 hit: the
// same workgroup has just read its 60-100 KB) to form the gradient.
__global__ __launch_bounds__(256) void subs_nll_bwd_vec_kernel(
    const bf16_t* __restrict__ logits, int64_t ld, const int64_t* __restrict__ xt,
    const int64_t* __restrict__ x0, const float* __restrict__ row_w, bf16_t* __restrict__ dl,
    int64_t ldo, int V, int mask) {
  __shared__ float red[8];
  const int64_t r = blockIdx.x;
  const int tid = threadIdx.x;
  bf16_t* o = dl + r * ldo;
  const float w = row_w[r];
  if (xt[r] != mask || w == 0.f) {
    for (int64_t c = (int64_t)tid * 8; c < ldo; c += 256 * 8)
      *reinterpret_cast<uint4*>(o + c) = make_uint4(0, 0, 0, 0);
    return;
  }
  const bf16_t* lr = logits + r * ld;
  const int V8 = (V + 7) & ~7;
  auto unpack = [&](int c0, float (&f)[8]) {
    const uint4 q = *reinterpret_cast<const uint4*>(lr + c0);
    const unsigned u[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      f[2 * k] = __uint_as_float(u[k] << 16);
      f[2 * k + 1] = __uint_as_float(u[k] & 0xffff0000u);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (c0 + k >= V || c0 + k == mask) f[k] = -INFINITY;
  };
  constexpr float LOG2E = 1.4426950408889634f;
  float mx = -INFINITY, sm = 0.f;
  for (int c0 = tid * 8; c0 < V8; c0 += 256 * 8) {
    float f[8];
    unpack(c0, f);
    float m8 = f[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) m8 = fmaxf(m8, f[k]);
    if (m8 > mx) { sm *= __builtin_amdgcn_exp2f((mx - m8) * LOG2E); mx = m8; }
    if (mx > -INFINITY) {
#pragma unroll
      for (int k = 0; k < 8; ++k) sm += __builtin_amdgcn_exp2f((f[k] - mx) * LOG2E);
    }
  }
  const float wm = wave_max(mx);
  if (mx > -INFINITY) sm *= __builtin_amdgcn_exp2f((mx - wm) * LOG2E);
  sm = wave_sum(sm);
  if ((tid & 63) == 0) { red[tid >> 6] = wm; red[4 + (tid >> 6)] = sm; }
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (red[k] > -INFINITY) s += red[4 + k] * __builtin_amdgcn_exp2f((red[k] - mx) * LOG2E);
  const float ws = w / s;
  const int tgt = (int)x0[r];
  for (int64_t c0 = (int64_t)tid * 8; c0 < ldo; c0 += 256 * 8) {
    float f[8];
    if (c0 < V8) {
      unpack((int)c0, f);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float g = ws * __builtin_amdgcn_exp2f((f[k] - mx) * LOG2E);   // exp2(-inf) = 0
        if ((int)c0 + k == tgt) g -= w;
        f[k] = g;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) f[k] = 0.f;
    }
    *reinterpret_cast<uint4*>(o + c0) = make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]),
                                                   pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
  }
}

// ------------------------------------------------- global norm, AdamW + EMA
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ partial) {
  __shared__ float red[4];
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += g[i] * g[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ partial, int n, float* __restrict__ out) {
  __shared__ double red[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)partial[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (float)((red[0] + red[1]) + (red[2] + red[3]));
}

// torch.optim.AdamW single-tensor step (diffusion.py:449-459) on flat buffers,
// with the clip coefficient of clip_grad_norm_ taken from the device-side sum
// of squares, the EMA update of models/ema.py:44-49 and the bf16 copy of the
// new weights that the next forward's GEMMs read.
__global__ __launch_bounds__(256) void adamw_ema_kernel(
    float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
    float* __restrict__ ema, bf16_t* __restrict__ w16, int64_t n, float lr, float b1, float b2,
    float eps, float wd, float bc1, float bc2_sqrt, float ema_omd, const float* __restrict__ sumsq,
    float max_norm, float gscale) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float coef = gscale;
  if (sumsq && max_norm > 0.f) {
    const float total = sqrtf(sumsq[0]) * gscale;
    coef *= fminf(1.f, max_norm / (total + 1e-6f));
  }
  const float gi = g[i] * coef;
  float pi = p[i] * (1.f - lr * wd);
  const float mi = m[i] + (gi - m[i]) * (1.f - b1);     // lerp_
  const float vi = v[i] * b2 + (1.f - b2) * gi * gi;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  pi -= (lr / bc1) * (mi / denom);
  p[i] = pi; m[i] = mi; v[i] = vi;
  if (ema) { const float e = ema[i]; ema[i] = e - ema_omd * (e - pi); }
  if (w16) w16[i] = f32_to_bf16(pi);
}

}  // namespace mdlm

using namespace mdlm;
#define ST(s) ((hipStream_t)(s))

extern "C" int mdlm_ln_modulate_bwd(const void* dh_bf16, const float* x, const float* ln_w,
                                    const float* scale, float* dx_inout, float* partial,
                                    int64_t rows, int D, int64_t rows_per_mod, int n_mod,
                                    int64_t mod_ld, int accumulate, void* stream) {
  if (rows == 0) return 0;
  MDLM_CHECK_ARG(dh_bf16 && x && ln_w && scale && dx_inout && partial, "ln_modulate_bwd: null pointer");
  MDLM_CHECK_ARG(rows > 0 && D > 0 && D % 4 == 0 && D <= TR_NV * 256, "ln_modulate_bwd: D must be a multiple of 4, <= 1024");
  MDLM_CHECK_ARG(n_mod >= 1 && rows_per_mod >= 1 && (n_mod == 1 || rows_per_mod % LNB_ROWS == 0),
                 "ln_modulate_bwd: rows_per_mod must be a multiple of %d", LNB_ROWS);
  MDLM_CHECK_ARG((((uintptr_t)dh_bf16 | (uintptr_t)x | (uintptr_t)ln_w | (uintptr_t)scale | (uintptr_t)dx_inout | (uintptr_t)partial) & 15) == 0 && mod_ld % 4 == 0,
                 "ln_modulate_bwd: 16-byte alignment");
  const unsigned blocks = (unsigned)((rows + LNB_ROWS - 1) / LNB_ROWS);
  if (D <= 768)
    ln_modulate_bwd_kernel<3><<<blocks, 256, 4 * 3 * D * sizeof(float), ST(stream)>>>(
      (const bf16_t*)dh_bf16, x, ln_w, scale, dx_inout, partial, rows, D, rows_per_mod, n_mod, mod_ld, accumulate);
  else
    ln_modulate_bwd_kernel<4><<<blocks, 256, 4 * 3 * D * sizeof(float), ST(stream)>>>(
      (const bf16_t*)dh_bf16, x, ln_w, scale, dx_inout, partial, rows, D, rows_per_mod, n_mod, mod_ld, accumulate);
  MDLM_LAUNCH_CHECK();
  return 0;
}

extern "C" int mdlm_colsum_reduce(const float* partial, float* out, int64_t C, int per_out,
                                  int n_out, void* stream) {
  if (C == 0 || n_out == 0) return 0;
  MDLM_CHECK_ARG(partial && out && C > 0 && per_out > 0 && n_out > 0 && n_out < 65536, "colsum_reduce: bad args");
  colsum_reduce_kernel<<<dim3((unsigned)((C + 63) / 64), (unsigned)n_out), 1024, 0, ST(stream)>>>(partial, out, C, per_out);
  MDLM_LAUNCH_CHECK();
  return 0;
}

extern "C" int mdlm_colsum_bf16_partial(const void* x_bf16, int64_t ld, float* partial,
                                        int64_t rows, int N, void* stream) {
  if (rows == 0 || N == 0) return 0;
  MDLM_CHECK_ARG(x_bf16 && partial && rows > 0 && N > 0 && ld >= N && ld % 2 == 0 && ((uintptr_t)x_bf16 & 3) == 0,
                 "colsum_bf16_partial: bad args");
  const int64_t rb = (rows + CS_ROWS - 1) / CS_ROWS;
  MDLM_CHECK_ARG(rb < 65536, "colsum_bf16_partial: too many rows");
  colsum_bf16_partial_kernel<<<dim3((unsigned)((N + 511) / 512), (unsigned)rb), 256, 0, ST(stream)>>>(
    (const bf16_t*)x_bf16, ld, partial, rows, N);
  MDLM_LAUNCH_CHECK();
  return 0;
}

static int dgr_check(const char* who, int64_t rows, int D, int64_t rows_per_mod, int n_mod, int64_t mod_ld, float p) {
  MDLM_CHECK_ARG(rows > 0 && D > 0 && D % 4 == 0 && D <= 1024, "%s: D must be a multiple of 4, <= 1024", who);
  MDLM_CHECK_ARG(n_mod >= 1 && rows_per_mod >= 1 && mod_ld % 4 == 0, "%s: bad modulation layout", who);
  MDLM_CHECK_ARG(p >= 0.f && p < 1.f, "%s: dropout p must be in [0, 1)", who);
  return 0;
}

extern "C" int mdlm_dropout_gate_residual_fwd(const void* y_bf16, const float* gate,
                                              const float* residual, float* out, int64_t rows,
                                              int D, int64_t rows_per_mod, int n_mod,
                                              int64_t mod_ld, float p, uint64_t seed,
                                              uint64_t offset, void* stream) {
  if (rows == 0) return 0;
  MDLM_CHECK_ARG(y_bf16 && gate && residual && out, "dropout_gate_residual_fwd: null pointer");
  if (dgr_check("dropout_gate_residual_fwd", rows, D, rows_per_mod, n_mod, mod_ld, p)) return -1;
  MDLM_CHECK_ARG((((uintptr_t)y_bf16 | (uintptr_t)gate | (uintptr_t)residual | (uintptr_t)out) & 15) == 0, "dropout_gate_residual_fwd: 16-byte alignment");
  const int64_t n = rows * (D / 4);
  dropout_gate_residual_fwd_kernel<<<(unsigned)((n + 255) / 256), 256, 0, ST(stream)>>>(
    (const bf16_t*)y_bf16, gate, residual, out, rows, D, rows_per_mod, n_mod, mod_ld, p, seed, offset);
  MDLM_LAUNCH_CHECK();
  return 0;
}

extern "C" int mdlm_dropout_gate_residual_bwd(const float* dout, const void* y_bf16,
                                              const float* gate, void* dy_bf16, float* partial,
                                              int64_t rows, int D, int64_t rows_per_mod,
                                              int n_mod, int64_t mod_ld, float p, uint64_t seed,
                                              uint64_t offset, void* stream) {
  if (rows == 0) return 0;
  MDLM_CHECK_ARG(dout && y_bf16 && gate && dy_bf16 && partial, "dropout_gate_residual_bwd: null pointer");
  if (dgr_check("dropout_gate_residual_bwd", rows, D, rows_per_mod, n_mod, mod_ld, p)) return -1;
  MDLM_CHECK_ARG(n_mod == 1 || rows_per_mod % DGR_ROWS == 0, "dropout_gate_residual_bwd: rows_per_mod must be a multiple of %d", DGR_ROWS);
  MDLM_CHECK_ARG((((uintptr_t)dout | (uintptr_t)y_bf16 | (uintptr_t)gate | (uintptr_t)dy_bf16 | (uintptr_t)partial) & 15) == 0, "dropout_gate_residual_bwd: 16-byte alignment");
  dropout_gate_residual_bwd_kernel<<<(unsigned)((rows + DGR_ROWS - 1) / DGR_ROWS), 256, 0, ST(stream)>>>(
    dout, (const bf16_t*)y_bf16, gate, (bf16_t*)dy_bf16, partial, rows, D, rows_per_mod, n_mod, mod_ld, p, seed, offset);
  MDLM_LAUNCH_CHECK();
  return 0;
}

extern "C" int mdlm_gelu_fwd(const void* pre_bf16, void* act_bf16, int64_t n, void* stream) {
  if (n == 0) return 0;
  MDLM_CHECK_ARG(pre_bf16 && act_bf16 && n > 0 && n % 8 == 0 && (((uintptr_t)pre_bf16 | (uintptr_t)act_bf16) & 15) == 0, "gelu_fwd: n %% 8 and 16-byte alignment");
  gelu_fwd_kernel<<<(unsigned)((n / 8 + 255) / 256), 256, 0, ST(stream)>>>((const uint4*)pre_bf16, (uint4*)act_bf16, n / 8);
  MDLM_LAUNCH_CHECK();
  return 0;
}

extern "C" int mdlm_gelu_bwd(const void* dact_bf16, const void* pre_bf16, void* dpre_bf16, int64_t n, void* stream) {
  if (n == 0) return 0;
  MDLM_CHECK_ARG(dact_bf16 && pre_bf16 && dpre_bf16 && n > 0 && n % 8 == 0 &&
                 (((uintptr_t)dact_bf16 | (uintptr_t)pre_bf16 | (uintptr_t)dpre_bf16) & 15) == 0, "gelu_bwd: n %% 8 and 16-byte alignment");
  gelu_bwd_kernel<<<(unsigned)((n / 8 + 255) / 256), 256, 0, ST(stream)>>>((const uint4*)dact_bf16, (const uint4*)pre_bf16, (uint4*)dpre_bf16, n / 8);
  MDLM_LAUNCH_CHECK();
  return 0;
}

extern "C" int mdlm_transpose_bf16(const void* in, int64_t ld_in, int64_t batch_stride_in,
                                   void* out, int64_t ld_out, int64_t batch_stride_out,
                                   int R, int C, int batch, float* colsum_partial, void* stream) {
  if (R == 0 || C == 0 || batch == 0) return 0;
  MDLM_CHECK_ARG(in && out && R > 0 && C > 0 && batch > 0 && ld_in >= C && ld_out >= R, "transpose_bf16: bad args");
  MDLM_CHECK_ARG((R + 63) / 64 < 65536 && batch < 65536, "transpose_bf16: too large");
  const dim3 grid((unsigned)((C + 63) / 64), (unsigned)((R + 63) / 64), (unsigned)batch);
  const bool vec = R % 8 == 0 && C % 8 == 0 && ld_in % 8 == 0 && ld_out % 8 == 0 &&
                   batch_stride_in % 8 == 0 && batch_stride_out % 8 == 0 &&
                   (((uintptr_t)in | (uintptr_t)out) & 15) == 0;
  MDLM_CHECK_ARG(!colsum_partial || (vec && batch == 1),
                 "transpose_bf16: colsum_partial needs batch 1 and the 16-byte path (R, C, ld %% 8 == 0)");
  if (vec)
    transpose_bf16_vec_kernel<<<grid, 256, 0, ST(stream)>>>(
      (const bf16_t*)in, ld_in, batch_stride_in, (bf16_t*)out, ld_out, batch_stride_out, R, C, colsum_partial);
  else
    transpose_bf16_kernel<<<grid, 256, 0, ST(stream)>>>(
      (const bf16_t*)in, ld_in, batch_stride_in, (bf16_t*)out, ld_out, batch_stride_out, R, C);
  MDLM_LAUNCH_CHECK();
  return 0;
}

extern "C" int mdlm_qkv_grad_pack(const void* dq, const void* dk, const void* dv,
                                  const float* rot_cos, const float* rot_sin, void* dqkv,
                                  int n_seq, int H, int L, void* stream) {
  if (n_seq == 0) return 0;
  MDLM_CHECK_ARG(dq && dk && dv && rot_cos && rot_sin && dqkv && n_seq > 0 && H > 0 && L > 0, "qkv_grad_pack: bad args");
  MDLM_CHECK_ARG((((uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv | (uintptr_t)rot_cos | (uintptr_t)rot_sin | (uintptr_t)dqkv) & 15) == 0, "qkv_grad_pack: 16-byte alignment");
  const int64_t M = (int64_t)n_seq * L;
  const int64_t n = M * H * 4;
  qkv_grad_pack_kernel<<<(unsigned)((n + 255) / 256), 256, 0, ST(stream)>>>(
    (const bf16_t*)dq, (const bf16_t*)dk, (const bf16_t*)dv, rot_cos, rot_sin, (bf16_t*)dqkv, M, H, L);
  MDLM_LAUNCH_CHECK();
  return 0;
}

extern "C" int mdlm_embed_grad(const float* dx, const int64_t* idx, float* dtable,
                               int64_t rows, int D, int vocab, int hot_index,
                               float* hot_partial, void* stream) {
  if (rows == 0) return 0;
  MDLM_CHECK_ARG(dx && idx && dtable && rows > 0 && D > 0 && D % 4 == 0 && vocab > 0 && ((uintptr_t)dx & 15) == 0, "embed_grad: bad args");
  MDLM_CHECK_ARG(hot_index < vocab && (hot_index < 0 || hot_partial) && ((uintptr_t)hot_partial & 15) == 0,
                 "embed_grad: hot_index needs a 16-byte aligned hot_partial workspace");
  const int64_t nb = (rows + EG_ROWS - 1) / EG_ROWS;
  MDLM_CHECK_ARG(nb < (1ll << 31), "embed_grad: too many rows");
  embed_grad_kernel<<<(unsigned)nb, 256, 0, ST(stream)>>>(dx, idx, dtable, rows, D, vocab,
                                                          hot_index >= 0 ? hot_index : -1,
                                                          hot_index >= 0 ? hot_partial : nullptr);
  MDLM_LAUNCH_CHECK();
  if (hot_index >= 0) {   // the hot row receives nothing through the atomics
    colsum_reduce_kernel<<<dim3((unsigned)((D + 63) / 64), 1), 1024, 0, ST(stream)>>>(
      hot_partial, dtable + (int64_t)hot_index * D, D, (int)nb);
    MDLM_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int mdlm_subs_nll_bwd(const void* logits, int64_t ld, int dtype, const int64_t* xt,
                                 const int64_t* x0, const float* row_weight, void* dlogits_bf16,
                                 int64_t ldo, int64_t rows, int V, int mask_index, void* stream) {
  if (rows == 0) return 0;
  MDLM_CHECK_ARG(logits && xt && x0 && row_weight && dlogits_bf16, "subs_nll_bwd: null pointer");
  MDLM_CHECK_ARG(rows > 0 && rows < (1ll << 31) && V > 0 && ld >= V && ldo >= V && ldo % 8 == 0 && ((uintptr_t)dlogits_bf16 & 15) == 0,
                 "subs_nll_bwd: ldo must be a multiple of 8 and >= V, output 16-byte aligned");
  MDLM_CHECK_ARG(mask_index >= 0 && mask_index < V, "subs_nll_bwd: bad mask index");
  if (dtype == MDLM_BF16 && ld % 8 == 0 && ld >= ((V + 7) & ~7) && ((uintptr_t)logits & 15) == 0)
    subs_nll_bwd_vec_kernel<<<(unsigned)rows, 256, 0, ST(stream)>>>((const bf16_t*)logits, ld, xt, x0, row_weight, (bf16_t*)dlogits_bf16, ldo, V, mask_index);
  else if (dtype == MDLM_BF16)
    subs_nll_bwd_kernel<bf16_t><<<(unsigned)rows, 256, 0, ST(stream)>>>((const bf16_t*)logits, ld, xt, x0, row_weight, (bf16_t*)dlogits_bf16, ldo, V, mask_index);
  else if (dtype == MDLM_F32)
    subs_nll_bwd_kernel<float><<<(unsigned)rows, 256, 0, ST(stream)>>>((const float*)logits, ld, xt, x0, row_weight, (bf16_t*)dlogits_bf16, ldo, V, mask_index);
  else MDLM_CHECK_ARG(false, "subs_nll_bwd: bad dtype");
  MDLM_LAUNCH_CHECK();
  return 0;
}

extern "C" int mdlm_grad_sumsq(const float* g, int64_t n, float* partial_1024, float* out, void* stream) {
  MDLM_CHECK_ARG(g && partial_1024 && out && n >= 0, "grad_sumsq: bad args");
  sumsq_partial_kernel<<<1024, 256, 0, ST(stream)>>>(g, n, partial_1024);
  sumsq_final_kernel<<<1, 256, 0, ST(stream)>>>(partial_1024, 1024, out);
  MDLM_LAUNCH_CHECK();
  return 0;
}

// ---- the one entry point that allocates nothing AND needs no workspace ---------
// Same two kernels as mdlm_grad_sumsq; the 1024 per-workgroup partial sums live in
// a file-scope device array instead of a caller-supplied buffer, for callers that
// cannot allocate (HIP graph capture, tight serving loops).  Each GPU holds its
// own copy of the array at its own address for the life of the process (4 KiB).
constexpr int SUMSQ_PARTIALS = 1024;
constexpr int MDLM_MAX_DEVICES = 64;
static __device__ __align__(256) float scratch[SUMSQ_PARTIALS];
// Device address of `scratch` per GPU.  The array's NAME is only a host-side
// stand-in in host code; kernels, memsets and copies must be given this pointer.
static float* g_scratch_ptr[MDLM_MAX_DEVICES];
static std::mutex g_scratch_mu;

static int scratch_for_current_device(float** out) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) { set_error("grad_sumsq_static: hipGetDevice: %s", hipGetErrorString(e)); return (int)e; }
  if (dev < 0 || dev >= MDLM_MAX_DEVICES) { set_error("grad_sumsq_static: device %d outside [0,%d)", dev, MDLM_MAX_DEVICES); return -1; }
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  if (!g_scratch_ptr[dev]) {   // once per device, with that device current
    float* p = nullptr;
    e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !p) {
      set_error("grad_sumsq_static: hipGetSymbolAddress: %s", hipGetErrorString(e));
      return e != hipSuccess ? (int)e : -1;
    }
    g_scratch_ptr[dev] = p;
  }
  *out = g_scratch_ptr[dev];
  return 0;
}

extern "C" int mdlm_grad_sumsq_static(const float* g, int64_t n, float* out, void* stream) {
  MDLM_CHECK_ARG(g && out && n >= 0, "grad_sumsq_static: bad args");
  MDLM_CHECK_ARG(n <= MDLM_SUMSQ_STATIC_MAX_N,
                 "grad_sumsq_static: n = %lld exceeds the supported maximum of %lld elements; "
                 "use mdlm_grad_sumsq with a caller-owned workspace",
                 (long long)n, (long long)MDLM_SUMSQ_STATIC_MAX_N);
  float* part = nullptr;
  const int rc = scratch_for_current_device(&part);
  if (rc != 0) return rc;
  // every launch rewrites all 1024 partials, so nothing of the previous call survives
  sumsq_partial_kernel<<<SUMSQ_PARTIALS, 256, 0, ST(stream)>>>(g, n, part);
  sumsq_final_kernel<<<1, 256, 0, ST(stream)>>>(part, SUMSQ_PARTIALS, out);
  MDLM_LAUNCH_CHECK();
  return 0;
}

extern "C" int mdlm_adamw_ema(float* p, const float* g, float* m, float* v, float* ema,
                              void* w_bf16, int64_t n, float lr, float beta1, float beta2,
                              float eps, float weight_decay, int64_t step, float ema_one_minus_decay,
                              const float* sumsq, float max_norm, float grad_scale, void* stream) {
  if (n == 0) return 0;
  MDLM_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "adamw_ema: bad args");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  adamw_ema_kernel<<<(unsigned)((n + 255) / 256), 256, 0, ST(stream)>>>(
    p, g, m, v, ema, (bf16_t*)w_bf16, n, lr, beta1, beta2, eps, weight_decay, (float)bc1,
    (float)sqrt(bc2), ema_one_minus_decay, sumsq, max_norm, grad_scale);
  MDLM_LAUNCH_CHECK();
  return 0;
}
