// HBM-bound DiT glue kernels: embedding gather, timestep embedding, the small
// fp32 conditioning linears and LayerNorm+modulate.
// Reference: models/dit.py:126-134 (LayerNorm), 74-78 (modulate_fused),
// 150-189 (TimestepEmbedder), 249-250/318 (adaLN), 292-299 (EmbeddingLayer).
#include "common.h"

namespace mdlm {

// one wave per row, float4 copies
__global__ __launch_bounds__(256) void embed_rows_kernel(
    const float* __restrict__ table, const int64_t* __restrict__ idx,
    const int32_t* __restrict__ sample_map, int L, float* __restrict__ out,
    int64_t n_rows, int D, int vocab, const int32_t* __restrict__ n_rows_dev) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t live = n_rows_dev ? (int64_t)*n_rows_dev : n_rows;
  if (r >= n_rows || r >= live) return;
  const int lane = threadIdx.x & 63;
  int64_t src = r;
  if (sample_map) src = (int64_t)sample_map[r / L] * L + (r % L);
  int64_t tok = idx[src];
  tok = tok < 0 ? 0 : (tok >= vocab ? vocab - 1 : tok);  // never read out of bounds
  const float4* s = reinterpret_cast<const float4*>(table + tok * D);
  float4* d = reinterpret_cast<float4*>(out + r * D);
  for (int i = lane; i < D / 4; i += 64) d[i] = s[i];
}

// dst[sample_map[r / L] * L + r % L] = src[r]: the residual stream of the
// compacted active samples back to its per-sample home (hidden-state cache of
// the opt-in lazy sampler).
__global__ __launch_bounds__(256) void scatter_rows_kernel(
    const float* __restrict__ src, float* __restrict__ dst,
    const int32_t* __restrict__ sample_map, int L, int D, int64_t n_rows,
    const int32_t* __restrict__ n_rows_dev) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t live = n_rows_dev ? (int64_t)*n_rows_dev : n_rows;
  if (r >= n_rows || r >= live) return;
  const int lane = threadIdx.x & 63;
  const int64_t d = sample_map ? (int64_t)sample_map[r / L] * L + (r % L) : r;
  const float4* s = reinterpret_cast<const float4*>(src + r * D);
  float4* o = reinterpret_cast<float4*>(dst + d * D);
  for (int i = lane; i < D / 4; i += 64) o[i] = s[i];
}

// dst rows i < *n_rows_dev are copies of src rows gather[i]: the fp32 residual
// stream and the bf16 attention output of the rows the last block still has to
// finish (the masked positions), packed densely.
__global__ __launch_bounds__(256) void gather_rows_kernel(
    const float* __restrict__ xs, float* __restrict__ xd,
    const bf16_t* __restrict__ as, bf16_t* __restrict__ ad,
    const int32_t* __restrict__ gather, int D, int64_t n_rows,
    const int32_t* __restrict__ n_rows_dev) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t live = n_rows_dev ? (int64_t)*n_rows_dev : n_rows;
  if (r >= n_rows || r >= live) return;
  const int lane = threadIdx.x & 63;
  const int64_t sr = gather[r];
  const float4* s4 = reinterpret_cast<const float4*>(xs + sr * D);
  float4* d4 = reinterpret_cast<float4*>(xd + r * D);
  for (int i = lane; i < D / 4; i += 64) d4[i] = s4[i];
  const uint4* s8 = reinterpret_cast<const uint4*>(as + sr * D);
  uint4* d8 = reinterpret_cast<uint4*>(ad + r * D);
  for (int i = lane; i < D / 8; i += 64) d8[i] = s8[i];
}

__global__ void timestep_embedding_kernel(const float* __restrict__ t,
                                          float* __restrict__ out, int B, int dim) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int half = dim / 2;
  if (i >= B * half) return;
  const int b = i / half, j = i % half;
  // dit.py:174-179 in fp32: exp(-ln(10000) * j / half)
  const float f = expf((-9.210340371976184f * (float)j) / (float)half);
  const float a = t[b] * f;
  out[(int64_t)b * dim + j] = cosf(a);
  out[(int64_t)b * dim + half + j] = sinf(a);
  if ((dim & 1) && j == 0) out[(int64_t)b * dim + dim - 1] = 0.f;
}

__device__ __forceinline__ float silu(float x) { return x / (1.f + expf(-x)); }

// one wave per output element (b, n)
__global__ __launch_bounds__(256) void linear_f32_kernel(
    const float* __restrict__ in, const float* __restrict__ W,
    const float* __restrict__ bias, float* __restrict__ out, int B, int N, int K,
    int act_in, int act_out) {
  const int64_t o = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (o >= (int64_t)B * N) return;
  const int b = (int)(o / N), n = (int)(o % N);
  const int lane = threadIdx.x & 63;
  float acc = 0.f;
  for (int k = lane; k < K; k += 64) {
    float x = in[(int64_t)b * K + k];
    if (act_in == 1) x = silu(x);
    acc += x * W[(int64_t)n * K + k];
  }
  acc = wave_sum(acc);
  if (lane == 0) {
    if (bias) acc += bias[n];
    if (act_out == 1) acc = silu(acc);
    out[o] = acc;
  }
}

// one wave per row; row cached in registers (D <= 1024)
constexpr int LN_NV = 4;
__global__ __launch_bounds__(256) void ln_modulate_kernel(
    const float* __restrict__ x, const float* __restrict__ ln_w,
    const float* __restrict__ shift, const float* __restrict__ scale,
    bf16_t* __restrict__ out, int64_t rows, int D, int64_t rows_per_mod,
    int n_mod, int64_t mod_ld, const int32_t* __restrict__ row_gather,
    const int32_t* __restrict__ n_rows_dev) {
  const int64_t ro = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // output row
  const int64_t live = n_rows_dev ? (int64_t)*n_rows_dev : rows;
  if (ro >= rows || ro >= live) return;
  const int64_t r = row_gather ? (int64_t)row_gather[ro] : ro;       // source row
  const int lane = threadIdx.x & 63;
  const int nv = D / 4;
  const float4* xr = reinterpret_cast<const float4*>(x + r * D);
  float4 v[LN_NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_NV; ++i) {
    const int c = lane + i * 64;
    v[i] = c < nv ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_NV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)D + 1e-5f);
  const int64_t mrow = (r / rows_per_mod) % n_mod;
  const float4* w4 = reinterpret_cast<const float4*>(ln_w);
  const float4* sh4 = reinterpret_cast<const float4*>(shift + mrow * mod_ld);
  const float4* sc4 = reinterpret_cast<const float4*>(scale + mrow * mod_ld);
  uint2* o = reinterpret_cast<uint2*>(out + ro * D);
#pragma unroll
  for (int i = 0; i < LN_NV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      const float4 w = w4[c], sh = sh4[c], sc = sc4[c];
      // dit.py:133-134 then 74-78: (LN(x) * w) * (1 + scale) + shift
      const float a = ((v[i].x - mean) * rstd * w.x) * (1.f + sc.x) + sh.x;
      const float b = ((v[i].y - mean) * rstd * w.y) * (1.f + sc.y) + sh.y;
      const float cc = ((v[i].z - mean) * rstd * w.z) * (1.f + sc.z) + sh.z;
      const float d = ((v[i].w - mean) * rstd * w.w) * (1.f + sc.w) + sh.w;
      o[c] = make_uint2(pack_bf16x2(a, b), pack_bf16x2(cc, d));
    }
  }
}

}  // namespace mdlm

using namespace mdlm;
#define ST(s) ((hipStream_t)(s))

extern "C" int mdlm_embed_rows(const float* table, const int64_t* idx,
                               const int32_t* sample_map, int L, float* out,
                               int64_t n_rows, int D, int vocab,
                               const int32_t* n_rows_dev, void* stream) {
  if (n_rows == 0) return 0;
  MDLM_CHECK_ARG(table && idx && out, "embed_rows: null pointer");
  MDLM_CHECK_ARG(n_rows >= 0 && D > 0 && D % 4 == 0 && vocab > 0 && L > 0, "embed_rows: bad shape (D %% 4 must be 0)");
  if (n_rows == 0) return 0;
  embed_rows_kernel<<<(unsigned)((n_rows + 3) / 4), 256, 0, ST(stream)>>>(table, idx, sample_map, L, out, n_rows, D, vocab, n_rows_dev);
  MDLM_LAUNCH_CHECK();
  return 0;
}

extern "C" int mdlm_timestep_embedding(const float* t, float* out, int B, int dim, void* stream) {
  if (B == 0) return 0;
  MDLM_CHECK_ARG(t && out && B >= 0 && dim >= 2, "timestep_embedding: bad args");
  const int n = B * (dim / 2);
  timestep_embedding_kernel<<<(n + 255) / 256, 256, 0, ST(stream)>>>(t, out, B, dim);
  MDLM_LAUNCH_CHECK();
  return 0;
}

extern "C" int mdlm_linear_f32(const float* in, const float* W, const float* bias,
                               float* out, int B, int N, int K, int act_in,
                               int act_out, void* stream) {
  if (B == 0) return 0;
  MDLM_CHECK_ARG(in && W && out && B >= 0 && N > 0 && K > 0, "linear_f32: bad args");
  MDLM_CHECK_ARG((act_in == 0 || act_in == 1) && (act_out == 0 || act_out == 1), "linear_f32: bad activation");
  if (B == 0) return 0;
  const int64_t n = (int64_t)B * N;
  linear_f32_kernel<<<(unsigned)((n + 3) / 4), 256, 0, ST(stream)>>>(in, W, bias, out, B, N, K, act_in, act_out);
  MDLM_LAUNCH_CHECK();
  return 0;
}

extern "C" int mdlm_ln_modulate(const float* x, const float* ln_w,
                                const float* shift, const float* scale,
                                void* out_bf16, int64_t rows, int D,
                                int64_t rows_per_mod, int n_mod, int64_t mod_ld,
                                const int32_t* row_gather,
                                const int32_t* n_rows_dev, void* stream) {
  if (rows == 0) return 0;
  MDLM_CHECK_ARG(x && ln_w && shift && scale && out_bf16, "ln_modulate: null pointer");
  MDLM_CHECK_ARG(rows >= 0 && D > 0 && D % 4 == 0 && D <= 256 * LN_NV, "ln_modulate: D=%d must be a multiple of 4 and <= %d", D, 256 * LN_NV);
  MDLM_CHECK_ARG(rows_per_mod > 0 && n_mod > 0 && mod_ld % 4 == 0 && (n_mod == 1 || mod_ld >= D), "ln_modulate: bad modulation shape");
  MDLM_CHECK_ARG((((uintptr_t)x | (uintptr_t)ln_w | (uintptr_t)shift | (uintptr_t)scale | (uintptr_t)out_bf16) & 15) == 0, "ln_modulate: buffers must be 16-byte aligned");
  if (rows == 0) return 0;
  ln_modulate_kernel<<<(unsigned)((rows + 3) / 4), 256, 0, ST(stream)>>>(x, ln_w, shift, scale, (bf16_t*)out_bf16, rows, D, rows_per_mod, n_mod, mod_ld, row_gather, n_rows_dev);
  MDLM_LAUNCH_CHECK();
  return 0;
}

extern "C" int mdlm_scatter_rows_f32(const float* src, float* dst,
                                     const int32_t* sample_map, int L, int D,
                                     int64_t n_rows, const int32_t* n_rows_dev,
                                     void* stream) {
  MDLM_CHECK_ARG(src && dst && L > 0 && D > 0 && D % 4 == 0 && n_rows >= 0, "scatter_rows_f32: bad args");
  MDLM_CHECK_ARG((((uintptr_t)src | (uintptr_t)dst) & 15) == 0, "scatter_rows_f32: 16-byte alignment");
  if (n_rows == 0) return 0;
  scatter_rows_kernel<<<(unsigned)((n_rows + 3) / 4), 256, 0, ST(stream)>>>(src, dst, sample_map, L, D, n_rows, n_rows_dev);
  MDLM_LAUNCH_CHECK();
  return 0;
}

extern "C" int mdlm_gather_rows(const float* x_src, float* x_dst, const void* a_src,
                                void* a_dst, const int32_t* gather, int D,
                                int64_t n_rows, const int32_t* n_rows_dev,
                                void* stream) {
  if (n_rows == 0) return 0;
  MDLM_CHECK_ARG(x_src && x_dst && a_src && a_dst && gather, "gather_rows: null pointer");
  MDLM_CHECK_ARG(D > 0 && D % 8 == 0 && n_rows > 0, "gather_rows: D must be a multiple of 8");
  MDLM_CHECK_ARG((((uintptr_t)x_src | (uintptr_t)x_dst | (uintptr_t)a_src | (uintptr_t)a_dst) & 15) == 0,
                 "gather_rows: 16-byte alignment");
  gather_rows_kernel<<<(unsigned)((n_rows + 3) / 4), 256, 0, ST(stream)>>>(
    x_src, x_dst, (const bf16_t*)a_src, (bf16_t*)a_dst, gather, D, n_rows, n_rows_dev);
  MDLM_LAUNCH_CHECK();
  return 0;
}
