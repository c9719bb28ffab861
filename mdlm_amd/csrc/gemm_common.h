// Shared definitions of the bf16 MFMA GEMM kernels (gemm_bf16.hip: 256x256 tile,
// LDS-staged epilogue; gemm_pp.hip: 128x256 tile, two accumulator sets, register-
// direct epilogue interleaved with the next tile's main loop).
#pragma once
#include "common.h"
#include <type_traits>

namespace mdlm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 64;
constexpr int EPI_BIAS_GELU_SAVE = 4;   // internal: mdlm_gemm_bf16_gelu mode 0
constexpr int EPI_MUL_GELU_GRAD = 5;    // internal: mdlm_gemm_bf16_gelu mode 1

struct GemmArgs {
  const bf16_t* A; int64_t lda;
  const bf16_t* W; int64_t ldw;
  // split-K (fp32 EPI_STORE only): split s reduces K columns [s*K, (s+1)*K) of
  // A and W into C + s*c_split_stride (partial sums, added up by a second kernel)
  int splits; int64_t c_split_stride;
  // training-step GELU fusions (bf16 outputs, same ldc as C):
  //   EPI_BIAS_GELU_SAVE : aux = second OUTPUT, receives acc + bias (the pre-activation)
  //   EPI_MUL_GELU_GRAD  : aux = INPUT pre-activation; C = (acc) * gelu'(aux)
  void* aux;
  const float* bias;
  void* C; int64_t ldc;
  int64_t M; int N; int K;
  const float* gate; const float* residual;
  int64_t rows_per_mod; int n_mod; int64_t mod_ld;
  const int32_t* out_sample_map;
  const int32_t* out_row_map;
  const float* rot_cos; const float* rot_sin; int n_heads;
  const int32_t* m_rows_dev;
  // EPI_STORE bf16 only: per (row, 64-column group) softmax partials (max, sum of
  // exp(x - max)) of the bf16-rounded outputs, column lse_skip excluded
  float2* lse_part; int lse_skip; int lse_ng;
  int tiles_m, tiles_n;
  int vec_ok;  // ldc / alignment allow 16-byte row stores
  int dbg;     // MDLM_GEMM_ABLATION builds only (2: skip the epilogue, 4: drop the stores)
  int whole_tiles;   // tile bit 4: never cut the tiles of a short last round in two
};

// Timing ablations exist only in builds with -DMDLM_GEMM_ABLATION (scratch/
// experiments); the release library carries no such switch and no global state.
#ifdef MDLM_GEMM_ABLATION
#define GEMM_DBG(g) ((g).dbg)
#else
#define GEMM_DBG(g) 0
#endif

#define GLDS16(gptr, lptr) \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr), \
                                   (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

__device__ __forceinline__ float gelu_tanh(float x) {
  // nn.GELU(approximate='tanh'), dit.py:227:
  //   0.5 x (1 + tanh(y)) = x * sigmoid(2y),  y = sqrt(2/pi) (x + 0.044715 x^3)
  // exp(-z) = exp2(-z log2 e); v_exp_f32 / v_rcp_f32 are 1-ulp hardware ops,
  // far inside the bf16 output rounding.
  const float z2 = -2.3022082f * x * (1.f + 0.044715f * x * x);  // -1.5957691*log2(e)
  return x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(z2));
}

// the same on a pair: five packed fp32 ops + 2 x (v_exp_f32, v_rcp_f32)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_tanh2(f32x2 x) {
  const f32x2 k1 = {-2.3022082f * 0.044715f, -2.3022082f * 0.044715f};
  const f32x2 k0 = {-2.3022082f, -2.3022082f};
  const f32x2 one = {1.f, 1.f};
  const f32x2 w = x * __builtin_elementwise_fma(x * x, k1, k0);
  const f32x2 e = {__builtin_amdgcn_exp2f(w.x), __builtin_amdgcn_exp2f(w.y)};
  const f32x2 d = e + one;
  const f32x2 r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  return x * r;
}

// d/dx of gelu_tanh: 0.5 (1 + th) + 0.5 x (1 - th^2) sqrt(2/pi) (1 + 3*0.044715 x^2),
// th = tanh(y) = 1 - 2 / (1 + exp(2y))
__device__ __forceinline__ float gelu_tanh_grad_fast(float x) {
  const float x2 = x * x;
  const float y2l = 2.3022082f * x * (1.f + 0.044715f * x2);   // 2y * log2(e)
  const float th = 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(y2l));
  return 0.5f * (1.f + th) + 0.5f * x * (1.f - th * th) * (0.7978845608f * (1.f + 0.134145f * x2));
}

__device__ __forceinline__ float bf16_round(float x) { return bf16_to_f32(f32_to_bf16(x)); }
// round a pair to bf16 and back: one v_cvt_pk_bf16_f32 + shift + and
__device__ __forceinline__ f32x2 bf16_round2(float a, float b) {
  const unsigned u = pack_bf16x2(a, b);
  return (f32x2){__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};
}

// 16-byte row store of the epilogue.  bf16 outputs (q/k, MLP hidden, logits)
// are written once and read by a later kernel: non-temporal stores keep them
// from displacing the A/W panels in L2 (measured -7..-14 % per launch); the
// fp32 residual stream is read-modify-written in place and stays plain.
// dbg bit 4 (timing experiments only) drops the store.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ void store16(void* p, uint4 v, int dbg) {
  if (dbg & 4) return;
  const u32x4 vv = {v.x, v.y, v.z, v.w};
  if (NT) __builtin_nontemporal_store(vv, reinterpret_cast<u32x4*>(p));
  else *reinterpret_cast<u32x4*>(p) = vv;
}

}  // namespace mdlm
