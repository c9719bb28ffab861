// Sampler-side kernels of the MDLM hot path (HBM-bound, one workgroup per
// vocab row): SUBS parameterisation, ddpm_cache update with the Gumbel-argmax
// race, noise-removal argmax, q_xt and the NELBO gather.
// Reference: diffusion.py:24-28, 261-277, 575-610, 690-697, 884-894.
#include "common.h"

namespace mdlm {

constexpr int NT = 256;            // threads per row-workgroup (4 waves)
constexpr float NEG_INF = -1000000.0f;  // diffusion.py:150

// ------------------------------------------------------------------ helpers
template <typename T> struct vec_traits;
template <> struct vec_traits<float> { static constexpr int EPV = 4; };
template <> struct vec_traits<bf16_t> { static constexpr int EPV = 8; };

template <typename T>
__device__ __forceinline__ void unpack(const uint4& v, float (&f)[vec_traits<T>::EPV]);
template <>
__device__ __forceinline__ void unpack<float>(const uint4& v, float (&f)[4]) {
  f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y);
  f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
}
template <>
__device__ __forceinline__ void unpack<bf16_t>(const uint4& v, float (&f)[8]) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}

// elements of `row` before the first 16-byte boundary
template <typename T>
__device__ __forceinline__ int row_head(const T* row, int V) {
  int h = (int)(((16 - ((uintptr_t)row & 15)) & 15) / sizeof(T));
  return h < V ? h : V;
}

// Visit every element of a row once: 16-byte vector loads for the aligned
// body, scalar loads for the (<16 B) head and tail.  fv(idx0, float[EPV]) and
// fs(idx, float).  Per thread the visited indices are ascending.
template <typename T, typename FV, typename FS>
__device__ __forceinline__ void row_visit(const T* row, int V, FV&& fv, FS&& fs) {
  constexpr int EPV = vec_traits<T>::EPV;
  const int head = row_head(row, V);
  for (int i = threadIdx.x; i < head; i += NT) fs(i, elem<T>::load(row + i));
  const int nvec = (V - head) / EPV;
  const uint4* vp = reinterpret_cast<const uint4*>(row + head);
  // four 16-byte loads in flight per thread: inside the update kernel (few
  // resident waves) one load per round trip left this pass latency-bound
  int i = threadIdx.x;
  for (; i + 3 * NT < nvec; i += 4 * NT) {
    uint4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = vp[i + k * NT];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float f[EPV];
      unpack<T>(v[k], f);
      fv(head + (i + k * NT) * EPV, f);
    }
  }
  for (; i < nvec; i += NT) {
    float f[EPV];
    unpack<T>(vp[i], f);
    fv(head + i * EPV, f);
  }
  for (int j = head + nvec * EPV + threadIdx.x; j < V; j += NT)
    fs(j, elem<T>::load(row + j));
}

struct LseState {
  float m = -INFINITY, s = 0.f;
  __device__ __forceinline__ void add1(float x) {
    if (x > m) { s = s * __expf(m - x); m = x; }
    s += __expf(x - m);
  }
  // N elements at once (N even): max by v_max3, exp2(x log2e - m log2e) with one
  // packed fma per pair, packed adds -- 3.5 VALU instructions per element instead
  // of 6 (the pass is VALU-bound: the row is L2-warm or streams at a fraction of
  // the VALU time)
  template <int N> __device__ __forceinline__ void addv(const float (&x)[N]) {
    static_assert(N % 2 == 0, "pairs");
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    constexpr float L2E = 1.4426950408889634f;
    float vm = fmaxf(x[0], x[1]);
#pragma unroll
    for (int k = 2; k < N; k += 2) vm = fmaxf(fmaxf(vm, x[k]), x[k + 1]);
    if (vm > m) { s = s * __builtin_amdgcn_exp2f((m - vm) * L2E); m = vm; }
    const float nm = m == -INFINITY ? 0.f : -m * L2E;   // a row of -inf so far adds exp2(-inf) = 0
    const f32x2 l2 = {L2E, L2E}, nm2 = {nm, nm};
    f32x2 acc = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < N; k += 2) {
      const f32x2 t = __builtin_elementwise_fma((f32x2){x[k], x[k + 1]}, l2, nm2);
      acc += (f32x2){__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
    }
    s += acc.x + acc.y;
  }
};

// block-wide log-sum-exp from per-thread online states; result in all threads
__device__ __forceinline__ float block_lse(LseState st, float* sm /*[2*4+1]*/) {
  float M = wave_max(st.m);
  float s = (st.m == -INFINITY) ? 0.f : st.s * __expf(st.m - M);
  float S = wave_sum(s);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sm[w] = M; sm[4 + w] = S; }
  __syncthreads();
  float Mb = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
  float Sb = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) Sb += sm[4 + i] * __expf(sm[i] - Mb);
  __syncthreads();
  return Mb + logf(Sb);
}

// SUBS row log-sum-exp with the mask column shifted by -1e6 (diffusion.py:263-268)
template <typename T>
__device__ __forceinline__ float subs_row_lse(const T* row, int V, int mask, float* sm) {
  LseState st;
  constexpr int EPV = vec_traits<T>::EPV;
  row_visit<T>(row, V,
    [&](int i0, float (&f)[EPV]) {
      if (mask >= i0 && mask < i0 + EPV) {
#pragma unroll
        for (int k = 0; k < EPV; ++k) if (i0 + k == mask) f[k] += NEG_INF;
      }
      st.addv(f);
    },
    [&](int i, float x) { st.add1(i == mask ? x + NEG_INF : x); });
  return block_lse(st, sm);
}

struct ArgMax {
  float v = -INFINITY; int i = 0x7fffffff;
  // torch.argmax semantics: NaN counts as the maximum and the first index
  // wins ties; per thread the visit order is ascending, so strict > suffices.
  // Branch-free (two compares, lane-mask logic, two selects) so that the row
  // loop can be unrolled with several vector loads in flight.
  __device__ __forceinline__ void put(float x, int idx) {
    const bool first = i == 0x7fffffff;
    const bool take = first | ((v == v) & ((x != x) | (x > v)));
    v = take ? x : v;
    i = take ? idx : i;
  }
};
__device__ __forceinline__ bool am_better(float av, int ai, float bv, int bi) {
  // torch.argmax: NaN counts as the maximum; first index wins ties
  const bool an = av != av, bn = bv != bv;
  if (an != bn) return an;
  if (!an && av != bv) return av > bv;
  return ai < bi;
}
__device__ __forceinline__ int block_argmax(ArgMax a, float* smv, int* smi,
                                            float* best_val = nullptr) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float ov = __shfl_xor(a.v, o); int oi = __shfl_xor(a.i, o);
    if (am_better(ov, oi, a.v, a.i)) { a.v = ov; a.i = oi; }
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { smv[w] = a.v; smi[w] = a.i; }
  __syncthreads();
  float bv = smv[0]; int bi = smi[0];
#pragma unroll
  for (int k = 1; k < 4; ++k)
    if (am_better(smv[k], smi[k], bv, bi)) { bv = smv[k]; bi = smi[k]; }
  __syncthreads();
  if (best_val) *best_val = bv;
  return bi;
}

// ------------------------------------------------------ (a) SUBS log-probs
template <typename TI, typename TO, bool PROBS>
__global__ __launch_bounds__(NT) void subs_logprobs_kernel(
    const TI* __restrict__ logits, int64_t ld_in, const int64_t* __restrict__ xt,
    TO* __restrict__ out, int64_t ld_out, int V, int mask) {
  __shared__ float sm[9];
  const int64_t r = blockIdx.x;
  const TI* row = logits + r * ld_in;
  TO* orow = out + r * ld_out;
  const int64_t tok = xt[r];
  if (tok != mask) {  // diffusion.py:274-276
    for (int i = threadIdx.x; i < V; i += NT)
      elem<TO>::store(orow + i, i == tok ? (PROBS ? 1.f : 0.f) : (PROBS ? 0.f : NEG_INF));
    return;
  }
  const float lse = subs_row_lse<TI>(row, V, mask, sm);
  for (int i = threadIdx.x; i < V; i += NT) {
    float x = elem<TI>::load(row + i);
    if (i == mask) x += NEG_INF;
    elem<TO>::store(orow + i, PROBS ? expf(x - lse) : x - lse);
  }
}

// vectorised variant: both rows 16-byte aligned, same element type
template <typename T, bool PROBS>
__global__ __launch_bounds__(NT) void subs_logprobs_vec_kernel(
    const T* __restrict__ logits, int64_t ld_in, const int64_t* __restrict__ xt,
    T* __restrict__ out, int64_t ld_out, int V, int mask) {
  __shared__ float sm[9];
  constexpr int EPV = vec_traits<T>::EPV;
  const int64_t r = blockIdx.x;
  const T* row = logits + r * ld_in;
  T* orow = out + r * ld_out;
  const int64_t tok = xt[r];
  const int nvec = V / EPV;
  uint4* ov = reinterpret_cast<uint4*>(orow);
  auto pack = [](const float (&f)[EPV]) {
    uint4 o;
    if constexpr (EPV == 4) {
      o = make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]),
                     __float_as_uint(f[2]), __float_as_uint(f[3]));
    } else {
      o = make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]),
                     pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
    }
    return o;
  };
  if (tok != mask) {
    for (int i = threadIdx.x; i < nvec; i += NT) {
      float f[EPV];
#pragma unroll
      for (int k = 0; k < EPV; ++k)
        f[k] = (i * EPV + k == tok) ? (PROBS ? 1.f : 0.f) : (PROBS ? 0.f : NEG_INF);
      ov[i] = pack(f);
    }
    for (int i = nvec * EPV + threadIdx.x; i < V; i += NT)
      elem<T>::store(orow + i, i == tok ? (PROBS ? 1.f : 0.f) : (PROBS ? 0.f : NEG_INF));
    return;
  }
  const float lse = subs_row_lse<T>(row, V, mask, sm);
  const uint4* iv = reinterpret_cast<const uint4*>(row);
  for (int i = threadIdx.x; i < nvec; i += NT) {
    float f[EPV];
    unpack<T>(iv[i], f);
#pragma unroll
    for (int k = 0; k < EPV; ++k) {
      if (i * EPV + k == mask) f[k] += NEG_INF;
      f[k] = PROBS ? expf(f[k] - lse) : f[k] - lse;
    }
    ov[i] = pack(f);
  }
  for (int i = nvec * EPV + threadIdx.x; i < V; i += NT) {
    float x = elem<T>::load(row + i);
    if (i == mask) x += NEG_INF;
    elem<T>::store(orow + i, PROBS ? expf(x - lse) : x - lse);
  }
}

// bf16 rows of up to REG_VEC * NT * 8 elements (51,200: the GPT-2 vocabulary fits):
// the row is read from HBM ONCE into registers (25 x 16 bytes per thread, all
// loads in flight together), the log-sum-exp is taken over the registers in the
// same per-thread order as subs_row_lse, and the normalised row is written from
// them -- one read + one write per masked row instead of two reads + one write.
constexpr int REG_VEC = 25;
template <bool PROBS>
__global__ __launch_bounds__(NT, 3) void subs_logprobs_reg_kernel(
    const bf16_t* __restrict__ logits, int64_t ld_in, const int64_t* __restrict__ xt,
    bf16_t* __restrict__ out, int64_t ld_out, int V, int mask) {
  __shared__ float sm[9];
  const int64_t r = blockIdx.x;
  const bf16_t* row = logits + r * ld_in;
  bf16_t* orow = out + r * ld_out;
  const int64_t tok = xt[r];
  const int nvec = V / 8;
  uint4* ov = reinterpret_cast<uint4*>(orow);
  if (tok != mask) {  // diffusion.py:274-276
    const unsigned fill = PROBS ? 0u : pack_bf16x2(NEG_INF, NEG_INF);
    for (int i = threadIdx.x; i < nvec; i += NT) {
      uint4 o = make_uint4(fill, fill, fill, fill);
      if ((unsigned)(tok - (int64_t)i * 8) < 8u) {
        float f[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
          f[k] = (i * 8 + k == tok) ? (PROBS ? 1.f : 0.f) : (PROBS ? 0.f : NEG_INF);
        o = make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]),
                       pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
      }
      ov[i] = o;
    }
    for (int i = nvec * 8 + threadIdx.x; i < V; i += NT)
      elem<bf16_t>::store(orow + i, i == tok ? (PROBS ? 1.f : 0.f) : (PROBS ? 0.f : NEG_INF));
    return;
  }
  const uint4* iv = reinterpret_cast<const uint4*>(row);
  uint4 reg[REG_VEC];
#pragma unroll
  for (int j = 0; j < REG_VEC; ++j) {
    const int i = threadIdx.x + j * NT;
    reg[j] = i < nvec ? iv[i] : make_uint4(0u, 0u, 0u, 0u);
  }
  LseState st;
#pragma unroll
  for (int j = 0; j < REG_VEC; ++j) {
    const int i = threadIdx.x + j * NT;
    if (i < nvec) {
      float f[8];
      unpack<bf16_t>(reg[j], f);
      if ((unsigned)(mask - i * 8) < 8u) {
#pragma unroll
        for (int k = 0; k < 8; ++k) if (i * 8 + k == mask) f[k] += NEG_INF;
      }
      st.addv(f);
    }
  }
  for (int i = nvec * 8 + threadIdx.x; i < V; i += NT) {
    const float x = elem<bf16_t>::load(row + i);
    st.add1(i == mask ? x + NEG_INF : x);
  }
  const float lse = block_lse(st, sm);
  // keep the row PACKED across the reduction: without this the compiler reuses
  // the first pass's unpacked fp32 values (200 more registers, one wave per SIMD)
#pragma unroll
  for (int j = 0; j < REG_VEC; ++j)
    asm volatile("" : "+v"(reg[j].x), "+v"(reg[j].y), "+v"(reg[j].z), "+v"(reg[j].w));
#pragma unroll
  for (int j = 0; j < REG_VEC; ++j) {
    const int i = threadIdx.x + j * NT;
    if (i < nvec) {
      float f[8];
      unpack<bf16_t>(reg[j], f);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (i * 8 + k == mask) f[k] += NEG_INF;
        f[k] = PROBS ? expf(f[k] - lse) : f[k] - lse;
      }
      ov[i] = make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]),
                         pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
    }
  }
  for (int i = nvec * 8 + threadIdx.x; i < V; i += NT) {
    float x = elem<bf16_t>::load(row + i);
    if (i == mask) x += NEG_INF;
    elem<bf16_t>::store(orow + i, PROBS ? expf(x - lse) : x - lse);
  }
}

template <typename T>
__global__ __launch_bounds__(NT) void subs_row_lse_kernel(
    const T* __restrict__ logits, int64_t ld, const int64_t* __restrict__ xt,
    float* __restrict__ lse, int V, int mask,
    const int32_t* __restrict__ sample_map, int L,
    const int32_t* __restrict__ n_rows_dev) {
  __shared__ float sm[9];
  int64_t r = blockIdx.x;
  if (n_rows_dev && r >= (int64_t)*n_rows_dev) return;
  if (sample_map) r = (int64_t)sample_map[r / L] * L + (r % L);
  if (xt[r] != mask) { if (threadIdx.x == 0) lse[r] = 0.f; return; }
  const float v = subs_row_lse<T>(logits + r * ld, V, mask, sm);
  if (threadIdx.x == 0) lse[r] = v;
}

// --------------------------------------------------- (a') noise removal argmax
template <typename T>
__global__ __launch_bounds__(NT) void subs_argmax_kernel(
    const T* __restrict__ logits, int64_t ld, const int64_t* __restrict__ xt,
    int64_t* __restrict__ out, int V, int mask) {
  __shared__ float smv[4]; __shared__ int smi[4];
  constexpr int EPV = vec_traits<T>::EPV;
  const int64_t r = blockIdx.x;
  const int64_t tok = xt[r];
  if (tok != mask) { if (threadIdx.x == 0) out[r] = tok; return; }
  // argmax(logit - lse) == argmax(shifted logit): subtracting the row
  // constant is monotone; ties broken by first index like torch.argmax.
  ArgMax am;
  row_visit<T>(logits + r * ld, V,
    [&](int i0, float (&f)[EPV]) {
#pragma unroll
      for (int k = 0; k < EPV; ++k)
        am.put(i0 + k == mask ? f[k] + NEG_INF : f[k], i0 + k);
    },
    [&](int i, float x) { am.put(i == mask ? x + NEG_INF : x, i); });
  const int best = block_argmax(am, smv, smi);
  if (threadIdx.x == 0) out[r] = best;
}

// ----------------------------------------- (b) ddpm_cache update / Gumbel race
// The race of diffusion.py:24-28 in its DEFINING arithmetic (what every token id
// is bit-exact to):  key_v = q_v / g_v  (IEEE fp32 division),
//   g_v = 1e-10f - RN(ln(u_v + 1e-10f))   (correctly rounded: fp64 log, rounded once)
//   q_v = p_v * factor (fp32), q_mask = ms;  SRC 1: p_v = expf(logit_v - lse).
// argmax with torch semantics (NaN is the maximum, first index wins ties).
__device__ __forceinline__ float gumbel_norm_exact(float u) {
  return 1e-10f - (float)log((double)(u + 1e-10f));
}
__device__ __forceinline__ float race_exact(float q, float u) {
  return q / gumbel_norm_exact(u);
}

// SRC 0: row holds probabilities p_x0.  SRC 1: row holds raw logits + lse.
template <int SRC>
__device__ __forceinline__ float q_of(float x, int v, int mask, float lse,
                                      float factor, float ms) {
  if (v == mask) return ms;  // diffusion.py:606
  float p;
  if constexpr (SRC == 0) p = x;
  else p = expf(x - lse);
  return p * factor;  // diffusion.py:605
}

// One row of the race in the defining arithmetic; result in all threads.
template <typename T, int SRC, bool PHILOX>
__device__ __forceinline__ int exact_row_race(
    const T* __restrict__ row, int V, int mask, float lse, float factor, float ms,
    const float* __restrict__ urow, uint64_t seed, uint64_t ctr0, float* smv,
    int* smi) {
  constexpr int EPV = vec_traits<T>::EPV;
  ArgMax am;
  const int G = (V + 3) >> 2;
  if constexpr (PHILOX) {
    if (row_head(row, V) == 0) {
      // aligned rows: 16-byte loads, EPV/4 Philox groups per vector
      const int nvec = V / EPV;
      const uint4* vp = reinterpret_cast<const uint4*>(row);
      for (int i = threadIdx.x; i < nvec; i += NT) {
        float f[EPV];
        unpack<T>(vp[i], f);
#pragma unroll
        for (int gq = 0; gq < EPV / 4; ++gq) {
          const int v0 = i * EPV + gq * 4;
          const uint4 rn = Philox::gen(seed, ctr0 + (uint64_t)(v0 >> 2));
          const unsigned rr[4] = {rn.x, rn.y, rn.z, rn.w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float q = q_of<SRC>(f[gq * 4 + k], v0 + k, mask, lse, factor, ms);
            am.put(race_exact(q, Philox::u01(rr[k])), v0 + k);
          }
        }
      }
      for (int v = nvec * EPV + threadIdx.x; v < V; v += NT) {
        const uint4 rn = Philox::gen(seed, ctr0 + (uint64_t)(v >> 2));
        const unsigned rr[4] = {rn.x, rn.y, rn.z, rn.w};
        const float q = q_of<SRC>(elem<T>::load(row + v), v, mask, lse, factor, ms);
        am.put(race_exact(q, Philox::u01(rr[v & 3])), v);
      }
    } else {
      // unaligned rows: one Philox group (4 consecutive columns) per thread
      for (int g = threadIdx.x; g < G; g += NT) {
        const uint4 rn = Philox::gen(seed, ctr0 + (uint64_t)g);
        const unsigned rr[4] = {rn.x, rn.y, rn.z, rn.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int v = g * 4 + k;
          if (v < V) {
            const float q = q_of<SRC>(elem<T>::load(row + v), v, mask, lse, factor, ms);
            am.put(race_exact(q, Philox::u01(rr[k])), v);
          }
        }
      }
    }
  } else {
    for (int v = threadIdx.x; v < V; v += NT) {
      const float q = q_of<SRC>(elem<T>::load(row + v), v, mask, lse, factor, ms);
      am.put(race_exact(q, urow[v]), v);
    }
  }
  return block_argmax(am, smv, smi);
}

// Per-sample race weights: ddpm_cache (diffusion.py:598-606) or the generic pair.
__device__ __forceinline__ void race_weights(const float* t, float dt,
                                             const float* q_scale,
                                             const float* q_mask, int b,
                                             float& factor, float& ms) {
  if (q_scale) {   // generic race: q_v = p_v * q_scale[b], q_mask = q_mask[b]
    factor = q_scale[b]; ms = q_mask[b];
  } else {
    const float mt = t[b];
    ms = mt - dt;          // diffusion.py:599 (fp32)
    factor = mt - ms;      // diffusion.py:605 (not fp32(dt))
  }
}

// Reference kernel of the race: every element in the defining arithmetic.  Used
// when the uniforms are supplied (`u`), or on request (fast_math = 0).
template <typename T, int SRC, bool PHILOX>
__global__ __launch_bounds__(NT) void ddpm_update_kernel(
    const T* __restrict__ src, int64_t ld, float* __restrict__ lse_arr,
    const int32_t* __restrict__ fresh,
    const int64_t* __restrict__ x, const float* __restrict__ t, float dt,
    const float* __restrict__ q_scale, const float* __restrict__ q_mask,
    const float* __restrict__ u, uint64_t seed, uint64_t offset,
    int64_t* __restrict__ x_next, int32_t* __restrict__ changed, int L, int V,
    int mask) {
  __shared__ float smv[4]; __shared__ int smi[4];
  const int64_t r = blockIdx.x;
  const int64_t tok = x[r];
  if (tok != mask) {  // carry-over, diffusion.py:609-610
    if (threadIdx.x == 0) x_next[r] = tok;
    return;
  }
  const int b = (int)(r / L);
  float ms, factor;
  race_weights(t, dt, q_scale, q_mask, b, factor, ms);
  const T* row = src + r * ld;
  float lse = 0.f;
  if constexpr (SRC == 1) {
    if (fresh && fresh[b]) {  // logits of this sample are new: (re)compute the row lse
      __shared__ float sml[9];
      lse = subs_row_lse<T>(row, V, mask, sml);
      if (threadIdx.x == 0) lse_arr[r] = lse;
    } else {
      lse = lse_arr[r];
    }
  }
  const uint64_t ctr0 = offset + (uint64_t)r * (uint64_t)((V + 3) >> 2);
  const int best = exact_row_race<T, SRC, PHILOX>(
      row, V, mask, lse, factor, ms, PHILOX ? nullptr : u + r * (int64_t)V, seed,
      ctr0, smv, smi);
  if (threadIdx.x == 0) {
    x_next[r] = best;
    if (best != mask && changed) changed[b] = 1;  // benign race: all write 1
  }
}

// ---- throughput path: SAME winner, cheaper arithmetic -------------------------
// Pass 1 ranks the columns by an approximate key with fewer transcendentals
// (hardware v_log_f32 / v_rcp_f32, <= 1 ulp each):
//   SRC 1 (logits), log2 domain:  k~_v = l_v log2(e) - log2(g~_v)     (2 x v_log_f32)
//                                 k~_mask = c_mask - log2(g~_mask),
//                                 c_mask = log2(ms) - log2(factor) + lse log2(e)
//   SRC 0 (probabilities):        k~_v = p_v factor / g~_v            (v_log_f32 + v_rcp_f32)
//   g~ = fma(-ln 2, v_log_f32(u + 1e-10f), 1e-10f)
// and keeps, per row, the best key B and the best key of any OTHER column C.
// Error budget (SRC 1; D_v = log2 of the defining key + lse log2(e) - log2(factor),
// in real arithmetic; ulp-sized terms as multiples of e = 2^-24):
//   pass 1:  |log2(g~) - log2(g)| <= 5 e max(1, |log2 g|) <= 121 e  (measured over ALL 2^24
//            uniforms: 4.47, tests/test_gpu_sampler.py::test_race_key_error_exhaustive; |log2 g| <= 24.1),
//            fma rounding e |k~|, fp32 log2(e) constant 0.5 e |l_v| log2(e);
//   defining arithmetic:  RN(l - lse) -> 1.45 e |l - lse|; expf 1 ulp, two roundings -> 5.8 e.
//   With |l_v| log2(e) <= |k~_v| + 24.1 and |l_v - lse| log2(e) <= |k~_v| + 24.1 + |lse| log2(e):
//   |k~_v - D_v| <= e (2.5 |k~_v| + 1.45 |lse| log2(e) + 165);  the mask column instead has the
//   three v_log_f32 and two additions of c_mask:  <= e (3 |k~| + 3 |lse| log2(e) + 210).
// If B - C > thr = 2^-20 (|B| + |lse| log2(e) + 64) = e (16 |B| + 16 |lse| log2(e) + 1024)
// (>= 2.7x the sum of two such bounds) the approximate winner has the strictly largest
// defining key, so it IS the reference argmax.  Otherwise (probability ~ thr ln 2 ~ 1e-4 of
// the rows in which two tokens compete, far fewer while the mask column dominates), or when
// the row is outside the analysed range (non-finite lse / weights, winner key near the fp32
// underflow range, where the defining arithmetic loses relative precision), the whole row is
// re-evaluated in the defining arithmetic (exact_row_race).  Hence: bit-identical token ids
// to ddpm_update_kernel on the same Philox stream, at the cost of the approximate pass.
// SRC 0: same scheme in the linear domain, relative threshold 2^-18 = 64 e (budget per key:
// g~ vs g 6 e -- measured over all 2^24 uniforms by the same test: 2.9 --, v_rcp_f32 2 e,
// four roundings 4 e -> 12 e).
//
// Prefilter (SRC 1, q_mask > 0).  While the mask column carries most of the mass (every
// ddpm_cache step but the last few) almost no column can beat ITS key, which is known
// before the row is read: bar = k~_mask.  -ln(s) >= 1 - s gives g_v >= 1 - u_v, and
// 1 - u_v > 2^(-1 - n_v) with n_v the number of leading bits of the 32 random bits r_v that
// equal its top bit (n_v leading ones: u_v < 1 - 2^-n_v; leading zeros: u_v < 1/2, g_v > ln 2).
// So k_v < l_v log2(e) + 1 + n_v, and a column with
//     l_v log2(e) + n_v < bar - margin - 1,   margin = 1 + 2^-18 (|bar| + |lse| log2(e) + 64)
// lies more than `margin` >= thr + (every rounding above) below B >= bar: it can neither win
// nor tie, in either arithmetic, and skips both logarithms (4.5 full-rate VALU operations
// instead of 2 quarter-rate + 8 full-rate ones).  A vector of 8 columns is evaluated in full
// as soon as the largest of its 8 bounds passes the test.
//
// Vector skip (SRC 1, same bar; round 3).  The bound n_v never needs the random bits to be
// large: g_v >= 1e-10 - ln(1 - 2^-24 + 1e-10) > 2^-24.1 for EVERY 24-bit uniform (in both
// arithmetics: measured over all 2^24 of them, test_race_key_error_exhaustive), i.e.
// k_v <= l_v log2(e) + 24.1 whatever is drawn.  A vector of 8 columns with
//     max_v l_v log2(e) + 24.5 < bar - margin - 1
// therefore fails the prefilter for every possible draw -- it would be skipped anyway -- and
// its two Philox blocks are not generated at all: the kernel is bound by exactly those
// integer multiplies (VALU issue 0.99), so every such vector is ~90 VALU operations saved
// for one v_max3 chain.  It fires for columns with p_v < ~2^-35 (t - dt) / (dt g_mask):
// almost never on a near-uniform output layer, for most of the vocabulary once the model's
// distribution is peaked (a trained checkpoint's tail tokens).  Same token ids by
// construction: only columns the prefilter would have dropped are dropped.
struct FastRace {
  float best = -INFINITY, sec = -INFINITY; int idx = 0x7fffffff;
  __device__ __forceinline__ void put(float k, int i) {
    // sec <= best always, so the median of (best, k, sec) is the new runner-up
    sec = __builtin_amdgcn_fmed3f(best, k, sec);
    const bool take = k > best;   // any tie ends in the exact re-evaluation
    best = take ? k : best; idx = take ? i : idx;
  }
};

__device__ __forceinline__ float fast_gumbel_norm(unsigned rbits) {
  const float u = Philox::u01(rbits);
  const float a = __builtin_amdgcn_logf(u + 1e-10f);              // log2
  return __builtin_fmaf(-0.6931471805599453f, a, 1e-10f);
}

template <int SRC>
__device__ __forceinline__ float fast_key(float x, unsigned rbits, float factor) {
  const float g = fast_gumbel_norm(rbits);
  if constexpr (SRC == 1) return __builtin_fmaf(x, 1.4426950408889634f, -__builtin_amdgcn_logf(g));
  else return x * factor * __builtin_amdgcn_rcpf(g);
}

// prefilter: upper bound l_v log2(e) + n_v of (k_v - 1) for one column (see above)
__device__ __forceinline__ float key_bound(float x, unsigned rbits) {
  // v_ffbh_i32: leading bits equal to the top bit; -1 (-> 4.29e9 as unsigned) if all are
  unsigned n;   // (no clang builtin for this instruction)
  asm("v_ffbh_i32 %0, %1" : "=v"(n) : "v"(rbits));
  return __builtin_fmaf(x, 1.4426950408889634f, (float)n);
}

template <typename T, int SRC>
__global__ __launch_bounds__(NT) void ddpm_update_fast_kernel(
    const T* __restrict__ src, int64_t ld, float* __restrict__ lse_arr,
    const int32_t* __restrict__ fresh,
    const int64_t* __restrict__ x, const float* __restrict__ t, float dt,
    const float* __restrict__ q_scale, const float* __restrict__ q_mask,
    uint64_t seed, uint64_t offset, int64_t* __restrict__ x_next,
    int32_t* __restrict__ changed, int L, int V, int mask,
    unsigned long long* __restrict__ n_exact) {
  __shared__ float smv[4]; __shared__ int smi[4];
  __shared__ int sm_near;
  constexpr int EPV = vec_traits<T>::EPV;
  const int64_t r = blockIdx.x;
  const int64_t tok = x[r];
  if (tok != mask) {  // carry-over, diffusion.py:609-610
    if (threadIdx.x == 0) x_next[r] = tok;
    return;
  }
  const int b = (int)(r / L);
  float ms, factor;
  race_weights(t, dt, q_scale, q_mask, b, factor, ms);
  const T* row = src + r * ld;
  const int G = (V + 3) >> 2;
  const uint64_t ctr0 = offset + (uint64_t)r * (uint64_t)G;
  float lse_r = 0.f;
  if constexpr (SRC == 1) {
    if (fresh && fresh[b]) {  // new logits: one extra pass for the row lse (L2-warm after it)
      __shared__ float sml[9];
      lse_r = subs_row_lse<T>(row, V, mask, sml);
      if (threadIdx.x == 0) lse_arr[r] = lse_r;
    } else {
      lse_r = lse_arr[r];
    }
  }
  if (threadIdx.x == 0) sm_near = 0;
  // rows outside the analysed range go straight to the defining arithmetic
  // (block-uniform: all operands are per-row values)
  bool exact = !(factor > 0x1p-60f && factor < 0x1p60f) || !(ms < 0x1p60f) ||
               !(fabsf(lse_r) < 0x1p60f);
  int best = 0;
  if (!exact) {
    const float lse2 = lse_r * 1.4426950408889634f;
    // key of the mask column (diffusion.py:606: q = ms), from its own random bits
    float kmask;
    {
      const uint4 rn = Philox::gen(seed, ctr0 + (uint64_t)(mask >> 2));
      const unsigned rr[4] = {rn.x, rn.y, rn.z, rn.w};
      const float g = fast_gumbel_norm(rr[mask & 3]);
      if constexpr (SRC == 1) {
        // q_mask <= 0 never beats a positive key
        kmask = ms > 0.f ? (__builtin_amdgcn_logf(ms) - __builtin_amdgcn_logf(factor) + lse2) -
                               __builtin_amdgcn_logf(g)
                         : -INFINITY;
      } else {
        kmask = ms * __builtin_amdgcn_rcpf(g);
      }
    }
    FastRace fr;
    if (threadIdx.x == 0) fr.put(kmask, mask);
    float acc = 0.f;   // SRC 0: NaN / inf detector (any non-finite key poisons the sum)
    // SRC 1: a column passes when l_v log2(e) + n_v >= c_bar = bar - margin - 1;
    // -inf (no usable bar) lets every column through
    const float c_bar = (SRC == 1 && kmask > -INFINITY)
        ? kmask - 2.f - 0x1p-18f * (fabsf(kmask) + fabsf(lse2) + 64.f) : -INFINITY;
    // the mask column's own key was put above; its row entry never competes
    // (SRC 0: 0 rather than -inf so that it does not trip the non-finite detector)
    auto key_of = [&](float xv, unsigned rbits, int v) -> float {
      return v == mask ? (SRC == 1 ? -INFINITY : 0.f) : fast_key<SRC>(xv, rbits, factor);
    };
    if (row_head(row, V) == 0) {
      const int nvec = V / EPV;
      const uint4* vp = reinterpret_cast<const uint4*>(row);
      // Vector skip: decided per WAVE from its lanes' first vectors (three quarters of the 64
      // must be skippable), and the loop exists twice, so that waves that do not skip -- every
      // wave on a near-uniform output layer -- run exactly the code without the test.  A speed
      // heuristic only: tested or not, the same columns reach the race.
      auto vec_loop = [&](auto skip_c, auto probe_c, int i0) -> bool {
        constexpr bool SKIP = decltype(skip_c)::value, PROBE = decltype(probe_c)::value;
        bool probe = false;
        for (int i = i0; i < nvec; i += NT) {
          float f[EPV];
          unpack<T>(vp[i], f);
          const int v0 = i * EPV;
          if constexpr (SKIP || PROBE) {
            // no draw can lift any of these columns to the bar: skip their random numbers too
            // (a NaN logit drops out of the maximum: finite logits are this path's precondition,
            // include/mdlm_hip.h)
            float mx = f[0];
#pragma unroll
            for (int k = 1; k < EPV; ++k) mx = fmaxf(mx, f[k]);
            const bool sk = __builtin_fmaf(mx, 1.4426950408889634f, 24.5f) < c_bar;
            if constexpr (PROBE) probe = sk;
            else if (sk) continue;
          }
          unsigned rb[EPV];
#pragma unroll
          for (int gq = 0; gq < EPV / 4; ++gq) {
            const uint4 rn = Philox::gen(seed, ctr0 + (uint64_t)((v0 >> 2) + gq));
            rb[gq * 4 + 0] = rn.x; rb[gq * 4 + 1] = rn.y;
            rb[gq * 4 + 2] = rn.z; rb[gq * 4 + 3] = rn.w;
          }
          bool go = true;
          if constexpr (SRC == 1) {
            float hi = key_bound(f[0], rb[0]);
#pragma unroll
            for (int k = 1; k < EPV; ++k) hi = fmaxf(hi, key_bound(f[k], rb[k]));
            go = hi >= c_bar;
          }
          if (go) {
            float key[EPV];
#pragma unroll
            for (int k = 0; k < EPV; ++k) key[k] = key_of(f[k], rb[k], v0 + k);
#pragma unroll
            for (int k = 0; k < EPV; ++k) {
              fr.put(key[k], v0 + k);
              if constexpr (SRC == 0) acc += key[k];
            }
          }
          if constexpr (PROBE) break;   // the first vector only
        }
        return probe;
      };
      if constexpr (SRC == 1) {
        // first vector of every lane: evaluated in full AND probed for skippability
        const bool sk0 = vec_loop(std::false_type{}, std::true_type{}, threadIdx.x);
        const bool wave_skips = c_bar > -INFINITY && __popcll(__ballot(sk0)) >= 48;
        if (wave_skips) vec_loop(std::true_type{}, std::false_type{}, threadIdx.x + NT);
        else vec_loop(std::false_type{}, std::false_type{}, threadIdx.x + NT);
      } else {
        vec_loop(std::false_type{}, std::false_type{}, threadIdx.x);
      }
      for (int v = nvec * EPV + threadIdx.x; v < V; v += NT) {
        const uint4 rn = Philox::gen(seed, ctr0 + (uint64_t)(v >> 2));
        const unsigned rr[4] = {rn.x, rn.y, rn.z, rn.w};
        const float k = key_of(elem<T>::load(row + v), rr[v & 3], v);
        fr.put(k, v);
        if constexpr (SRC == 0) acc += k;
      }
    } else {
      for (int g = threadIdx.x; g < G; g += NT) {
        const uint4 rn = Philox::gen(seed, ctr0 + (uint64_t)g);
        const unsigned rr[4] = {rn.x, rn.y, rn.z, rn.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int v = g * 4 + k;
          if (v < V) {
            const float kk = key_of(elem<T>::load(row + v), rr[k], v);
            fr.put(kk, v);
            if constexpr (SRC == 0) acc += kk;
          }
        }
      }
    }
    ArgMax am; am.v = fr.best; am.i = fr.idx;
    float B;
    best = block_argmax(am, smv, smi, &B);
    // runner-up seen from this thread: its own best unless it holds the winner
    const float C = fr.idx == best ? fr.sec : fr.best;
    bool near;
    if constexpr (SRC == 1) {
      const float thr = 0x1p-20f * (fabsf(B) + fabsf(lse2) + 64.f);
      near = !(C < B - thr);
      // log2 of the winner's probability is >= B - lse2 - 24.1; keep p, p*factor and
      // the key well inside the normal fp32 range
      if (threadIdx.x == 0 &&
          !(B - lse2 + fminf(0.f, __builtin_amdgcn_logf(factor)) > -90.f)) near = true;
    } else {
      near = !(C < B - 0x1p-18f * fabsf(B)) || !(fabsf(acc) < INFINITY);
      if (threadIdx.x == 0 && !(B > 0x1p-90f)) near = true;
    }
    if (near) sm_near = 1;   // benign race: all write 1
    __syncthreads();
    exact = sm_near != 0;
  }
  if (exact) {   // block-uniform
    best = exact_row_race<T, SRC, true>(row, V, mask, lse_r, factor, ms, nullptr,
                                        seed, ctr0, smv, smi);
    if (threadIdx.x == 0 && n_exact) atomicAdd(n_exact, 1ull);
  }
  if (threadIdx.x == 0) {
    x_next[r] = best;
    if (best != mask && changed) changed[b] = 1;  // benign race: all write 1
  }
}

// Diagnostic for the error budget above (tests only; no reference counterpart):
// for uniform bit patterns rbits[i] (u = (rbits >> 8) 2^-24) writes the pass-1 values
// g~ and log2(g~) and the defining g = 1e-10f - RN(ln(u + 1e-10f)).
__global__ void race_key_probe_kernel(const uint32_t* __restrict__ rbits,
                                      float* __restrict__ g_fast,
                                      float* __restrict__ log2_g_fast,
                                      float* __restrict__ g_exact, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  g_fast[i] = fast_gumbel_norm(rbits[i]);
  log2_g_fast[i] = __builtin_amdgcn_logf(fast_gumbel_norm(rbits[i]));
  g_exact[i] = gumbel_norm_exact(Philox::u01(rbits[i]));
}

// Discrete-time SUBS loss term, diffusion.py:329-358.  For SUBS outputs the
// mask-column probability exp(model_output[..., mask]) is exactly 0 (the column
// holds -1e6 - lse or -1e6), so only log p(x0) is needed; every other factor is
// restated literally, including the 0/0 = NaN the reference produces for the
// samples with t == 1/T (t - dt == 0).
__global__ void d3pm_loss_kernel(const float* __restrict__ neg_logp_x0,
                                 const int64_t* __restrict__ xt,
                                 const float* __restrict__ t, int T,
                                 const float* __restrict__ attn_mask,
                                 float* __restrict__ out, int64_t n, int L, int mask) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int b = (int)(i / L);
  const float dt = (float)(1.0 / (double)T);                          // 330
  const float tt = fminf(fmaxf(t[b], 0.f), (float)(1. - 1e-4));       // 335
  const float alpha_t = 1.f - tt;                                      // 336
  const float alpha_s = 1.f - (tt - dt);                               // 337
  const float x_m = 0.f;                                               // 341-342
  const float lp = -neg_logp_x0[i];                                    // 339-340
  const float c1 = dt / tt;                                            // 344
  const float nr = logf(__fadd_rn(__fdiv_rn(__fmul_rn(alpha_t, x_m), tt), 1.f));        // 345
  const float c2 = 1.f - dt / tt;                                      // 348
  const float dr2 = logf(__fadd_rn(__fdiv_rn(__fmul_rn(alpha_s, x_m), tt - dt), 1.f));  // 350
  const float lvb = __fadd_rn(__fmul_rn(c1, nr - lp), __fmul_rn(c2, nr - dr2));         // 352-354
  float o = (float)T * __fmul_rn(lvb, xt[i] == mask ? 1.f : 0.f);     // 356-358
  if (attn_mask) o = __fmul_rn(o, attn_mask[i]);                       // 908
  out[i] = o;
}

__global__ void philox_uniform_kernel(float* out, int64_t rows, int V,
                                      uint64_t seed, uint64_t offset) {
  const uint64_t G = (uint64_t)((V + 3) >> 2);
  const int64_t r = blockIdx.x;
  for (int g = threadIdx.x; g < (int)G; g += blockDim.x) {
    const uint4 rn = Philox::gen(seed, offset + (uint64_t)r * G + (uint64_t)g);
    const unsigned rr[4] = {rn.x, rn.y, rn.z, rn.w};
    for (int k = 0; k < 4; ++k)
      if (g * 4 + k < V) out[r * (int64_t)V + g * 4 + k] = Philox::u01(rr[k]);
  }
}

// ------------------------------------------------------------- (c1) q_xt
__global__ void q_xt_kernel(const int64_t* __restrict__ x0,
                            const float* __restrict__ move_chance,
                            const float* __restrict__ u, uint64_t seed,
                            uint64_t offset, int64_t* __restrict__ xt, int64_t n,
                            int L, int mask) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float ui;
  if (u) ui = u[i];
  else {
    const uint4 rn = Philox::gen(seed, offset + (uint64_t)(i >> 2));
    const unsigned rr[4] = {rn.x, rn.y, rn.z, rn.w};
    ui = Philox::u01(rr[i & 3]);
  }
  xt[i] = (ui < move_chance[i / L]) ? (int64_t)mask : x0[i];  // diffusion.py:583-585
}

// ----------------------------------------------------------- (c2) SUBS NLL
template <typename T>
__global__ __launch_bounds__(NT) void subs_nll_kernel(
    const T* __restrict__ logits, int64_t ld, const int64_t* __restrict__ xt,
    const int64_t* __restrict__ x0, const float* __restrict__ weight,
    const float* __restrict__ attn_mask, float* __restrict__ nll, int L, int V,
    int mask) {
  __shared__ float sm[9];
  const int64_t r = blockIdx.x;
  const int64_t tok = xt[r], tgt = x0[r];
  const float w = weight[r / L];
  const float am = attn_mask ? attn_mask[r] : 1.f;
  float logp;
  if (tok != mask) {
    logp = (tgt == tok) ? 0.f : NEG_INF;  // diffusion.py:274-276
  } else {
    const T* row = logits + r * ld;
    const float lse = subs_row_lse<T>(row, V, mask, sm);
    float x = elem<T>::load(row + tgt);
    if (tgt == mask) x += NEG_INF;
    logp = x - lse;
  }
  // diffusion.py:893-894 then 908
  if (threadIdx.x == 0) nll[r] = (-logp * w) * am;
}

__global__ __launch_bounds__(1024) void nll_sums_kernel(
    const float* __restrict__ nll, const float* __restrict__ attn_mask,
    double* __restrict__ sums, int64_t n) {
  __shared__ double s0[1024], s1[1024];
  double a = 0.0, c = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    a += (double)nll[i];
    c += attn_mask ? (double)attn_mask[i] : 1.0;
  }
  s0[threadIdx.x] = a; s1[threadIdx.x] = c;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (threadIdx.x < o) { s0[threadIdx.x] += s0[threadIdx.x + o]; s1[threadIdx.x] += s1[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { sums[0] += s0[0]; sums[1] += s1[0]; }
}

__global__ void compact_active_kernel(const int32_t* __restrict__ changed,
                                      int32_t* __restrict__ active,
                                      int32_t* __restrict__ n_active,
                                      int32_t* __restrict__ n_rows, int B, int L,
                                      int all_or_nothing,
                                      int64_t* __restrict__ nfe_accum,
                                      int32_t* __restrict__ fresh) {
  // single wave; ordered compaction via ballot, 64 samples per pass
  bool any = false;
  if (all_or_nothing) {
    for (int b0 = 0; b0 < B; b0 += 64) {
      const int b = b0 + threadIdx.x;
      any = any || (__ballot(b < B && changed[b] != 0) != 0ull);
    }
  }
  int base = 0;
  for (int b0 = 0; b0 < B; b0 += 64) {
    const int b = b0 + threadIdx.x;
    const bool on = b < B && (any || changed[b] != 0);
    const unsigned long long m = __ballot(on);
    if (on) active[base + __popcll(m & ((1ull << threadIdx.x) - 1ull))] = b;
    if (fresh && b < B) fresh[b] = on ? 1 : 0;
    base += __popcll(m);
  }
  if (threadIdx.x == 0) {
    *n_active = base; *n_rows = base * L;
    // atomic: the stream groups of one sampling step compact concurrently
    if (nfe_accum) atomicAdd(reinterpret_cast<unsigned long long*>(nfe_accum), (unsigned long long)base);
  }
}

// Row lists for the masked rows of the active samples (one workgroup, ordered):
// gather[k]  = compacted workspace row (i * L + pos, i = index into active[])
// scatter[k] = logits-cache row (active[i] * L + pos)
// padded with (0, -1) up to a multiple of `pad`; *n_out = padded count.
__global__ __launch_bounds__(1024) void masked_row_map_kernel(
    const int64_t* __restrict__ x, const int32_t* __restrict__ active,
    const int32_t* __restrict__ n_active, int L, int mask, int pad,
    int32_t* __restrict__ gather, int32_t* __restrict__ scatter,
    int32_t* __restrict__ n_out) {
  // one workgroup; every thread owns RPT consecutive rows per sweep, so a
  // sweep covers 8192 rows with one wave scan + one cross-wave scan
  constexpr int RPT = 8;
  __shared__ int wsum[16];
  __shared__ int base_s;
  const int n_rows = *n_active * L;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (threadIdx.x == 0) base_s = 0;
  __syncthreads();
  for (int r0 = 0; r0 < n_rows; r0 += 1024 * RPT) {
    const int rbeg = r0 + threadIdx.x * RPT;
    int dst[RPT];
    unsigned on = 0;
#pragma unroll
    for (int c = 0; c < RPT; ++c) {
      const int r = rbeg + c;
      dst[c] = -1;
      if (r < n_rows) {
        const int i = r / L, pos = r - i * L;
        dst[c] = active[i] * L + pos;
        if (x[dst[c]] == mask) on |= 1u << c;
      }
    }
    const int cnt = __popc(on);
    int incl = cnt;   // inclusive wave scan of the per-thread counts
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(incl, d);
      if (lane >= d) incl += o;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int off = base_s + incl - cnt;
    for (int k = 0; k < w; ++k) off += wsum[k];
#pragma unroll
    for (int c = 0; c < RPT; ++c)
      if (on & (1u << c)) { gather[off] = rbeg + c; scatter[off] = dst[c]; ++off; }
    __syncthreads();
    if (threadIdx.x == 0) {
      int t = 0;
      for (int k = 0; k < 16; ++k) t += wsum[k];
      base_s += t;
    }
    __syncthreads();
  }
  const int n = base_s;
  const int np = (n + pad - 1) / pad * pad;
  for (int k = n + threadIdx.x; k < np; k += 1024) { gather[k] = 0; scatter[k] = -1; }
  if (threadIdx.x == 0) *n_out = np;
}

// ---- opt-in two-stage ("lazy") ddpm_cache step --------------------------------
// The reference race over {q_v = p_v (t - s)} U {q_mask = s} leaves a masked
// position masked with probability s/t and otherwise draws its token from p_x0
// (sum_v p_v = 1, the mask column of a SUBS output is 0).  Stage 1 makes that
// Bernoulli decision with ONE uniform per masked position and lists the
// positions that unmask; stage 2 (lazy_race_kernel) draws their tokens from
// logits computed for those rows only.  Same distribution as
// diffusion.py:592-610, different mapping from random numbers to outcomes, so
// it is NOT bit-identical to the reference on a shared uniform tensor: opt-in.
__global__ __launch_bounds__(1024) void lazy_select_kernel(
    const int64_t* __restrict__ x, const float* __restrict__ t, float dt,
    uint64_t seed, uint64_t offset, int n_rows, int L, int mask, int pad,
    int64_t* __restrict__ x_next, int32_t* __restrict__ rows_out,
    int32_t* __restrict__ n_out, int32_t* __restrict__ changed) {
  constexpr int RPT = 8;
  __shared__ int wsum[16];
  __shared__ int base_s;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (threadIdx.x == 0) base_s = 0;
  __syncthreads();
  for (int r0 = 0; r0 < n_rows; r0 += 1024 * RPT) {
    const int rbeg = r0 + threadIdx.x * RPT;   // RPT = 8 rows = two Philox groups
    unsigned on = 0;
    if (rbeg < n_rows) {
      const uint4 ra = Philox::gen(seed, offset + (uint64_t)(rbeg >> 2));
      const uint4 rb = Philox::gen(seed, offset + (uint64_t)(rbeg >> 2) + 1);
      const unsigned rr[8] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
#pragma unroll
      for (int c = 0; c < RPT; ++c) {
        const int r = rbeg + c;
        if (r < n_rows) {
          const int64_t tok = x[r];
          x_next[r] = tok;
          if (tok == mask) {
            const float mt = t[r / L];
            const float ms = mt - dt;                 // diffusion.py:599
            // P(stay masked) = ms / mt  (diffusion.py:605-607)
            if (!(Philox::u01(rr[c]) * mt < ms)) on |= 1u << c;
          }
        }
      }
    }
    const int cnt = __popc(on);
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(incl, d);
      if (lane >= d) incl += o;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int off = base_s + incl - cnt;
    for (int k = 0; k < w; ++k) off += wsum[k];
#pragma unroll
    for (int c = 0; c < RPT; ++c)
      if (on & (1u << c)) {
        rows_out[off++] = rbeg + c;
        changed[(rbeg + c) / L] = 1;   // benign race: all write 1
      }
    __syncthreads();
    if (threadIdx.x == 0) {
      int tt = 0;
      for (int k = 0; k < 16; ++k) tt += wsum[k];
      base_s += tt;
    }
    __syncthreads();
  }
  const int n = base_s;
  const int np = (n + pad - 1) / pad * pad;
  for (int k = n + threadIdx.x; k < np; k += 1024) rows_out[k] = 0;  // padding rows: computed, never used
  if (threadIdx.x == 0) { n_out[0] = np; n_out[1] = n; }
}

// Stage 2: categorical draw from softmax(logits row) over v != mask for the
// listed rows (dense logits row i belongs to position rows[i]).
template <typename T>
__global__ __launch_bounds__(NT) void lazy_race_kernel(
    const T* __restrict__ logits, int64_t ld, const int32_t* __restrict__ rows,
    const int32_t* __restrict__ n_rows, uint64_t seed, uint64_t offset,
    int64_t* __restrict__ x_next, int V, int mask) {
  __shared__ float smv[4]; __shared__ int smi[4];
  const int i = blockIdx.x;
  if (i >= n_rows[1]) return;
  const T* row = logits + (int64_t)i * ld;
  const uint64_t G = (uint64_t)((V + 3) >> 2);
  const uint64_t ctr0 = offset + (uint64_t)i * G;
  FastRace fr;
  for (int g = threadIdx.x; g < (int)G; g += NT) {
    const uint4 rn = Philox::gen(seed, ctr0 + (uint64_t)g);
    const unsigned rr[4] = {rn.x, rn.y, rn.z, rn.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int v = g * 4 + k;
      if (v < V && v != mask) fr.put(fast_key<1>(elem<T>::load(row + v), rr[k], 1.f), v);
    }
  }
  ArgMax am; am.v = fr.best; am.i = fr.idx;
  const int best = block_argmax(am, smv, smi);
  if (threadIdx.x == 0) x_next[rows[i]] = best;
}

static bool is16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace mdlm

using namespace mdlm;

#define ST(s) ((hipStream_t)(s))

extern "C" int mdlm_subs_logprobs(const void* logits, int64_t ld_in, int dtype_in,
                                  const int64_t* xt, void* out, int64_t ld_out,
                                  int dtype_out, int64_t rows, int V,
                                  int mask_index, int as_probs, void* stream) {
  if (rows == 0) return 0;  // empty batch: nothing to do (empty tensors carry null pointers)
  MDLM_CHECK_ARG(logits && xt && out, "subs_logprobs: null pointer");
  MDLM_CHECK_ARG(rows >= 0 && rows < (1ll << 31) && V > 0 && ld_in >= V && ld_out >= V,
                 "subs_logprobs: bad shape rows=%lld V=%d ld_in=%lld ld_out=%lld",
                 (long long)rows, V, (long long)ld_in, (long long)ld_out);
  MDLM_CHECK_ARG(mask_index >= 0 && mask_index < V, "subs_logprobs: mask_index %d outside [0,%d)", mask_index, V);
  MDLM_CHECK_ARG((dtype_in == MDLM_F32 || dtype_in == MDLM_BF16) &&
                 (dtype_out == MDLM_F32 || dtype_out == MDLM_BF16), "subs_logprobs: bad dtype");
  if (rows == 0) return 0;
  dim3 g((unsigned)rows), b(NT);
  const size_t es = dtype_in == MDLM_F32 ? 4 : 2;
  const bool vec = dtype_in == dtype_out && is16(logits) && is16(out) &&
                   (ld_in * es) % 16 == 0 && (ld_out * es) % 16 == 0;
#define MDLM_SL(P) \
  if (dtype_in == MDLM_F32 && dtype_out == MDLM_F32) { \
    if (vec) subs_logprobs_vec_kernel<float, P><<<g, b, 0, ST(stream)>>>((const float*)logits, ld_in, xt, (float*)out, ld_out, V, mask_index); \
    else subs_logprobs_kernel<float, float, P><<<g, b, 0, ST(stream)>>>((const float*)logits, ld_in, xt, (float*)out, ld_out, V, mask_index); \
  } else if (dtype_in == MDLM_BF16 && dtype_out == MDLM_BF16) { \
    if (vec && V / 8 <= REG_VEC * NT) subs_logprobs_reg_kernel<P><<<g, b, 0, ST(stream)>>>((const bf16_t*)logits, ld_in, xt, (bf16_t*)out, ld_out, V, mask_index); \
    else if (vec) subs_logprobs_vec_kernel<bf16_t, P><<<g, b, 0, ST(stream)>>>((const bf16_t*)logits, ld_in, xt, (bf16_t*)out, ld_out, V, mask_index); \
    else subs_logprobs_kernel<bf16_t, bf16_t, P><<<g, b, 0, ST(stream)>>>((const bf16_t*)logits, ld_in, xt, (bf16_t*)out, ld_out, V, mask_index); \
  } else if (dtype_in == MDLM_BF16) { \
    subs_logprobs_kernel<bf16_t, float, P><<<g, b, 0, ST(stream)>>>((const bf16_t*)logits, ld_in, xt, (float*)out, ld_out, V, mask_index); \
  } else { \
    subs_logprobs_kernel<float, bf16_t, P><<<g, b, 0, ST(stream)>>>((const float*)logits, ld_in, xt, (bf16_t*)out, ld_out, V, mask_index); \
  }
  if (as_probs) { MDLM_SL(true) } else { MDLM_SL(false) }
#undef MDLM_SL
  MDLM_LAUNCH_CHECK();
  return 0;
}

extern "C" int mdlm_subs_row_lse(const void* logits, int64_t ld, int dtype,
                                 const int64_t* xt, float* lse, int64_t rows,
                                 int V, int mask_index,
                                 const int32_t* sample_map, int L,
                                 const int32_t* n_rows_dev, void* stream) {
  if (rows == 0) return 0;
  MDLM_CHECK_ARG(logits && xt && lse, "subs_row_lse: null pointer");
  MDLM_CHECK_ARG(rows >= 0 && rows < (1ll << 31) && V > 0 && ld >= V, "subs_row_lse: bad shape");
  MDLM_CHECK_ARG(mask_index >= 0 && mask_index < V, "subs_row_lse: mask_index out of range");
  MDLM_CHECK_ARG(dtype == MDLM_F32 || dtype == MDLM_BF16, "subs_row_lse: bad dtype");
  if (rows == 0) return 0;
  MDLM_CHECK_ARG(!sample_map || L > 0, "subs_row_lse: sample_map needs L > 0");
  if (dtype == MDLM_F32) subs_row_lse_kernel<float><<<(unsigned)rows, NT, 0, ST(stream)>>>((const float*)logits, ld, xt, lse, V, mask_index, sample_map, L, n_rows_dev);
  else subs_row_lse_kernel<bf16_t><<<(unsigned)rows, NT, 0, ST(stream)>>>((const bf16_t*)logits, ld, xt, lse, V, mask_index, sample_map, L, n_rows_dev);
  MDLM_LAUNCH_CHECK();
  return 0;
}

extern "C" int mdlm_subs_argmax(const void* logits, int64_t ld, int dtype,
                                const int64_t* xt, int64_t* out, int64_t rows,
                                int V, int mask_index, void* stream) {
  if (rows == 0) return 0;
  MDLM_CHECK_ARG(logits && xt && out, "subs_argmax: null pointer");
  MDLM_CHECK_ARG(rows >= 0 && rows < (1ll << 31) && V > 0 && ld >= V, "subs_argmax: bad shape");
  MDLM_CHECK_ARG(mask_index >= 0 && mask_index < V, "subs_argmax: mask_index out of range");
  MDLM_CHECK_ARG(dtype == MDLM_F32 || dtype == MDLM_BF16, "subs_argmax: bad dtype");
  if (rows == 0) return 0;
  if (dtype == MDLM_F32) subs_argmax_kernel<float><<<(unsigned)rows, NT, 0, ST(stream)>>>((const float*)logits, ld, xt, out, V, mask_index);
  else subs_argmax_kernel<bf16_t><<<(unsigned)rows, NT, 0, ST(stream)>>>((const bf16_t*)logits, ld, xt, out, V, mask_index);
  MDLM_LAUNCH_CHECK();
  return 0;
}

template <int SRC>
static int launch_update(const void* src, int64_t ld, int dtype, float* lse,
                         const int32_t* fresh,
                         const int64_t* x, const float* t, double dt,
                         const float* q_scale, const float* q_mask,
                         const float* u, uint64_t seed, uint64_t offset,
                         int64_t* x_next, int32_t* changed, int B, int L, int V,
                         int mask_index, int fast_math, uint64_t* n_exact_rows,
                         void* stream) {
  const char* nm = q_scale ? "subs_race_update" : SRC ? "subs_ddpm_update" : "ddpm_cache_update";
  if (B == 0) return 0;
  MDLM_CHECK_ARG(src && x && (t || q_scale) && x_next, "%s: null pointer", nm);
  MDLM_CHECK_ARG(!q_scale == !q_mask, "%s: q_scale and q_mask go together", nm);
  MDLM_CHECK_ARG(SRC == 0 || lse, "%s: lse is required", nm);
  MDLM_CHECK_ARG(B >= 0 && L > 0 && V > 0 && ld >= V && (int64_t)B * L < (1ll << 31), "%s: bad shape B=%d L=%d V=%d ld=%lld", nm, B, L, V, (long long)ld);
  MDLM_CHECK_ARG(mask_index >= 0 && mask_index < V, "%s: mask_index out of range", nm);
  MDLM_CHECK_ARG(dtype == MDLM_F32 || dtype == MDLM_BF16, "%s: bad dtype", nm);
  hipStream_t s = ST(stream);
  if (changed) {
    hipError_t e = hipMemsetAsync(changed, 0, sizeof(int32_t) * (size_t)B, s);
    if (e != hipSuccess) { set_error("%s: memset failed: %s", nm, hipGetErrorString(e)); return (int)e; }
  }
  const unsigned rows = (unsigned)((int64_t)B * L);
  const float dtf = (float)dt;  // python float -> fp32 scalar, as torch does
  unsigned long long* nx = reinterpret_cast<unsigned long long*>(n_exact_rows);
#define MDLM_UPD(T, PH) \
  ddpm_update_kernel<T, SRC, PH><<<rows, NT, 0, s>>>((const T*)src, ld, lse, fresh, x, t, dtf, q_scale, q_mask, u, seed, offset, x_next, changed, L, V, mask_index)
  const bool ph = (u == nullptr), fast = fast_math != 0;
  if (ph && fast) {
    if (dtype == MDLM_F32)
      ddpm_update_fast_kernel<float, SRC><<<rows, NT, 0, s>>>((const float*)src, ld, lse, fresh, x, t, dtf, q_scale, q_mask, seed, offset, x_next, changed, L, V, mask_index, nx);
    else
      ddpm_update_fast_kernel<bf16_t, SRC><<<rows, NT, 0, s>>>((const bf16_t*)src, ld, lse, fresh, x, t, dtf, q_scale, q_mask, seed, offset, x_next, changed, L, V, mask_index, nx);
  } else if (dtype == MDLM_F32) {
    if (ph) MDLM_UPD(float, true); else MDLM_UPD(float, false);
  } else {
    if (ph) MDLM_UPD(bf16_t, true); else MDLM_UPD(bf16_t, false);
  }
#undef MDLM_UPD
  MDLM_LAUNCH_CHECK();
  return 0;
}

extern "C" int mdlm_ddpm_cache_update(const void* p_x0, int64_t ld, int dtype,
                                      const int64_t* x, const float* t, double dt,
                                      const float* u, uint64_t seed, uint64_t offset,
                                      int64_t* x_next, int32_t* changed, int B,
                                      int L, int V, int mask_index, int fast_math,
                                      uint64_t* n_exact_rows, void* stream) {
  return launch_update<0>(p_x0, ld, dtype, nullptr, nullptr, x, t, dt, nullptr, nullptr, u, seed, offset, x_next, changed, B, L, V, mask_index, fast_math, n_exact_rows, stream);
}

extern "C" int mdlm_subs_ddpm_update(const void* logits, int64_t ld, int dtype,
                                     float* lse, const int32_t* fresh,
                                     const int64_t* x,
                                     const float* t, double dt, const float* u,
                                     uint64_t seed, uint64_t offset,
                                     int64_t* x_next, int32_t* changed, int B,
                                     int L, int V, int mask_index, int fast_math,
                                     uint64_t* n_exact_rows, void* stream) {
  return launch_update<1>(logits, ld, dtype, lse, fresh, x, t, dt, nullptr, nullptr, u, seed, offset, x_next, changed, B, L, V, mask_index, fast_math, n_exact_rows, stream);
}

extern "C" int mdlm_subs_race_update(const void* logits, int64_t ld, int dtype,
                                     float* lse, const int32_t* fresh,
                                     const int64_t* x, const float* q_scale,
                                     const float* q_mask, const float* u,
                                     uint64_t seed, uint64_t offset,
                                     int64_t* x_next, int32_t* changed, int B,
                                     int L, int V, int mask_index, int fast_math,
                                     uint64_t* n_exact_rows, void* stream) {
  MDLM_CHECK_ARG(q_scale && q_mask, "subs_race_update: q_scale / q_mask are required");
  return launch_update<1>(logits, ld, dtype, lse, fresh, x, nullptr, 0.0, q_scale, q_mask, u, seed, offset, x_next, changed, B, L, V, mask_index, fast_math, n_exact_rows, stream);
}

extern "C" int mdlm_race_key_probe(const uint32_t* rbits, float* g_fast,
                                   float* log2_g_fast, float* g_exact, int64_t n,
                                   void* stream) {
  MDLM_CHECK_ARG(rbits && g_fast && log2_g_fast && g_exact && n >= 0, "race_key_probe: bad args");
  if (n == 0) return 0;
  race_key_probe_kernel<<<(unsigned)((n + 255) / 256), 256, 0, ST(stream)>>>(rbits, g_fast, log2_g_fast, g_exact, n);
  MDLM_LAUNCH_CHECK();
  return 0;
}

extern "C" int mdlm_philox_uniform(float* out, int64_t rows, int V, uint64_t seed,
                                   uint64_t offset, void* stream) {
  MDLM_CHECK_ARG(out && rows >= 0 && rows < (1ll << 31) && V > 0, "philox_uniform: bad args");
  if (rows == 0) return 0;
  philox_uniform_kernel<<<(unsigned)rows, 256, 0, ST(stream)>>>(out, rows, V, seed, offset);
  MDLM_LAUNCH_CHECK();
  return 0;
}

extern "C" int mdlm_q_xt(const int64_t* x0, const float* move_chance,
                         const float* u, uint64_t seed, uint64_t offset,
                         int64_t* xt, int B, int L, int mask_index, void* stream) {
  if (B == 0) return 0;
  MDLM_CHECK_ARG(x0 && move_chance && xt, "q_xt: null pointer");
  MDLM_CHECK_ARG(B >= 0 && L > 0, "q_xt: bad shape");
  const int64_t n = (int64_t)B * L;
  if (n == 0) return 0;
  q_xt_kernel<<<(unsigned)((n + 255) / 256), 256, 0, ST(stream)>>>(x0, move_chance, u, seed, offset, xt, n, L, mask_index);
  MDLM_LAUNCH_CHECK();
  return 0;
}

namespace mdlm {
// the same with the row lse already known (from the vocabulary GEMM's epilogue,
// mdlm_lse_combine): one gathered logit per row, one thread per row
template <typename T>
__global__ __launch_bounds__(256) void subs_nll_lse_kernel(
    const T* __restrict__ logits, int64_t ld, const int64_t* __restrict__ xt,
    const int64_t* __restrict__ x0, const float* __restrict__ weight,
    const float* __restrict__ attn_mask, const float* __restrict__ row_lse,
    float* __restrict__ nll, int64_t rows, int L, int mask) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  const int64_t tok = xt[r], tgt = x0[r];
  float logp;
  if (tok != mask) {
    logp = (tgt == tok) ? 0.f : NEG_INF;
  } else {
    float x = elem<T>::load(logits + r * ld + tgt);
    if (tgt == mask) x += NEG_INF;
    logp = x - row_lse[r];
  }
  nll[r] = (-logp * weight[r / L]) * (attn_mask ? attn_mask[r] : 1.f);
}
}  // namespace mdlm

extern "C" int mdlm_subs_nll(const void* logits, int64_t ld, int dtype,
                             const int64_t* xt, const int64_t* x0,
                             const float* weight, const float* attn_mask,
                             const float* row_lse, float* nll, int B, int L, int V,
                             int mask_index, void* stream) {
  if (B == 0) return 0;
  MDLM_CHECK_ARG(logits && xt && x0 && weight && nll, "subs_nll: null pointer");
  MDLM_CHECK_ARG(B >= 0 && L > 0 && V > 0 && ld >= V && (int64_t)B * L < (1ll << 31), "subs_nll: bad shape");
  MDLM_CHECK_ARG(mask_index >= 0 && mask_index < V, "subs_nll: mask_index out of range");
  MDLM_CHECK_ARG(dtype == MDLM_F32 || dtype == MDLM_BF16, "subs_nll: bad dtype");
  if (B == 0) return 0;
  const unsigned rows = (unsigned)((int64_t)B * L);
  if (row_lse) {
    const unsigned nb = (rows + 255) / 256;
    if (dtype == MDLM_F32) subs_nll_lse_kernel<float><<<nb, 256, 0, ST(stream)>>>((const float*)logits, ld, xt, x0, weight, attn_mask, row_lse, nll, rows, L, mask_index);
    else subs_nll_lse_kernel<bf16_t><<<nb, 256, 0, ST(stream)>>>((const bf16_t*)logits, ld, xt, x0, weight, attn_mask, row_lse, nll, rows, L, mask_index);
    MDLM_LAUNCH_CHECK();
    return 0;
  }
  if (dtype == MDLM_F32) subs_nll_kernel<float><<<rows, NT, 0, ST(stream)>>>((const float*)logits, ld, xt, x0, weight, attn_mask, nll, L, V, mask_index);
  else subs_nll_kernel<bf16_t><<<rows, NT, 0, ST(stream)>>>((const bf16_t*)logits, ld, xt, x0, weight, attn_mask, nll, L, V, mask_index);
  MDLM_LAUNCH_CHECK();
  return 0;
}

extern "C" int mdlm_nll_sums(const float* nll, const float* attn_mask,
                             double* sums, int64_t n, void* stream) {
  MDLM_CHECK_ARG(nll && sums && n >= 0, "nll_sums: bad args");
  nll_sums_kernel<<<1, 1024, 0, ST(stream)>>>(nll, attn_mask, sums, n);
  MDLM_LAUNCH_CHECK();
  return 0;
}

extern "C" int mdlm_compact_active(const int32_t* changed, int32_t* active,
                                   int32_t* n_active, int32_t* n_rows, int B,
                                   int L, int all_or_nothing,
                                   int64_t* nfe_accum, int32_t* fresh,
                                   void* stream) {
  MDLM_CHECK_ARG(changed && active && n_active && n_rows && B >= 0 && L > 0, "compact_active: bad args");
  compact_active_kernel<<<1, 64, 0, ST(stream)>>>(changed, active, n_active, n_rows, B, L, all_or_nothing, nfe_accum, fresh);
  MDLM_LAUNCH_CHECK();
  return 0;
}

extern "C" int mdlm_masked_row_map(const int64_t* x, const int32_t* active,
                                   const int32_t* n_active, int L,
                                   int mask_index, int pad, int32_t* gather,
                                   int32_t* scatter, int32_t* n_out,
                                   void* stream) {
  MDLM_CHECK_ARG(x && active && n_active && gather && scatter && n_out, "masked_row_map: null pointer");
  MDLM_CHECK_ARG(L > 0 && pad > 0, "masked_row_map: bad shape");
  masked_row_map_kernel<<<1, 1024, 0, ST(stream)>>>(x, active, n_active, L, mask_index, pad, gather, scatter, n_out);
  MDLM_LAUNCH_CHECK();
  return 0;
}

extern "C" int mdlm_d3pm_loss(const float* neg_logp_x0, const int64_t* xt,
                              const float* t, int T, const float* attn_mask,
                              float* out, int B, int L, int mask_index,
                              void* stream) {
  if (B == 0) return 0;
  MDLM_CHECK_ARG(neg_logp_x0 && xt && t && out, "d3pm_loss: null pointer");
  MDLM_CHECK_ARG(B >= 0 && L > 0 && T > 0, "d3pm_loss: bad shape / T");
  const int64_t n = (int64_t)B * L;
  if (n == 0) return 0;
  d3pm_loss_kernel<<<(unsigned)((n + 255) / 256), 256, 0, ST(stream)>>>(neg_logp_x0, xt, t, T, attn_mask, out, n, L, mask_index);
  MDLM_LAUNCH_CHECK();
  return 0;
}

extern "C" int mdlm_lazy_select(const int64_t* x, const float* t, double dt,
                                uint64_t seed, uint64_t offset, int B, int L,
                                int mask_index, int pad, int64_t* x_next,
                                int32_t* rows, int32_t* n_rows, int32_t* changed,
                                void* stream) {
  MDLM_CHECK_ARG(x && t && x_next && rows && n_rows && changed, "lazy_select: null pointer");
  MDLM_CHECK_ARG(B >= 0 && L > 0 && pad > 0 && (int64_t)B * L < (1ll << 31), "lazy_select: bad shape");
  hipStream_t s = ST(stream);
  hipError_t e = hipMemsetAsync(changed, 0, sizeof(int32_t) * (size_t)B, s);
  if (e != hipSuccess) { set_error("lazy_select: memset failed: %s", hipGetErrorString(e)); return (int)e; }
  lazy_select_kernel<<<1, 1024, 0, s>>>(x, t, (float)dt, seed, offset, B * L, L, mask_index, pad, x_next, rows, n_rows, changed);
  MDLM_LAUNCH_CHECK();
  return 0;
}

extern "C" int mdlm_lazy_race(const void* logits, int64_t ld, int dtype,
                              const int32_t* rows, const int32_t* n_rows,
                              int max_rows, uint64_t seed, uint64_t offset,
                              int64_t* x_next, int V, int mask_index, void* stream) {
  MDLM_CHECK_ARG(logits && rows && n_rows && x_next, "lazy_race: null pointer");
  MDLM_CHECK_ARG(max_rows >= 0 && V > 0 && ld >= V && mask_index >= 0 && mask_index < V, "lazy_race: bad shape");
  MDLM_CHECK_ARG(dtype == MDLM_F32 || dtype == MDLM_BF16, "lazy_race: bad dtype");
  if (max_rows == 0) return 0;
  if (dtype == MDLM_F32) lazy_race_kernel<float><<<(unsigned)max_rows, NT, 0, ST(stream)>>>((const float*)logits, ld, rows, n_rows, seed, offset, x_next, V, mask_index);
  else lazy_race_kernel<bf16_t><<<(unsigned)max_rows, NT, 0, ST(stream)>>>((const bf16_t*)logits, ld, rows, n_rows, seed, offset, x_next, V, mask_index);
  MDLM_LAUNCH_CHECK();
  return 0;
}
