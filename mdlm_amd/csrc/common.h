// Shared helpers for libmdlm_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/mdlm_hip.h"

#define MDLM_VERSION 100

namespace mdlm {

void set_error(const char* fmt, ...);

// Per-device launch state.  hipFuncSetAttribute (dynamic-LDS opt-in) acts on the
// CURRENT device's copy of a kernel and a persistent grid is sized by THAT
// device's CU count, so both are kept per device, not per process: a process
// that drives several GPUs gets each one set up on its first launch there.
constexpr int MAX_DEVICES = 64;
int current_device();          // hipGetDevice; -1 (with the error string set) on failure
int device_cus(int dev);       // CU count of device `dev` (multiple of 8 XCDs), cached

// Opt `kernel` in to `bytes` of dynamic LDS on the current device, once per device.
// `done` is the caller's per-device flag array (one static array per kernel).
template <typename K>
inline int ensure_dynamic_lds(K kernel, int bytes, bool (&done)[MAX_DEVICES], const char* who, int* dev_out = nullptr) {
  const int dev = current_device();
  if (dev < 0) return -1;
  if (dev_out) *dev_out = dev;
  if (!done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) { set_error("%s: hipFuncSetAttribute: %s", who, hipGetErrorString(e)); return (int)e; }
    done[dev] = true;
  }
  return 0;
}

#define MDLM_CHECK_ARG(cond, ...)            \
  do {                                       \
    if (!(cond)) {                           \
      mdlm::set_error(__VA_ARGS__);          \
      return -1;                             \
    }                                        \
  } while (0)

#define MDLM_LAUNCH_CHECK()                                        \
  do {                                                             \
    hipError_t e__ = hipGetLastError();                            \
    if (e__ != hipSuccess) {                                       \
      mdlm::set_error("launch failed: %s", hipGetErrorString(e__)); \
      return (int)e__;                                             \
    }                                                              \
  } while (0)

typedef unsigned short bf16_t;  // raw bf16 bits

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
  return __uint_as_float(((unsigned)v) << 16);
}
// round-to-nearest-even; a plain cast lowers to v_cvt_pk_bf16_f32 on gfx950
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  __bf16 b = (__bf16)f;
  return *reinterpret_cast<bf16_t*>(&b);
}
// one v_cvt_pk_bf16_f32 for the pair (two scalar casts make hipcc pair values
// of different words and re-shuffle them with and/shift/or)
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v = {lo, hi};
  const bf16x2_t b = __builtin_convertvector(v, bf16x2_t);
  return *reinterpret_cast<const unsigned*>(&b);
}

template <typename T> struct elem;
template <> struct elem<float> {
  static __device__ __forceinline__ float load(const float* p) { return *p; }
  static __device__ __forceinline__ void store(float* p, float v) { *p = v; }
};
template <> struct elem<bf16_t> {
  static __device__ __forceinline__ float load(const bf16_t* p) { return bf16_to_f32(*p); }
  static __device__ __forceinline__ void store(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// ---- Philox4x32-7 (counter-based; Salmon et al., SC'11) --------------------
// Seven rounds is the smallest round count the paper reports as Crush-resistant
// (it passes BigCrush); ten is its default "with safety margin".  The generator
// is this library's own (the reference calls torch.rand_like, diffusion.py:27;
// parity is defined on supplied / materialised uniforms, mdlm_philox_uniform),
// and the race kernel is bound by exactly these integer multiplies, so the
// minimum is used: 30 % fewer of them.
constexpr int PHILOX_ROUNDS = 7;
struct Philox {
  static __device__ __forceinline__ uint4 gen(uint64_t seed, uint64_t ctr) {
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
    unsigned c0 = (unsigned)ctr, c1 = (unsigned)(ctr >> 32), c2 = 0x6d646c6du, c3 = 0;
#pragma unroll
    for (int r = 0; r < PHILOX_ROUNDS; ++r) {
      // one 32x32->64 multiply per lane pair (v_mad_u64_u32) instead of a
      // separate v_mul_hi_u32 + v_mul_lo_u32
      const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
      const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
      unsigned hi0 = (unsigned)(p0 >> 32), lo0 = (unsigned)p0;
      unsigned hi1 = (unsigned)(p1 >> 32), lo1 = (unsigned)p1;
      // three-input xor in one v_bitop3_b32 (truth table 0x96)
      unsigned n0 = __builtin_amdgcn_bitop3_b32(hi1, c1, k0, 0x96);
      unsigned n2 = __builtin_amdgcn_bitop3_b32(hi0, c3, k1, 0x96);
      c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
      k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
  }
  // 24-bit uniform in [0,1), like torch's CPU float uniform
  static __device__ __forceinline__ float u01(unsigned x) {
    return (float)(x >> 8) * (1.0f / 16777216.0f);
  }
};

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

}  // namespace mdlm
