// Does a burst of global stores block the issuing wave (issue back-pressure), or
// only a later vmcnt wait?  256 workgroups x 512 threads; each wave issues NS
// 1-KiB stores, then runs NM dependent MFMAs (register only).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>  // 0: mfma only, 1: stores then mfma, 2: stores, vmcnt(0), mfma, 3: stores only
__global__ __launch_bounds__(512, 2) void k(u32x4* out, float* sink, int ns, int nm, int rounds) {
  const int wave = (blockIdx.x * 8 + (threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(lane + i); b[i] = (__bf16)(float)(lane - i); }
  u32x4 v = {(unsigned)lane, 1u, 2u, 3u};
  for (int r = 0; r < rounds; ++r) {
    if (MODE == 1 || MODE == 2 || MODE == 3) {
      u32x4* p = out + ((size_t)(r * gridDim.x * 8 + wave) * ns) * 64 + lane;
      for (int s = 0; s < ns; ++s) __builtin_nontemporal_store(v, p + s * 64);
    }
    if (MODE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE != 3)
      for (int m = 0; m < nm; ++m)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0];
  if (s == 123.456f) sink[0] = s;
}

int main() {
  const int ns = 32, nm = 96, rounds = 8;   // 96*8 MFMAs ~ 12k cycles ~ 6 us per round
  const size_t bytes = (size_t)rounds * 256 * 8 * ns * 1024;
  u32x4* out; float* sink;
  hipMalloc(&out, bytes); hipMalloc(&sink, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](int mode) {
    float best = 1e9;
    for (int it = 0; it < 5; ++it) {
      hipEventRecord(e0);
      if (mode == 0) k<0><<<256, 512>>>(out, sink, ns, nm, rounds);
      if (mode == 1) k<1><<<256, 512>>>(out, sink, ns, nm, rounds);
      if (mode == 2) k<2><<<256, 512>>>(out, sink, ns, nm, rounds);
      if (mode == 3) k<3><<<256, 512>>>(out, sink, ns, nm, rounds);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best * 1e3f;
  };
  printf("bytes per round per CU: %d KiB, total %.1f MB\n", 8 * ns, bytes / 1e6);
  printf("mfma only          : %8.1f us\n", run(0));
  printf("stores only        : %8.1f us  (%.2f TB/s)\n", run(3), bytes / run(3) / 1e6);
  printf("stores then mfma   : %8.1f us\n", run(1));
  printf("stores, wait, mfma : %8.1f us\n", run(2));
  return 0;
}
