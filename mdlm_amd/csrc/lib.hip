// Library-level entry points: version + thread-local error string.
#include "common.h"

namespace mdlm {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
int current_device() {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) { set_error("hipGetDevice: %s", hipGetErrorString(e)); return -1; }
  if (dev < 0 || dev >= MAX_DEVICES) { set_error("device index %d outside 0..%d", dev, MAX_DEVICES - 1); return -1; }
  return dev;
}
int device_cus(int dev) {
  static int cus[MAX_DEVICES] = {};   // racing first calls store the same value
  if (dev < 0 || dev >= MAX_DEVICES) return 256;
  if (!cus[dev]) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 8) v = 256;
    cus[dev] = v & ~7;   // the tile walks deal workgroups to 8 XCDs
  }
  return cus[dev];
}
}  // namespace mdlm

extern "C" int mdlm_version(void) { return MDLM_VERSION; }
extern "C" const char* mdlm_last_error_string(void) { return mdlm::g_err; }

// ---- workspace sizes (SURVEY 8(b) "Ownership": the caller allocates every
// buffer, sized by these queries; host-only arithmetic, no GPU needed).
// Each returns bytes, or -1 for a bad shape.
static inline int64_t up(int64_t a, int64_t b) { return (a + b - 1) / b; }
extern "C" int64_t mdlm_attn_bwd_workspace_bytes(int n_seq, int H, int L) {
  if (n_seq < 0 || H <= 0 || L <= 0) return -1;
  return (int64_t)n_seq * H * L * 4;                       // delta_ws
}
extern "C" int64_t mdlm_grad_sumsq_workspace_bytes(void) { return 1024 * 4; }   // partial_1024
extern "C" int64_t mdlm_gemm_bf16_splitk_workspace_bytes(int64_t M, int N, int splits) {
  if (M < 0 || N <= 0 || splits < 1) return -1;
  return (int64_t)splits * M * N * 4;                      // partials
}
extern "C" int64_t mdlm_gemm_tn_splitk_workspace_bytes(int N, int K, int splits,
                                                       int64_t* bias_partials_bytes) {
  if (N <= 0 || K <= 0 || splits < 1) return -1;
  if (bias_partials_bytes) *bias_partials_bytes = (int64_t)splits * N * 4;
  return (int64_t)splits * N * K * 4;                      // partials
}
extern "C" int64_t mdlm_embed_grad_workspace_bytes(int64_t rows, int D) {
  if (rows < 0 || D <= 0) return -1;
  return up(rows, 128) * D * 4;                            // hot_partial
}
extern "C" int64_t mdlm_ln_modulate_bwd_workspace_bytes(int64_t rows, int D) {
  if (rows < 0 || D <= 0) return -1;
  return up(rows, 32) * 3 * D * 4;                         // partial
}
extern "C" int64_t mdlm_dropout_gate_residual_bwd_workspace_bytes(int64_t rows, int D) {
  if (rows < 0 || D <= 0) return -1;
  return up(rows, 32) * D * 4;                             // partial
}
extern "C" int64_t mdlm_colsum_bf16_partial_workspace_bytes(int64_t rows, int N) {
  if (rows < 0 || N <= 0) return -1;
  return up(rows, 128) * N * 4;                            // partial
}
extern "C" int64_t mdlm_transpose_bf16_workspace_bytes(int R, int C) {
  if (R < 0 || C <= 0) return -1;
  return up(R, 64) * C * 4;                                // colsum_partial
}
extern "C" int64_t mdlm_masked_row_map_workspace_bytes(int64_t rows, int pad) {
  if (rows < 0 || pad <= 0) return -1;
  return (rows + pad) * 4;                                 // gather and scatter, each
}
