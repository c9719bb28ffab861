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
}  // namespace mdlm

extern "C" int mdlm_version(void) { return MDLM_VERSION; }
extern "C" const char* mdlm_last_error_string(void) { return mdlm::g_err; }
