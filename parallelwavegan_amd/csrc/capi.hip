// capi.hip -- error reporting and version queries of the C ABI.
#include "common.h"

namespace pwg {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace pwg

extern "C" const char* pwg_last_error(void) { return pwg::g_err; }
extern "C" int pwg_abi_version(void) { return 1; }
extern "C" int pwg_target_arch(void) { return 950; }
