// Error reporting and version for libembnet_hip.so.
#include "common.h"
#include "../../include/embnet.h"

namespace embnet {

char* last_error_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(last_error_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

}  // namespace embnet

extern "C" int embnet_abi_version(void) { return EMBNET_ABI_VERSION; }
extern "C" const char* embnet_last_error(void) { return embnet::last_error_buf(); }
