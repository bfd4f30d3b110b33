// Error plumbing + version exports of libflowers_hip.so.
#include "fl_common.h"
#include <cstdarg>
#include <cstdio>

namespace fl {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: HIP launch error: %s", what, hipGetErrorString(e));
    return FL_ERR_LAUNCH;
  }
  return FL_OK;
}

}  // namespace fl

extern "C" {
int fl_version(void) { return 100; }
const char* fl_last_error_string(void) { return fl::g_err; }
}
