// Thread-local error string + ABI version for libunet_hip.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/unet_hip.h"

static thread_local char g_err[512] = "";

void unet_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* unet_last_error(void) { return g_err; }
extern "C" int unet_abi_version(void) { return UNET_ABI_VERSION; }
