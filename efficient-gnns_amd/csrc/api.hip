// Library-level entry points of libegnn_hip.so (version, error strings, build info).
#include <stdio.h>
#include <string.h>

#include "common.h"

extern "C" int egnn_abi_version(void) { return EGNN_ABI_VERSION; }

extern "C" const char* egnn_error_string(int code) {
  switch (code) {
    case EGNN_OK: return "ok";
    case EGNN_EINVAL: return "invalid argument";
    case EGNN_ELAUNCH: return "HIP launch error";
    case EGNN_EWORKSPACE: return "workspace too small";
    case EGNN_EALIGN: return "misaligned pointer or leading dimension";
    default: return "unknown error";
  }
}

extern "C" int egnn_build_info(char* buf, size_t buf_bytes) {
  if (!buf || buf_bytes == 0) return EGNN_EINVAL;
  snprintf(buf, buf_bytes, "libegnn_hip abi=%d arch=gfx950 wave=64 built=%s %s", EGNN_ABI_VERSION, __DATE__, __TIME__);
  return EGNN_OK;
}
