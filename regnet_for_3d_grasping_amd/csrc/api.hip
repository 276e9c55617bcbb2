// api.hip -- library identification + error strings for libregnet_hip.so.
#include "common.h"

extern "C" int regnet_abi_version(void) { return 1; }

extern "C" const char* regnet_build_info(void) {
  return "libregnet_hip gfx950 (CDNA4, wave64) built " __DATE__ " " __TIME__;
}

extern "C" const char* regnet_strerror(int code) {
  switch (code) {
    case REGNET_OK: return "ok";
    case REGNET_ERR_SHAPE: return "shape/argument check failed";
    case REGNET_ERR_NULL: return "null pointer for a non-empty tensor";
    case REGNET_ERR_UNSUPPORTED: return "size not supported by this build";
    default: break;
  }
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "unknown error";
}
