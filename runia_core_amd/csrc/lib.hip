// Library-level entry points of librunia_hip.so.
#include "common.hpp"

extern "C" int runia_abi_version(void) { return 4; }

extern "C" const char* runia_error_string(int code) {
  switch (code) {
    case RUNIA_OK: return "ok";
    case RUNIA_E_INVALID: return "invalid argument (shape, null pointer or unsupported size)";
    case RUNIA_E_LAUNCH: return "HIP kernel launch failed";
    case RUNIA_E_NODEVICE: return "no HIP device visible";
    case RUNIA_E_WORKSPACE: return "workspace too small";
    default: return "unknown error";
  }
}

extern "C" int runia_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
