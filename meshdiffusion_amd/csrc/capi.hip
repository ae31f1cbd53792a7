// ABI version / device probe for libmeshdiffusion_hip.so.
#include "md_common.h"

extern "C" int md_abi_version(void) { return MD_ABI_VERSION; }

extern "C" int md_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return MD_ERR_NO_DEVICE;
  return n;
}
