// rlc.hip - batch verification by random linear combination (placeholder; filled in below)
#include "eddsa_kernels.h"
extern "C" size_t edk_rlc_ws_bytes(size_t capacity) { return capacity ? 256 : 0; }
extern "C" hipError_t edk_verify_rlc(uint8_t*, uint32_t*, const edk_verify_src*, size_t, const uint32_t*,
                                     const edk_verify_ws*, const edk_rlc_ws*, hipStream_t) { return hipErrorNotSupported; }
