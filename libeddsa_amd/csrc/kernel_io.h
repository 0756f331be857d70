// kernel_io.h - how the kernels move packed byte arrays and shared tables: used by kernels.hip (the product) and by
// probe.hip (the layer probes, a separate library).
#pragma once
#include "lanes.h"

namespace ed {

// ---- packed byte-array access: 32 bytes per item as eight little-endian words ---------------

ED_DEV void load32(uint32_t w[8], const uint8_t* base, size_t item, size_t stride) {
  const uint8_t* p = base + item * stride;
  if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
    const uint4 a = reinterpret_cast<const uint4*>(p)[0];
    const uint4 b = reinterpret_cast<const uint4*>(p)[1];
    w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
    w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
  } else {
#pragma unroll
    for (int i = 0; i < 8; i++)
      w[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) |
             ((uint32_t)p[4 * i + 3] << 24);
  }
}

ED_DEV void store32(uint8_t* base, size_t item, size_t stride, const uint32_t w[8]) {
  uint8_t* p = base + item * stride;
  if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
    reinterpret_cast<uint4*>(p)[0] = make_uint4(w[0], w[1], w[2], w[3]);
    reinterpret_cast<uint4*>(p)[1] = make_uint4(w[4], w[5], w[6], w[7]);
  } else {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      p[4 * i] = (uint8_t)w[i]; p[4 * i + 1] = (uint8_t)(w[i] >> 8);
      p[4 * i + 2] = (uint8_t)(w[i] >> 16); p[4 * i + 3] = (uint8_t)(w[i] >> 24);
    }
  }
}

// copy a table of `words` 32-bit words (a multiple of 4, 16-byte aligned) from HBM into LDS (whole block)
ED_DEV void stage_table(uint32_t* lds, const uint32_t* src, int words) {
  word4* d = reinterpret_cast<word4*>(lds);
  const word4* s = reinterpret_cast<const word4*>(src);
  for (int j = threadIdx.x; j < words / 4; j += (int)blockDim.x) d[j] = s[j];
  __syncthreads();
}

}  // namespace ed
