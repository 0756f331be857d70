// Single-call latency: how long are the candidate critical paths of ONE ed25519_verify after the points are
// decompressed, on an idle chip, in the four-lanes-per-point form of quad_lanes.h?  One wave, one quad active.
//   chain   what k_verify_main_sums_quad runs today: 34 windows x (4 doublings + 1 addition of a ready window sum)
//   sliced  the proposal: the doubling chain 2^34 P, 2^68 P, 2^102 P (102 doublings, needs no scalar), then for the last
//           slice a table of 8 multiples (7 operations), its 8 windows x (4 doublings + 1 addition) + 1, and 3 additions
//           to join the four slices (the other slices and the second point run beside it in other quads / waves)
//   dbl     102 doublings alone; add: 34 additions alone (what the slicing takes off the path)
// build: hipcc -O3 --offload-arch=gfx950 -I../../libeddsa_amd/csrc -I../../include -mllvm -amdgpu-dpp-combine=false slice_path.hip -o slice_path.bin
#include "quad_lanes.h"
#include <cstdio>
#include <cstdlib>
using namespace ed;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

template <int MODE>
__global__ void __launch_bounds__(64) k(uint32_t* out, const uint32_t* entries) {
  if (threadIdx.x >= 4) return;
  const int q = (int)(threadIdx.x & 3u);
  fe r;
  fe_set(r, (uint32_t)(q & 1));
  quad_add_entry(r, entries, false, true, q);                   // some point
  if (MODE == 0) {
#pragma unroll 1
    for (int w = 0; w < 34; w++) {
#pragma unroll 1
      for (int d = 0; d < 4; d++) quad_dbl(r, q);
      quad_add_entry(r, entries + 32 * (w & 7), false, true, q);
    }
  } else if (MODE == 1) {
#pragma unroll 1
    for (int d = 0; d < 102; d++) quad_dbl(r, q);
#pragma unroll 1
    for (int t = 0; t < 7; t++) quad_add_entry(r, entries + 32 * t, false, true, q);      // the slice's table of multiples
#pragma unroll 1
    for (int w = 0; w < 8; w++) {
#pragma unroll 1
      for (int d = 0; d < 4; d++) quad_dbl(r, q);
      quad_add_entry(r, entries + 32 * (w & 7), false, true, q);
    }
#pragma unroll 1
    for (int t = 0; t < 4; t++) quad_add_entry(r, entries + 32 * t, false, true, q);      // its last window, and the join
  } else if (MODE == 2) {
#pragma unroll 1
    for (int d = 0; d < 102; d++) quad_dbl(r, q);
  } else {
#pragma unroll 1
    for (int w = 0; w < 34; w++) quad_add_entry(r, entries + 32 * (w & 7), false, true, q);
  }
  uint32_t x = 0;
  for (int i = 0; i < 10; i++) x ^= r.v[i];
  out[threadIdx.x] = x;
}

int main() {
  uint32_t *out, *entries;
  CK(hipMalloc(&out, 256)); CK(hipMalloc(&entries, 8 * 128));
  uint32_t h[8 * 32];
  for (int i = 0; i < 8 * 32; i++) h[i] = (i * 2654435761u) >> ((i & 7) == 7 ? 2 : 0);     // (any 255-bit words will do: timing only)
  CK(hipMemcpy(entries, h, sizeof(h), hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct { const char* name; void (*f)(uint32_t*, const uint32_t*); int ops; } es[] = {
    {"chain: 34 x (4 dbl + add)", k<0>, 170}, {"sliced: 102 dbl + 7 + 8 x (4 dbl + add) + 4", k<1>, 153},
    {"102 doublings", k<2>, 102}, {"34 additions", k<3>, 34}};
  for (auto& e : es) {
    for (int i = 0; i < 3; i++) e.f<<<1, 64>>>(out, entries);
    CK(hipDeviceSynchronize());
    float best = 1e9;
    for (int rep = 0; rep < 10; rep++) {
      CK(hipEventRecord(e0)); e.f<<<1, 64>>>(out, entries); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%-46s %7.1f us   (%d point operations: %.2f us each, launch included)\n", e.name, best * 1e3, e.ops, best * 1e3 / e.ops);
  }
  return 0;
}
