// Throughput of the device field layer (fe_mul / fe_sq) in cycles per wave64 operation.
// build: hipcc -O3 --offload-arch=gfx950 -I../../libeddsa_amd/csrc fe_rates.hip -o fe_rates.bin
#include "fe25519.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace ed;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int N = 4096;

__device__ void seed(fe& a, uint32_t s) {
  for (int i = 0; i < 10; i++) a.v[i] = (s * 2654435761u + i * 40503u) & limb_mask(i);
}
__device__ uint32_t fold(const fe& a) { uint32_t r = 0; for (int i = 0; i < 10; i++) r ^= a.v[i]; return r; }

template <int MODE>
__global__ void __launch_bounds__(256) k(uint32_t* out, uint64_t* cyc) {
  fe a, b, c, d;
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  seed(a, tid); seed(b, tid + 7777); seed(c, tid + 31337); seed(d, tid + 99);
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  if (MODE == 0) { for (int i = 0; i < N; i++) fe_mul(a, a, b); }
  if (MODE == 1) { for (int i = 0; i < N; i++) fe_sq(a, a); }
  if (MODE == 2) { for (int i = 0; i < N / 2; i++) { fe_mul(a, a, b); fe_mul(c, c, d); } }
  if (MODE == 3) { for (int i = 0; i < N / 2; i++) { fe_sq(a, a); fe_sq(c, c); } }
  if (MODE == 4) { for (int i = 0; i < N / 4; i++) { fe_mul(a, a, b); fe_mul(c, c, d); fe_mul(b, b, a); fe_mul(d, d, c); } }
  if (MODE == 5) { for (int i = 0; i < N; i++) { fe t; fe_sub(t, a, b); fe_add(a, t, b); fe_carry(a); } }
  // partial waves: does a wave64 instruction with only 16 (32) active lanes still take 4 passes?
  if (MODE == 6) { if ((threadIdx.x & 63) < 16) for (int i = 0; i < N; i++) fe_mul(a, a, b); }
  if (MODE == 7) { if ((threadIdx.x & 63) < 32) for (int i = 0; i < N; i++) fe_mul(a, a, b); }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  out[tid] = fold(a) ^ fold(b) ^ fold(c) ^ fold(d);
  if ((threadIdx.x & 63) == 0) cyc[tid >> 6] = t1 - t0;
}

typedef void (*kern_t)(uint32_t*, uint64_t*);
int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  struct { const char* name; kern_t f; } es[] = {
    {"fe_mul dependent chain", k<0>}, {"fe_sq dependent chain", k<1>},
    {"fe_mul 2 chains", k<2>}, {"fe_sq 2 chains", k<3>}, {"fe_mul 4-cross", k<4>},
    {"sub+add+carry", k<5>}, {"fe_mul chain, 16 lanes", k<6>}, {"fe_mul chain, 32 lanes", k<7>} };
  uint32_t* out; uint64_t* cyc;
  CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4)); CK(hipMalloc(&cyc, (size_t)cus * 8 * 4 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("%-26s %4s %14s %14s %12s\n", "op", "w/S", "cyc/op (wave)", "cyc/op/SIMD", "Gop/s chip");
  for (auto& e : es) for (int wps : {1, 2, 4, 5, 6, 8}) {
    int blocks = cus * wps;
    e.f<<<blocks, 256>>>(out, cyc); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); e.f<<<blocks, 256>>>(out, cyc); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<uint64_t> h(blocks * 4); CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
    double avg = 0; for (auto v : h) avg += (double)v; avg /= h.size();
    printf("%-26s %4d %14.1f %14.1f %12.2f  (%.3f ms)\n", e.name, wps, avg / N, avg / N / wps,
           (double)N * 256.0 * blocks / (ms * 1e-3) / 1e9, ms);
  }
  return 0;
}
