// How many single-wave blocks of a given LDS size and VGPR count does a gfx950 CU hold?  (hipcc -O3 --offload-arch=gfx950)
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int LDS, int VG> __global__ void __launch_bounds__(64, 2) k(int* out) {
  __shared__ char s[LDS];
  if (VG > 128) asm volatile("v_mov_b32 v188, 0" ::: "v188");
  s[threadIdx.x] = 1;
  __syncthreads();
  out[threadIdx.x] = s[(threadIdx.x + 1) % LDS];
}
template <int LDS, int VG> void probe() {
  int n = 0;
  (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k<LDS, VG>, 64, 0);
  printf("LDS %6d B  vgpr %s  -> %d blocks per CU\n", LDS, VG > 128 ? ">128" : "small", n);
}
int main() {
  hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
  printf("%s: sharedMemPerMultiprocessor %zu, maxSharedMemoryPerBlock %zu, regsPerMultiprocessor %d\n", p.gcnArchName, p.maxSharedMemoryPerMultiProcessor, p.sharedMemPerBlock, p.regsPerMultiprocessor);
  probe<16384, 189>(); probe<17408, 189>(); probe<18176, 189>(); probe<18484, 189>(); probe<19456, 189>(); probe<20480, 189>(); probe<18484, 64>(); probe<8192, 189>(); probe<8192, 64>();
  return 0;
}
