// v_subrev_u32_dpp on gfx950: the lane permutation lands on the wrong operand.
// hipcc's DPP combiner (GCNDPPCombine) folds a v_mov_b32_dpp into its user; when the permuted value is the subtrahend it
// emits v_subrev_u32_dpp (D = src1 - dpp(src0)).  On the MI355X of this pool that instruction computes dpp(src1) - src0:
// half of the lanes of a quad_perm:[1,1,3,3] come out wrong, while v_add_u32_dpp and v_sub_u32_dpp are right, with or
// without dst == src0.  Found when a rewrite of the four-lane X25519 ladder (quad_lanes.h) made the compiler emit it;
// the product is built with -mllvm -amdgpu-dpp-combine=false (Makefile) since.
//   hipcc --offload-arch=gfx950 -O1 -o dpp_subrev tools/microbench/dpp_subrev.hip && ./dpp_subrev
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(uint32_t* o) {
  uint32_t a = 1000 * (threadIdx.x + 1), b = 7 * (threadIdx.x + 1), r0, r1, r2, r3;
  // (0) subrev, dst != src: r0 = b - perm(a)
  asm volatile("v_subrev_u32_dpp %0, %1, %2 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=&v"(r0) : "v"(a), "v"(b));
  // (1) subrev, dst == dpp src
  r1 = a;
  asm volatile("v_subrev_u32_dpp %0, %0, %1 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r1) : "v"(b));
  // (2) add, dst == dpp src
  r2 = a;
  asm volatile("v_add_u32_dpp %0, %0, %1 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r2) : "v"(b));
  // (3) sub, dst == dpp src: r3 = perm(a) - b
  r3 = a;
  asm volatile("v_sub_u32_dpp %0, %0, %1 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r3) : "v"(b));
  o[threadIdx.x] = r0; o[64 + threadIdx.x] = r1; o[128 + threadIdx.x] = r2; o[192 + threadIdx.x] = r3;
}
int main() {
  uint32_t out[256], *o; hipMalloc(&o, sizeof(out));
  k<<<1, 64>>>(o); hipMemcpy(out, o, sizeof(out), hipMemcpyDeviceToHost);
  int bad[4] = {0, 0, 0, 0};
  for (int l = 0; l < 64; l++) {
    const uint32_t pa = 1000 * ((l | 1) + 1), b = 7 * (l + 1);
    if (out[l] != b - pa) bad[0]++;
    if (out[64 + l] != b - pa) bad[1]++;
    if (out[128 + l] != pa + b) bad[2]++;
    if (out[192 + l] != pa - b) bad[3]++;
  }
  printf("wrong lanes: subrev dst!=src %d, subrev dst==src %d, add dst==src %d, sub dst==src %d\n", bad[0], bad[1], bad[2], bad[3]);
  return 0;
}
