// VALU issue-rate microbenchmark for gfx950 (MI355X).
// Decides the limb formulation of the GF(2^255-19) multiplier: measures the sustained
// per-SIMD issue cost (cycles per wave64 instruction) of every candidate multiply /
// carry instruction, at 1, 2 and 4 waves per SIMD.
//
// build: hipcc -O3 --offload-arch=gfx950 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int ITERS = 2048;
constexpr int PER_ITER = 32;   // instructions per loop iteration (8 chains x 4)

// 32-bit destination, 8 independent chains a0..a7, sources x,y
#define REP8_32(INS) \
  INS("%0") INS("%1") INS("%2") INS("%3") INS("%4") INS("%5") INS("%6") INS("%7")
#define BODY32(INS) REP8_32(INS) REP8_32(INS) REP8_32(INS) REP8_32(INS)

#define KERNEL32(NAME, INS)                                                         \
__global__ void __launch_bounds__(256) NAME(uint32_t* out, uint64_t* cyc, uint32_t x, uint32_t y) { \
  uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3,                \
           a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;                      \
  uint32_t vx = x + threadIdx.x, vy = y ^ threadIdx.x;                              \
  uint64_t t0 = __builtin_amdgcn_s_memtime();                                       \
  for (int i = 0; i < ITERS; ++i) {                                                 \
    asm volatile(BODY32(INS)                                                        \
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
      : "v"(vx), "v"(vy) : "vcc", "s12", "s13");                                    \
  }                                                                                 \
  uint64_t t1 = __builtin_amdgcn_s_memtime();                                       \
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7; \
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0; \
}

#define KERNEL64(NAME, INS)                                                         \
__global__ void __launch_bounds__(256) NAME(uint32_t* out, uint64_t* cyc, uint32_t x, uint32_t y) { \
  uint64_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3,                \
           a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;                      \
  uint32_t vx = x + threadIdx.x, vy = y ^ threadIdx.x;                              \
  uint64_t wx = ((uint64_t)x << 32) | threadIdx.x, wy = ((uint64_t)y << 20) + 77;   \
  uint64_t t0 = __builtin_amdgcn_s_memtime();                                       \
  for (int i = 0; i < ITERS; ++i) {                                                 \
    asm volatile(BODY32(INS)                                                        \
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
      : "v"(vx), "v"(vy), "v"(wx), "v"(wy) : "vcc", "s10", "s11");                                \
  }                                                                                 \
  uint64_t t1 = __builtin_amdgcn_s_memtime();                                       \
  uint64_t r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                               \
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)r ^ (uint32_t)(r >> 32);   \
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0; \
}

// ---- 32-bit ops (operands: %8 = vx, %9 = vy)
#define I_ADD(d)      "v_add_u32 " d ", %8, " d "\n"
#define I_XOR(d)      "v_xor_b32 " d ", %8, " d "\n"
#define I_ADD3(d)     "v_add3_u32 " d ", %8, %9, " d "\n"
#define I_LSHLADD(d)  "v_lshl_add_u32 " d ", " d ", 3, %8\n"
#define I_ANDOR(d)    "v_and_or_b32 " d ", " d ", %8, %9\n"
#define I_ALIGNBIT(d) "v_alignbit_b32 " d ", " d ", %8, 13\n"
#define I_BFE(d)      "v_bfe_u32 " d ", " d ", 5, 26\n"
#define I_PERM(d)     "v_perm_b32 " d ", " d ", %8, %9\n"
#define I_MULLO(d)    "v_mul_lo_u32 " d ", " d ", %8\n"
#define I_MULHI(d)    "v_mul_hi_u32 " d ", " d ", %8\n"
#define I_MULU24(d)   "v_mul_u32_u24 " d ", " d ", %8\n"
#define I_MULHIU24(d) "v_mul_hi_u32_u24 " d ", " d ", %8\n"
#define I_MADU24(d)   "v_mad_u32_u24 " d ", %8, %9, " d "\n"
#define I_MADI24(d)   "v_mad_i32_i24 " d ", %8, %9, " d "\n"
#define I_MADU16(d)   "v_mad_u32_u16 " d ", %8, %9, " d "\n"
#define I_DOT4U8(d)   "v_dot4_u32_u8 " d ", %8, %9, " d "\n"
#define I_DOT2U16(d)  "v_dot2_u32_u16 " d ", %8, %9, " d "\n"
#define I_DOT8U4(d)   "v_dot8_u32_u4 " d ", %8, %9, " d "\n"
#define I_PKMADU16(d) "v_pk_mad_u16 " d ", %8, %9, " d "\n"
#define I_PKADDU16(d) "v_pk_add_u16 " d ", %8, " d "\n"
#define I_ADDCO(d)    "v_add_co_u32 " d ", vcc, %8, " d "\n"
#define I_ADDC(d)     "v_addc_co_u32 " d ", vcc, %8, " d ", vcc\n"
#define I_FMAF32(d)   "v_fma_f32 " d ", %8, %9, " d "\n"
#define I_CNDMASK(d)  "v_cndmask_b32 " d ", " d ", %8, vcc\n"
#define I_CNDMASK64(d) "v_cndmask_b32_e64 " d ", " d ", %8, s[12:13]\n"
#define I_CNDMASKI(d) "v_cndmask_b32 " d ", %9, %8, vcc\n"
#define I_BFI(d)      "v_bfi_b32 " d ", %8, %9, " d "\n"
#define I_AND(d)      "v_and_b32 " d ", %8, " d "\n"
#define I_OR(d)       "v_or_b32 " d ", %8, " d "\n"
#define I_LSHL(d)     "v_lshlrev_b32 " d ", 3, " d "\n"
#define I_LSHR(d)     "v_lshrrev_b32 " d ", 3, " d "\n"
#define I_SUB(d)      "v_sub_u32 " d ", %8, " d "\n"
#define I_MOV(d)      "v_mov_b32 " d ", %8\n"
#define I_XAD(d)      "v_xad_u32 " d ", " d ", %8, %9\n"
#define I_ADDLSHL(d)  "v_add_lshl_u32 " d ", " d ", %8, 1\n"
#define I_LSHLOR(d)   "v_lshl_or_b32 " d ", " d ", 3, %8\n"

KERNEL32(k_add, I_ADD)
KERNEL32(k_xor, I_XOR)
KERNEL32(k_add3, I_ADD3)
KERNEL32(k_lshladd, I_LSHLADD)
KERNEL32(k_andor, I_ANDOR)
KERNEL32(k_alignbit, I_ALIGNBIT)
KERNEL32(k_bfe, I_BFE)
KERNEL32(k_perm, I_PERM)
KERNEL32(k_mullo, I_MULLO)
KERNEL32(k_mulhi, I_MULHI)
KERNEL32(k_mulu24, I_MULU24)
KERNEL32(k_mulhiu24, I_MULHIU24)
KERNEL32(k_madu24, I_MADU24)
KERNEL32(k_madi24, I_MADI24)
KERNEL32(k_madu16, I_MADU16)
KERNEL32(k_dot4u8, I_DOT4U8)
KERNEL32(k_dot2u16, I_DOT2U16)
KERNEL32(k_dot8u4, I_DOT8U4)
KERNEL32(k_pkmadu16, I_PKMADU16)
KERNEL32(k_pkaddu16, I_PKADDU16)
KERNEL32(k_addco, I_ADDCO)
KERNEL32(k_addc, I_ADDC)
KERNEL32(k_fmaf32, I_FMAF32)
KERNEL32(k_cndmask, I_CNDMASK)
KERNEL32(k_cndmask64, I_CNDMASK64)
KERNEL32(k_cndmaski, I_CNDMASKI)
KERNEL32(k_bfi, I_BFI)
KERNEL32(k_and, I_AND)
KERNEL32(k_or, I_OR)
KERNEL32(k_lshl, I_LSHL)
KERNEL32(k_lshr, I_LSHR)
KERNEL32(k_sub, I_SUB)
KERNEL32(k_mov, I_MOV)
KERNEL32(k_xad, I_XAD)
KERNEL32(k_addlshl, I_ADDLSHL)
KERNEL32(k_lshlor, I_LSHLOR)

// ---- 64-bit destination ops (operands: %8 = vx, %9 = vy (32-bit), %10 = wx, %11 = wy (64-bit))
#define I_MAD64(d)     "v_mad_u64_u32 " d ", vcc, %8, %9, " d "\n"
#define I_MAD64S(d)    "v_mad_u64_u32 " d ", s[10:11], %8, %9, " d "\n"
#define I_MADI64(d)    "v_mad_i64_i32 " d ", vcc, %8, %9, " d "\n"
#define I_FMAF64(d)    "v_fma_f64 " d ", %10, %11, " d "\n"
#define I_ADDF64(d)    "v_add_f64 " d ", %10, " d "\n"
#define I_MULF64(d)    "v_mul_f64 " d ", %10, " d "\n"
#define I_LSHLADD64(d) "v_lshl_add_u64 " d ", " d ", 1, %10\n"
#define I_LSHR64(d)    "v_lshrrev_b64 " d ", 5, " d "\n"
#define I_ASHR64(d)    "v_ashrrev_i64 " d ", 5, " d "\n"
#define I_PKFMAF32(d)  "v_pk_fma_f32 " d ", %10, %11, " d "\n"
#define I_PKADDF32(d)  "v_pk_add_f32 " d ", %10, " d "\n"

KERNEL64(k_mad64, I_MAD64)
KERNEL64(k_mad64s, I_MAD64S)
KERNEL64(k_madi64, I_MADI64)
KERNEL64(k_fmaf64, I_FMAF64)
KERNEL64(k_addf64, I_ADDF64)
KERNEL64(k_mulf64, I_MULF64)
KERNEL64(k_lshladd64, I_LSHLADD64)
KERNEL64(k_lshr64, I_LSHR64)
KERNEL64(k_ashr64, I_ASHR64)
KERNEL64(k_pkfmaf32, I_PKFMAF32)
KERNEL64(k_pkaddf32, I_PKADDF32)

// mixed stream: 1 mad_u64_u32 followed by 1 cheap op, to see whether they dual-issue/overlap
#define I_MIX_MAD_ADD(d) "v_mad_u64_u32 " d ", vcc, %8, %9, " d "\n"
__global__ void __launch_bounds__(256) k_mix(uint32_t* out, uint64_t* cyc, uint32_t x, uint32_t y) {
  uint64_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  uint32_t b0 = 1, b1 = 2, b2 = 3, b3 = 4;
  uint32_t vx = x + threadIdx.x, vy = y ^ threadIdx.x;
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
    asm volatile(
#define MIXROW \
      "v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_add_u32 %4, %8, %4\n" \
      "v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_add_u32 %5, %8, %5\n" \
      "v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_add_u32 %6, %8, %6\n" \
      "v_mad_u64_u32 %3, vcc, %8, %9, %3\n v_add_u32 %7, %8, %7\n"
      MIXROW MIXROW MIXROW MIXROW
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3)
      : "v"(vx), "v"(vy) : "vcc");
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  uint64_t r = a0 ^ a1 ^ a2 ^ a3;
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)r ^ (uint32_t)(r >> 32) ^ b0 ^ b1 ^ b2 ^ b3;
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

typedef void (*kern_t)(uint32_t*, uint64_t*, uint32_t, uint32_t);
struct Entry { const char* name; kern_t k; int per_iter; };

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  printf("# device %s  CUs=%d  clock=%d kHz  wall-clock(s_memtime)=%d kHz\n", prop.name, cus,
         prop.clockRate, prop.clockInstructionRate);
  Entry es[] = {
    {"v_add_u32", k_add, 32}, {"v_xor_b32", k_xor, 32}, {"v_add3_u32", k_add3, 32},
    {"v_lshl_add_u32", k_lshladd, 32}, {"v_and_or_b32", k_andor, 32}, {"v_alignbit_b32", k_alignbit, 32},
    {"v_bfe_u32", k_bfe, 32}, {"v_perm_b32", k_perm, 32}, {"v_cndmask_b32(vcc,dep)", k_cndmask, 32},
    {"v_cndmask_b32_e64(sgpr)", k_cndmask64, 32}, {"v_cndmask_b32(vcc,indep)", k_cndmaski, 32},
    {"v_bfi_b32", k_bfi, 32}, {"v_and_b32", k_and, 32}, {"v_or_b32", k_or, 32}, {"v_lshlrev_b32", k_lshl, 32},
    {"v_lshrrev_b32", k_lshr, 32}, {"v_sub_u32", k_sub, 32}, {"v_mov_b32", k_mov, 32}, {"v_xad_u32", k_xad, 32},
    {"v_add_lshl_u32", k_addlshl, 32}, {"v_lshl_or_b32", k_lshlor, 32},
    {"v_add_co_u32", k_addco, 32}, {"v_addc_co_u32", k_addc, 32},
    {"v_mul_lo_u32", k_mullo, 32}, {"v_mul_hi_u32", k_mulhi, 32},
    {"v_mul_u32_u24", k_mulu24, 32}, {"v_mul_hi_u32_u24", k_mulhiu24, 32},
    {"v_mad_u32_u24", k_madu24, 32}, {"v_mad_i32_i24", k_madi24, 32}, {"v_mad_u32_u16", k_madu16, 32},
    {"v_dot4_u32_u8", k_dot4u8, 32}, {"v_dot2_u32_u16", k_dot2u16, 32}, {"v_dot8_u32_u4", k_dot8u4, 32},
    {"v_pk_mad_u16", k_pkmadu16, 32}, {"v_pk_add_u16", k_pkaddu16, 32},
    {"v_fma_f32", k_fmaf32, 32}, {"v_pk_fma_f32", k_pkfmaf32, 32}, {"v_pk_add_f32", k_pkaddf32, 32},
    {"v_mad_u64_u32(vcc)", k_mad64, 32}, {"v_mad_u64_u32(sgpr)", k_mad64s, 32}, {"v_mad_i64_i32", k_madi64, 32},
    {"v_fma_f64", k_fmaf64, 32}, {"v_add_f64", k_addf64, 32}, {"v_mul_f64", k_mulf64, 32},
    {"v_lshl_add_u64", k_lshladd64, 32}, {"v_lshrrev_b64", k_lshr64, 32}, {"v_ashrrev_i64", k_ashr64, 32},
    {"mix(mad64+add)", k_mix, 32},
  };
  const int wps_list[] = {1, 2, 4};
  uint32_t* out; uint64_t* cyc;
  size_t maxthreads = (size_t)cus * 256 * 4;
  CK(hipMalloc(&out, maxthreads * 4)); CK(hipMalloc(&cyc, maxthreads / 64 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("%-22s %4s %12s %12s %14s\n", "instr", "w/S", "cyc/instr", "cyc/instr/w", "Ginstr-lane/s");
  for (auto& e : es) {
    for (int wps : wps_list) {
      int blocks = cus * wps;  // 256 threads = 4 waves = one per SIMD
      e.k<<<blocks, 256>>>(out, cyc, 12345u, 67890u);  // warm
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      e.k<<<blocks, 256>>>(out, cyc, 12345u, 67890u);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      std::vector<uint64_t> h(blocks * 4);
      CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
      double avg = 0; for (auto v : h) avg += (double)v; avg /= h.size();
      double n = (double)ITERS * e.per_iter;
      // s_memtime ticks at a fixed 100 MHz-derived rate?  report both tick-based and wall-based
      double lane_ops = n * 64.0 * blocks * 4;
      printf("%-22s %4d %12.3f %12.3f %14.1f   (%.3f ms)\n", e.name, wps, avg / n * 1.0, avg / n / wps,
             lane_ops / (ms * 1e-3) / 1e9, ms);
    }
  }
  return 0;
}
