// ubench_valu.hip — issue rate of the integer vector instructions the banded-SW kernel is made of (gfx950).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o tools/ubench_valu && tools/ubench_valu
// Every lane runs kIter x 32 instructions of one kind on 8 independent registers; the grid is `waves_per_simd` resident
// waves on every SIMD.  Prints wave-instructions per second for the whole chip and cycles per instruction per SIMD at
// 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int kIter = 4096;

#define REP8(OP)                                                                                                        \
    OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)
#define BODY(NAME, ASM)                                                                                                 \
    __global__ __launch_bounds__(256) void NAME(int *out, int x) {                                                      \
        const unsigned long long msk = 0x5555aaaa5555aaaaull ^ (unsigned long long)x;                                  \
        int sx = x;                                                                                                     \
        int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        for (int i = 0; i < kIter; ++i) {                                                                               \
            REP8(ASM) REP8(ASM) REP8(ASM) REP8(ASM)                                                                     \
        }                                                                                                               \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + sx;                           \
    }

#define OP_MAX(r) asm volatile("v_max_i32 %0, %0, %1" : "+v"(r) : "v"(x));
#define OP_ADD(r) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r) : "v"(x));
#define OP_CND(r) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(r) : "v"(x), "s"(msk));
#define OP_NP0(r) asm volatile("v_max_i32 %0, %0, %1\n\ts_nop 0" : "+v"(r) : "v"(x));
#define OP_NP4(r) asm volatile("v_max_i32 %0, %0, %1\n\ts_nop 4" : "+v"(r) : "v"(x));
#define OP_SAL(r) asm volatile("v_max_i32 %0, %0, %2\n\ts_and_b32 %1, %1, 0x7ffffff" : "+v"(r), "+s"(sx) : "v"(x));
#define OP_CMC(r) asm volatile("v_cmp_lt_i32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(r) : "v"(x) : "vcc");
#define OP_MUL(r) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r) : "v"(x));
#define OP_M24(r) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r) : "v"(x));
#define OP_PKM(r) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(r) : "v"(x));
#define OP_PKA(r) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(r) : "v"(x));
#define OP_MX3(r) asm volatile("v_max3_i32 %0, %0, %1, %1" : "+v"(r) : "v"(x));
#define OP_DPP(r) asm volatile("v_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r));
#define OP_BFE(r) asm volatile("v_bfe_i32 %0, %0, %1, 8" : "+v"(r) : "v"(x));
#define OP_CMP(r) asm volatile("v_cmp_lt_i32 vcc, %0, %1" : : "v"(r), "v"(x) : "vcc");
#define OP_PRM(r) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(r) : "v"(x));

BODY(k_max, OP_MAX)
BODY(k_add, OP_ADD)
BODY(k_cnd, OP_CND)
BODY(k_mul, OP_MUL)
BODY(k_m24, OP_M24)
BODY(k_pkm, OP_PKM)
BODY(k_pka, OP_PKA)
BODY(k_mx3, OP_MX3)
BODY(k_dpp, OP_DPP)
BODY(k_bfe, OP_BFE)
BODY(k_cmp, OP_CMP)
BODY(k_prm, OP_PRM)
BODY(k_np0, OP_NP0)
BODY(k_np4, OP_NP4)
BODY(k_sal, OP_SAL)
BODY(k_cmc, OP_CMC)

int main(int argc, char **argv) {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    int *out;
    hipMalloc(&out, (size_t)cus * 16 * 256 * sizeof(int));
    struct { const char *name; void (*fn)(int *, int); } ks[] = {
        {"v_max_i32", k_max}, {"v_add_u32", k_add}, {"v_cndmask_b32", k_cnd}, {"v_mul_lo_u32", k_mul}, {"v_mul_u32_u24", k_m24},
        {"v_pk_max_i16", k_pkm}, {"v_pk_add_i16", k_pka}, {"v_max3_i32", k_mx3}, {"v_max_i32_dpp", k_dpp}, {"v_bfe_i32", k_bfe},
        {"v_cmp_lt_i32", k_cmp}, {"v_perm_b32", k_prm}, {"max+s_nop 0", k_np0}, {"max+s_nop 4", k_np4},
        {"max+s_and", k_sal}, {"cmp+cndmask", k_cmc}};
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("%d CUs, clock %d kHz\n", cus, p.clockRate);
    for (int wps : {1, 2, 4}) {                       // waves per SIMD: a block = 4 waves = one wave per SIMD of a CU
        for (auto &k : ks) {
            hipLaunchKernelGGL(k.fn, dim3(cus * wps), dim3(256), 0, 0, out, 3);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k.fn, dim3(cus * wps), dim3(256), 0, 0, out, 3);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double insts = (double)cus * wps * 4 * kIter * 32;
            const double per_simd_cycles = (ms * 1e-3 * 2.4e9) / ((double)wps * kIter * 32);
            printf("%-16s %d waves/SIMD: %8.1f G wave-inst/s, %.2f cycles per instruction per SIMD (at 2.4 GHz)\n", k.name, wps,
                   insts / (ms * 1e-3) * 1e-9, per_simd_cycles);
        }
    }
    return 0;
}
