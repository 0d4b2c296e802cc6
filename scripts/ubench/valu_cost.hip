// Issue cost of the vector instructions the BAQ kernels are made of (gfx950): one wave per SIMD (256-thread blocks, one block per CU),
// 8 independent chains per wave, every chain a string of ONE opcode (inline asm, so that the compiler neither fuses nor removes it);
// clocks per instruction = s_memtime ticks / instructions.  Not product code.
//   hipcc -O3 --offload-arch=gfx950 valu_cost.hip -o valu_cost && ./valu_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
#define LOOPS 4096
#define CH8(body) body(0) body(1) body(2) body(3) body(4) body(5) body(6) body(7)
#define KERNEL(name, asm_line)                                                                                                   \
    __global__ void __launch_bounds__(256) name(double *out, double a, double b, long long *cyc)                                 \
    {                                                                                                                            \
        double x0 = a + threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;  \
        double y = b; unsigned long long m = 0; (void)m; (void)y;                                                                 \
        unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7, uy = (unsigned)blockIdx.x + 3u; (void)uy; \
        (void)u0; (void)u1; (void)u2; (void)u3; (void)u4; (void)u5; (void)u6; (void)u7;                                                                 \
        long long t0 = clock64();                                                                                                \
        for (int i = 0; i < LOOPS; ++i) {                                                                                        \
            _Pragma("unroll") for (int u = 0; u < REP / 8; ++u) { asm_line }                                                     \
        }                                                                                                                        \
        long long t1 = clock64();                                                                                                \
        out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (double)m + (double)(u0 + u1 + u2 + u3 + u4 + u5 + u6 + u7);                                  \
        if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;                                                                 \
    }
#define A1(op) asm volatile(op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" op " %4, %4, %8\n" op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8" \
                            : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y));
KERNEL(k_mul, A1("v_mul_f64"))
KERNEL(k_add, A1("v_add_f64"))
KERNEL(k_max, A1("v_max_f64"))
#define A_FMA asm volatile("v_fma_f64 %0, %0, %8, %8\nv_fma_f64 %1, %1, %8, %8\nv_fma_f64 %2, %2, %8, %8\nv_fma_f64 %3, %3, %8, %8\nv_fma_f64 %4, %4, %8, %8\nv_fma_f64 %5, %5, %8, %8\nv_fma_f64 %6, %6, %8, %8\nv_fma_f64 %7, %7, %8, %8" \
                            : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y));
KERNEL(k_fma, A_FMA)
#define A_RCP asm volatile("v_rcp_f64 %0, %0\nv_rcp_f64 %1, %1\nv_rcp_f64 %2, %2\nv_rcp_f64 %3, %3\nv_rcp_f64 %4, %4\nv_rcp_f64 %5, %5\nv_rcp_f64 %6, %6\nv_rcp_f64 %7, %7" \
                            : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
KERNEL(k_rcp, A_RCP)
#define A_LDEXP asm volatile("v_ldexp_f64 %0, %0, 1\nv_ldexp_f64 %1, %1, 1\nv_ldexp_f64 %2, %2, 1\nv_ldexp_f64 %3, %3, 1\nv_ldexp_f64 %4, %4, 1\nv_ldexp_f64 %5, %5, 1\nv_ldexp_f64 %6, %6, 1\nv_ldexp_f64 %7, %7, 1" \
                            : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
KERNEL(k_ldexp, A_LDEXP)
#define A_CMP asm volatile("v_cmp_gt_f64 vcc, %0, %8\nv_cmp_gt_f64 vcc, %1, %8\nv_cmp_gt_f64 vcc, %2, %8\nv_cmp_gt_f64 vcc, %3, %8\nv_cmp_gt_f64 vcc, %4, %8\nv_cmp_gt_f64 vcc, %5, %8\nv_cmp_gt_f64 vcc, %6, %8\nv_cmp_gt_f64 vcc, %7, %8" \
                            : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y) : "vcc");
KERNEL(k_cmp, A_CMP)
#define A_CMPU64 asm volatile("v_cmp_eq_u64 vcc, %0, %8\nv_cmp_eq_u64 vcc, %1, %8\nv_cmp_eq_u64 vcc, %2, %8\nv_cmp_eq_u64 vcc, %3, %8\nv_cmp_eq_u64 vcc, %4, %8\nv_cmp_eq_u64 vcc, %5, %8\nv_cmp_eq_u64 vcc, %6, %8\nv_cmp_eq_u64 vcc, %7, %8" \
                            : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y) : "vcc");
KERNEL(k_cmpu64, A_CMPU64)
#define A_CND asm volatile("v_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %8, vcc\nv_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\nv_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc" \
                            : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(uy) : "vcc");
KERNEL(k_cnd, A_CND)
#define A_AND asm volatile("v_and_b32 %0, %0, %8\nv_and_b32 %1, %1, %8\nv_and_b32 %2, %2, %8\nv_and_b32 %3, %3, %8\nv_and_b32 %4, %4, %8\nv_and_b32 %5, %5, %8\nv_and_b32 %6, %6, %8\nv_and_b32 %7, %7, %8" \
                            : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(uy));
KERNEL(k_and, A_AND)
#define A_DIVFIX asm volatile("v_div_fixup_f64 %0, %0, %8, %8\nv_div_fixup_f64 %1, %1, %8, %8\nv_div_fixup_f64 %2, %2, %8, %8\nv_div_fixup_f64 %3, %3, %8, %8\nv_div_fixup_f64 %4, %4, %8, %8\nv_div_fixup_f64 %5, %5, %8, %8\nv_div_fixup_f64 %6, %6, %8, %8\nv_div_fixup_f64 %7, %7, %8, %8" \
                            : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y));
KERNEL(k_divfix, A_DIVFIX)

#define A_CND64 asm volatile("v_cndmask_b32_e64 %0, %0, %8, s[10:11]\nv_cndmask_b32_e64 %1, %1, %8, s[10:11]\nv_cndmask_b32_e64 %2, %2, %8, s[10:11]\nv_cndmask_b32_e64 %3, %3, %8, s[10:11]\nv_cndmask_b32_e64 %4, %4, %8, s[10:11]\nv_cndmask_b32_e64 %5, %5, %8, s[10:11]\nv_cndmask_b32_e64 %6, %6, %8, s[10:11]\nv_cndmask_b32_e64 %7, %7, %8, s[10:11]" \
                            : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(uy) : "s10", "s11");
KERNEL(k_cnd64, A_CND64)
#define A_BFI asm volatile("v_bfi_b32 %0, %8, %0, %8\nv_bfi_b32 %1, %8, %1, %8\nv_bfi_b32 %2, %8, %2, %8\nv_bfi_b32 %3, %8, %3, %8\nv_bfi_b32 %4, %8, %4, %8\nv_bfi_b32 %5, %8, %5, %8\nv_bfi_b32 %6, %8, %6, %8\nv_bfi_b32 %7, %8, %7, %8" \
                            : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(uy));
KERNEL(k_bfi, A_BFI)
#define A_BFE asm volatile("v_bfe_i32 %0, %0, 3, 1\nv_bfe_i32 %1, %1, 3, 1\nv_bfe_i32 %2, %2, 3, 1\nv_bfe_i32 %3, %3, 3, 1\nv_bfe_i32 %4, %4, 3, 1\nv_bfe_i32 %5, %5, 3, 1\nv_bfe_i32 %6, %6, 3, 1\nv_bfe_i32 %7, %7, 3, 1" \
                            : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7));
KERNEL(k_bfe, A_BFE)
#define A_MOV asm volatile("v_mov_b32 %0, %8\nv_mov_b32 %1, %8\nv_mov_b32 %2, %8\nv_mov_b32 %3, %8\nv_mov_b32 %4, %8\nv_mov_b32 %5, %8\nv_mov_b32 %6, %8\nv_mov_b32 %7, %8" \
                            : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(uy));
KERNEL(k_mov, A_MOV)
// the realistic pair: a compare of doubles, then the two halves of a double selected on it
#define A_CMPSEL asm volatile("v_cmp_gt_f64 vcc, %0, %8\nv_cndmask_b32 %4, %4, %9, vcc\nv_cndmask_b32 %5, %5, %9, vcc\nv_cmp_gt_f64 vcc, %1, %8\nv_cndmask_b32 %6, %6, %9, vcc\nv_cndmask_b32 %7, %7, %9, vcc\nv_cmp_gt_f64 vcc, %2, %8\nv_cndmask_b32 %4, %4, %9, vcc\nv_cndmask_b32 %5, %5, %9, vcc" \
                            : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(y), "v"(uy) : "vcc");
KERNEL(k_cmpsel, A_CMPSEL)
template <class K> void run(const char *name, K kern, int per_rep, int blocks_per_cu)
{
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    double *out; long long *cyc;
    hipMalloc(&out, (size_t)cus * blocks_per_cu * 256 * 8); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(cus * blocks_per_cu), dim3(256), 0, 0, out, 1.0000001, 0.9999999, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    const double per_wave = (double)LOOPS * (REP / 8) * per_rep;
    const double ns_per_instr_simd = ms * 1e6 / (per_wave * blocks_per_cu);      // one wave of every block on each SIMD
    printf("%-20s blocks/CU %d: %.3f ms, %.2f ns per instruction per SIMD = %.2f clocks at 2.4 GHz\n", name, blocks_per_cu, ms, ns_per_instr_simd, ns_per_instr_simd * 2.4);
    hipFree(out); hipFree(cyc);
}
int main()
{
    for (int w : { 1, 8 }) {
        run("v_mul_f64", k_mul, 8, w); run("v_add_f64", k_add, 8, w); run("v_fma_f64", k_fma, 8, w); run("v_max_f64", k_max, 8, w);
        run("v_cmp_gt_f64", k_cmp, 8, w); run("v_cndmask_b32", k_cnd, 8, w); run("v_cmp_eq_u64", k_cmpu64, 8, w);
        run("v_and_b32", k_and, 8, w); run("v_cndmask_e64 sgpr", k_cnd64, 8, w); run("v_bfi_b32", k_bfi, 8, w); run("v_bfe_i32", k_bfe, 8, w); run("v_mov_b32", k_mov, 8, w); run("cmp_f64+2cndmask (x3)", k_cmpsel, 9, w); run("v_rcp_f64", k_rcp, 8, w); run("v_ldexp_f64", k_ldexp, 8, w); run("v_div_fixup_f64", k_divfix, 8, w);
    }
    return 0;
}
