// fp64 VALU latency / throughput probe for gfx950 (used to size the BAQ kernels; not product code)
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 4096
template <int CH, int OP> __global__ void k(double *out, double a, double b, long long *cyc)
{
    double x[CH];
    for (int c = 0; c < CH; ++c) x[c] = a + threadIdx.x + c;
    long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < N / 16; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (OP == 0) x[c] = x[c] * b;
                else if (OP == 1) x[c] = x[c] + b;
                else if (OP == 2) x[c] = __builtin_fma(x[c], b, a);
                else { x[c] = x[c] * b; x[c] = x[c] + a; }
            }
    }
    long long t1 = clock64();
    double s = 0; for (int c = 0; c < CH; ++c) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int CH, int OP> void run(const char *name, int waves_per_simd)
{
    double *out; long long *cyc, h;
    hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 8);
    int threads = 256 * waves_per_simd;           // 4 SIMDs x waves_per_simd waves in one block on one CU
    if (threads > 1024) threads = 1024;
    hipLaunchKernelGGL((k<CH, OP>), dim3(1), dim3(threads), 0, 0, out, 1.0000001, 0.9999999, cyc);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    long long ops = (long long)N * CH * (OP == 3 ? 2 : 1);
    printf("%-28s chains=%d waves/SIMD=%d : %.2f cycles per instr per wave (%.2f per SIMD)\n", name, CH, threads / 256, (double)h / ops, (double)h / ops / (threads / 256));
    hipFree(out); hipFree(cyc);
}
int main()
{
    run<1, 0>("v_mul_f64 dependent", 1); run<4, 0>("v_mul_f64", 1); run<8, 0>("v_mul_f64", 1); run<8, 0>("v_mul_f64", 2); run<8, 0>("v_mul_f64", 4);
    run<1, 1>("v_add_f64 dependent", 1); run<8, 1>("v_add_f64", 1); run<8, 1>("v_add_f64", 4);
    run<1, 2>("v_fma_f64 dependent", 1); run<8, 2>("v_fma_f64", 1); run<8, 2>("v_fma_f64", 4);
    run<1, 3>("mul+add dependent", 1); run<8, 3>("mul+add", 1); run<1, 3>("mul+add dependent", 3);
    return 0;
}
