// fp64 VALU throughput per SIMD against the number of resident waves (gfx950): how many waves does a SIMD need before its fp64
// pipe is full?  Every wave runs CH independent chains of dependent v_mul_f64 / v_add_f64 (no memory); blocks of 256 threads
// = one wave per SIMD, B blocks per CU resident at once (grid = CUs x B, one round).  Wall time by HIP events.
// Not product code.  hipcc -O3 --offload-arch=gfx950 fp64_occ.hip -o fp64_occ
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 8192
template <int CH> __global__ void __launch_bounds__(256) k(double *out, double a, double b, long long *cyc)
{
    double x[CH];
    for (int c = 0; c < CH; ++c) x[c] = a + threadIdx.x + c;
    long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < N / 8; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) { x[c] = x[c] * b; x[c] = x[c] + a; }
    }
    long long t1 = clock64();
    double s = 0; for (int c = 0; c < CH; ++c) s += x[c];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int CH> void run(int cus, int blocks_per_cu)
{
    double *out; long long *cyc, h;
    hipMalloc(&out, (size_t)cus * blocks_per_cu * 256 * 8); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<CH>), dim3(cus * blocks_per_cu), dim3(256), 0, 0, out, 1.0000001, 0.9999999, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double instr_per_wave = 2.0 * N * CH;
    const double simd_instr = instr_per_wave * blocks_per_cu;          // one wave of every block on each SIMD
    printf("chains %d  waves/SIMD %d : %.3f ms  -> %.2f ns per fp64 instr per SIMD (= %.2f clocks at 2.4 GHz);  clock64 ticks per instr per wave %.2f, ticks per us %.0f\n",
           CH, blocks_per_cu, ms, ms * 1e6 / simd_instr, ms * 1e6 / simd_instr * 2.4, (double)h / instr_per_wave, (double)h / (ms * 1e3));
    hipFree(out); hipFree(cyc);
}
int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("CUs %d, clockRate %d kHz\n", cus, p.clockRate);
    for (int b : { 1, 2, 3, 4, 6, 8 }) run<1>(cus, b);
    for (int b : { 1, 2, 3, 4, 6, 8 }) run<4>(cus, b);
    for (int b : { 1, 2, 4, 8 }) run<8>(cus, b);
    return 0;
}
