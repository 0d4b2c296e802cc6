// Calibration of the TCC counters FETCH_SIZE / WRITE_SIZE on gfx950: kernels that move a KNOWN number of bytes
// with the access widths the engine's kernels use -- 16, 8, 4 and 1 byte per lane, temporal and non-temporal -- so that
// `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes) can be compared with the truth per access shape.
// Not product code.  Build: hipcc -O3 --offload-arch=gfx950 pmc_calib.hip -o pmc_calib ; run under rocprofv3 --kernel-trace --pmc X.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef double d2 __attribute__((ext_vector_type(2)));
#define GRID dim3(256 * 16), dim3(256)
template <typename T, bool NT> __global__ void k_rd(const T *p, size_t n, T *out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, s = (size_t)gridDim.x * blockDim.x;
    uint64_t a = 0;
    for (; i < n; i += s) { T v = NT ? __builtin_nontemporal_load(p + i) : p[i]; const unsigned char *b = (const unsigned char *)&v; for (unsigned k = 0; k < sizeof(T); ++k) a += b[k]; }
    if (a == 0x123456789abcull) out[0] = p[0];
}
template <typename T, bool NT> __global__ void k_wr(T *p, size_t n, T v)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, s = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += s) { if (NT) __builtin_nontemporal_store(v, p + i); else p[i] = v; }
}
template <typename T, bool NT> static void run(const char *name, void *buf, size_t bytes, void *out)
{
    size_t n = bytes / sizeof(T);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms;
    T v; unsigned char *b = (unsigned char *)&v; for (unsigned k = 0; k < sizeof(T); ++k) b[k] = 1;
    hipEventRecord(e0); hipLaunchKernelGGL((k_wr<T, NT>), GRID, 0, 0, (T *)buf, n, v); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("write %-8s %zu bytes  %.2f TB/s\n", name, bytes, bytes / ms / 1e9);
    hipEventRecord(e0); hipLaunchKernelGGL((k_rd<T, NT>), GRID, 0, 0, (const T *)buf, n, (T *)out); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("read  %-8s %zu bytes  %.2f TB/s\n", name, bytes, bytes / ms / 1e9);
}
int main()
{
    size_t bytes = (size_t)4 << 30;       // well beyond the 256 MB of MALL: every byte comes from / goes to HBM
    void *a, *o; hipMalloc(&a, bytes); hipMalloc(&o, 64);
    run<d2, true>("16B_nt", a, bytes, o);
    run<d2, false>("16B", a, bytes, o);
    run<double, true>("8B_nt", a, bytes, o);
    run<double, false>("8B", a, bytes, o);
    run<uint32_t, false>("4B", a, bytes, o);
    run<uint8_t, false>("1B", a, bytes / 4, o);
    hipDeviceSynchronize();
    return 0;
}
