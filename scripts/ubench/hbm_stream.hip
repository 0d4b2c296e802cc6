// HBM streaming ceilings on gfx950 for the access shapes the BAQ kernels use (16 B per lane, non-temporal): pure write,
// pure read, copy.  Not product code; used to put the BAQ row stream in perspective.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
__global__ void k_write(d2 *p, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, s = (size_t)gridDim.x * blockDim.x; d2 v = { 1.0 + i, 2.0 };
    for (; i < n; i += s) __builtin_nontemporal_store(v, p + i); }
__global__ void k_read(const d2 *p, size_t n, double *out) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, s = (size_t)gridDim.x * blockDim.x; double a = 0;
    for (; i < n; i += s) { d2 v = __builtin_nontemporal_load(p + i); a += v.x + v.y; } if (a == 12345.678) out[0] = a; }
__global__ void k_copy(const d2 *p, d2 *q, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, s = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += s) __builtin_nontemporal_store(__builtin_nontemporal_load(p + i), q + i); }
// the BAQ forward shape: every wave owns a 2.4 MB slab and writes 15 x 1 KiB per "row"
__global__ void k_write_slab(d2 *p, size_t slab_d2, int rows) { size_t w = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; int lane = threadIdx.x & 63; d2 *b = p + w * slab_d2 + lane; d2 v = { 1.0 + w, 2.0 };
    for (int r = 0; r < rows; ++r) for (int c = 0; c < 15; ++c) __builtin_nontemporal_store(v, b + ((size_t)r * 15 + c) * 64); }
int main()
{
    size_t bytes = (size_t)16 << 30, n = bytes / 16;
    d2 *a, *b; double *o; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0); hipLaunchKernelGGL(k_write, dim3(256 * 16), dim3(256), 0, 0, a, n); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("write  %.2f TB/s\n", bytes / ms / 1e9);
        hipEventRecord(e0); hipLaunchKernelGGL(k_read, dim3(256 * 16), dim3(256), 0, 0, a, n, o); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("read   %.2f TB/s\n", bytes / ms / 1e9);
        hipEventRecord(e0); hipLaunchKernelGGL(k_copy, dim3(256 * 16), dim3(256), 0, 0, a, b, n); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("copy   %.2f TB/s (read+write)\n", 2.0 * bytes / ms / 1e9);
        size_t slab = (size_t)150 * 15 * 64, waves = bytes / 16 / slab;
        hipEventRecord(e0); hipLaunchKernelGGL(k_write_slab, dim3((unsigned)(waves / 4)), dim3(256), 0, 0, a, slab, 150); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("write, BAQ slab shape (%zu waves) %.2f TB/s\n", waves / 4 * 4, (waves / 4 * 4) * slab * 16.0 / ms / 1e9);
    }
    return 0;
}
