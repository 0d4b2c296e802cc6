// kernels_cons.hip -- the consensus path on the device (gfx950): one lane per read for the read-shaped steps, one lane per
// (position, nth) column for the callers.  The arithmetic lives in cons_core.h / cons_window.h (shared with the CPU harness of
// the tests); this file is the launch geometry.  Built with -ffp-contract=off: the Bayesian caller's fp64 sums and the
// fast_log2 polynomial must round exactly as the reference's unfused x86-64 code does.
//
// Bound: integer / byte work with data-dependent control flow, HBM traffic = entry words (4 B, + 4 B in the Bayesian mode
// with mapping qualities) written once by the read walk and read once by the column caller, plus the staged reads.  No MFMA.
#include "sta_dev.h"
#include "cons_window.h"

using namespace cons;

__global__ void __launch_bounds__(256) k_cons_read_a(Win w, Par o, const Tables *t)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= w.n_reads) return;
    step_read_a(w, o, *t, r, [](uint32_t *p, uint32_t v) { atomicMax(p, v); }, [](unsigned long long *p, unsigned long long v) { atomicAdd(p, v); });
}

// lengths for the column-index scan: position i of the window owns 1 + ins[i + 1] columns (ins[0] is the look-back position)
__global__ void __launch_bounds__(256) k_cons_collen(const uint32_t *ins, uint32_t *len, int64_t W)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < W) len[i] = 1u + ins[i + 1];
}

__global__ void __launch_bounds__(256) k_cons_read_b(Win w)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < w.n_reads) step_read_b(w, r);
}

__global__ void __launch_bounds__(256) k_cons_walk(Win w, Par o)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < w.n_reads) step_walk(w, o, r);
}

__global__ void __launch_bounds__(256) k_cons_col(Win w, Par o, const Tables *t, int64_t n_cols)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n_cols) step_col(w, o, *t, c);
}

__global__ void __launch_bounds__(256) k_cons_text(Win w, int64_t n_cols)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n_cols) step_text(w, c);
}

static inline unsigned blocks_for(int64_t n) { return (unsigned)((n + 255) / 256); }

void sta_launch_cons_read_a(hipStream_t s, const Win &w, const Par &o, const Tables *t)
{
    if (w.n_reads > 0) hipLaunchKernelGGL(k_cons_read_a, dim3(blocks_for(w.n_reads)), dim3(256), 0, s, w, o, t);
}
void sta_launch_cons_collen(hipStream_t s, const uint32_t *ins, uint32_t *len, int64_t W)
{
    if (W > 0) hipLaunchKernelGGL(k_cons_collen, dim3(blocks_for(W)), dim3(256), 0, s, ins, len, W);
}
void sta_launch_cons_read_b(hipStream_t s, const Win &w)
{
    if (w.n_reads > 0) hipLaunchKernelGGL(k_cons_read_b, dim3(blocks_for(w.n_reads)), dim3(256), 0, s, w);
}
void sta_launch_cons_walk(hipStream_t s, const Win &w, const Par &o)
{
    if (w.n_reads > 0) hipLaunchKernelGGL(k_cons_walk, dim3(blocks_for(w.n_reads)), dim3(256), 0, s, w, o);
}
void sta_launch_cons_col(hipStream_t s, const Win &w, const Par &o, const Tables *t, int64_t n_cols)
{
    if (n_cols > 0) hipLaunchKernelGGL(k_cons_col, dim3(blocks_for(n_cols)), dim3(256), 0, s, w, o, t, n_cols);
}
void sta_launch_cons_text(hipStream_t s, const Win &w, int64_t n_cols)
{
    if (n_cols > 0) hipLaunchKernelGGL(k_cons_text, dim3(blocks_for(n_cols)), dim3(256), 0, s, w, n_cols);
}
