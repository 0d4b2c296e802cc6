// hipemu.h -- TEST INFRASTRUCTURE ONLY.  A stand-in for <hip/hip_runtime.h> that lets the library's .hip / .cpp sources be compiled
// for the host CPU (clang++, plain C++17) and executed there: every kernel launch runs its workgroups one after the other on the
// calling thread, the threads of a workgroup are fibers, and the wave-level operations (ballot, shuffles, readlane, wavefront fences,
// __syncthreads) are meeting points of those fibers.  It exists so that the `-m gpu` parity tests -- the same tests, the same oracle
// comparisons -- can be run in a container without a GPU (and under AddressSanitizer, which sees every out-of-bounds "device" access).
//
// It is NOT part of the product and never a fallback: the emulated library is built by tests/cpu/hipemu/Makefile into
// tests/cpu/hipemu/_build/, a directory the package (samtools_amd/_capi.py), bench.py and __graft_entry__ never look at; only
// tests/conftest.py selects it, and only when STA_HIPEMU=1 is set by hand.  Nothing here says anything about performance.
//
// What it models and what it does not:
//   * wave64, lanes of a wave meet at every wave-level operation; the lanes that meet are those waiting at the SAME call site (so an
//     operation inside a divergent branch sees the lanes that took the branch, like the EXEC mask).  When lanes of one wave wait at
//     different sites, the order is that of the control-flow graph (mkloops.py reads it out of the built library): fewer trips round a
//     shared loop first, then the earlier block in reverse post-order -- the inner divergent region runs to its end before the lanes
//     that wait at the join point go on.  README.md has the details; selftest.cpp checks them on hand-worked kernels.
//   * workgroups run sequentially in blockIdx order: look-back / ticket schemes that wait for EARLIER workgroups work, a kernel that
//     waits for a later one is reported as a deadlock.
//   * streams and events are ordered by program order (everything is synchronous); LDS is thread-local storage of the OS thread.
//   * fences at wavefront / workgroup scope are treated as meeting points of the wave (lock-step visibility of LDS and memory).
#pragma once
#define HIPEMU 1
// the sources pick their device variants (the host pass of hipcc never sees those; here one pass compiles both sides)
#define __HIPCC__ 1
#define __HIP_DEVICE_COMPILE__ 1
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <tuple>
#include <type_traits>
#include <utility>

// ---- language ----
// Device functions are `convergent`, as the device compiler makes every function of a HIP translation unit until it has proved
// otherwise: a call of a function that contains a wave operation must not be duplicated into the arms of a lane-varying branch any more
// than the operation itself (seen with the AddressSanitizer build: a not-yet-inlined file_pass() call cloned for `active == false`).
#define __global__ __attribute__((convergent))
#define __device__ __attribute__((convergent))
#define __host__
#define __constant__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#define __shared__ thread_local
#define HIP_KERNEL_NAME(...) __VA_ARGS__
// target attributes of the kernels: __attribute__((amdgpu_waves_per_eu(2, 2))) becomes an empty attribute
#define amdgpu_waves_per_eu(...)
#define amdgpu_flat_work_group_size(...)
#define address_space(n)

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct alignas(16) double2 { double x, y; };
static inline uint2 make_uint2(unsigned x, unsigned y) { uint2 v = { x, y }; return v; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { uint4 v = { x, y, z, w }; return v; }
static inline int2 make_int2(int x, int y) { int2 v = { x, y }; return v; }
static inline int4 make_int4(int x, int y, int z, int w) { int4 v = { x, y, z, w }; return v; }
static inline double2 make_double2(double x, double y) { double2 v = { x, y }; return v; }

namespace hipemu {

enum { K_BALLOT = 1, K_SHFL, K_SHFL_UP, K_SHFL_DOWN, K_SHFL_XOR, K_READLANE, K_READFIRST, K_SYNC };

struct Lane {
    void *sp;                 // saved stack pointer of the fiber
    dim3 tid;
    int lin;                  // linear thread index in the workgroup
    int state;
    const void *site; int kind; uint64_t val; int arg, width; uint64_t res;
    const void *stack_sites[24]; int n_stack;      // the return addresses of the waiting lane, outermost first (site == the innermost)
    // where the lane is in the kernel's control-flow graph (hipemu.cpp, "which group first"): the loops around its block with the number
    // of times it has gone round each (outermost first), the block's number in reverse post-order, the code address
    int rpo; uintptr_t kpos; int n_loops; struct { int id; int count; } loops[12]; const void *blk;
    char *stack;
    uintptr_t low_sp;         // the lowest stack pointer the fiber was seen with (what an AddressSanitizer build unpoisons before the stack is used again)
};
struct Block { dim3 bid, bdim, gdim; };

extern thread_local Lane *tl_lane;
extern thread_local Block tl_block;

// `convergent`: what the device compiler knows about these operations -- a call may not be duplicated into, or moved under, control
// flow it was not written in (the host compiler would otherwise clone a ballot into both arms of a lane-varying branch: two sites for
// what the kernel wrote as one)
__attribute__((convergent)) uint64_t wave_op(int kind, uint64_t val, int arg, int width);
__attribute__((convergent)) void block_barrier();
void lane_yield();            // s_sleep in a spin loop: let the other lanes (and, one day, workgroups) run

template <class T> inline uint64_t to_bits(T v) { static_assert(sizeof(T) <= 8 && std::is_trivially_copyable<T>::value, "wave operand"); uint64_t u = 0; memcpy(&u, &v, sizeof(T)); return u; }
template <class T> inline T from_bits(uint64_t u) { T v; memcpy(&v, &u, sizeof(T)); return v; }

struct Launch {
    dim3 grid, block; size_t shmem;
    void (*invoke)(void *); void *closure; const void *kernel;
    const char *name;
};
void run(const Launch &l);

template <class K> struct KernelArgs;
template <class... P> struct KernelArgs<void (*)(P...)> {
    typedef std::tuple<typename std::decay<P>::type...> tuple;
    template <size_t... I> static void call(void (*k)(P...), tuple &t, std::index_sequence<I...>) { k(std::get<I>(t)...); }
    static void call(void (*k)(P...), tuple &t) { call(k, t, std::index_sequence_for<P...>()); }
};

// every lane of every workgroup calls the kernel with its own copy-constructed view of the same argument tuple
template <class K, class... A>
inline void launch_kernel(const char *name, K k, dim3 g, dim3 b, size_t shmem, void *stream, A &&...args)
{
    (void)stream;
    typedef typename std::decay<K>::type Fn;
    typedef KernelArgs<Fn> KA;
    typename KA::tuple t(std::forward<A>(args)...);
    struct C { Fn k; typename KA::tuple *t; } c = { k, &t };
    Launch l; l.kernel = (const void *)k; l.grid = g; l.block = b; l.shmem = shmem; l.closure = &c; l.name = name;
    l.invoke = [](void *p) { C *c = (C *)p; KA::call(c->k, *c->t); };
    run(l);
}

}  // namespace hipemu

#define threadIdx (hipemu::tl_lane->tid)
#define blockIdx (hipemu::tl_block.bid)
#define blockDim (hipemu::tl_block.bdim)
#define gridDim (hipemu::tl_block.gdim)

#define hipLaunchKernelGGL(k, g, b, shm, s, ...) hipemu::launch_kernel(#k, k, dim3(g), dim3(b), (size_t)(shm), (void *)(s), ##__VA_ARGS__)

// ---- wave-level operations (wave64) ----
#define HIPEMU_WOP inline __attribute__((always_inline))
// every wrapper is always_inline, so the return address taken inside wave_op's caller chain (wave_op itself is noinline) is unique per
// place in the kernel's code
HIPEMU_WOP unsigned long long __ballot(int pred) { return hipemu::wave_op(hipemu::K_BALLOT, pred != 0, 0, 64); }
#define __builtin_amdgcn_ballot_w64(p) __ballot((p) ? 1 : 0)
HIPEMU_WOP int __any(int pred) { return __ballot(pred) != 0; }
HIPEMU_WOP int __all(int pred) { return __ballot(!pred) == 0; }
template <class T> HIPEMU_WOP T __shfl(T v, int src, int width = 64) { return hipemu::from_bits<T>(hipemu::wave_op(hipemu::K_SHFL, hipemu::to_bits(v), src, width)); }
template <class T> HIPEMU_WOP T __shfl_up(T v, unsigned d, int width = 64) { return hipemu::from_bits<T>(hipemu::wave_op(hipemu::K_SHFL_UP, hipemu::to_bits(v), (int)d, width)); }
template <class T> HIPEMU_WOP T __shfl_down(T v, unsigned d, int width = 64) { return hipemu::from_bits<T>(hipemu::wave_op(hipemu::K_SHFL_DOWN, hipemu::to_bits(v), (int)d, width)); }
template <class T> HIPEMU_WOP T __shfl_xor(T v, int m, int width = 64) { return hipemu::from_bits<T>(hipemu::wave_op(hipemu::K_SHFL_XOR, hipemu::to_bits(v), m, width)); }
HIPEMU_WOP int __builtin_amdgcn_readlane(int v, int lane) { return hipemu::from_bits<int>(hipemu::wave_op(hipemu::K_READLANE, hipemu::to_bits(v), lane, 64)); }
HIPEMU_WOP int __builtin_amdgcn_readfirstlane(int v) { return hipemu::from_bits<int>(hipemu::wave_op(hipemu::K_READFIRST, hipemu::to_bits(v), 0, 64)); }
// v_mov_b32_dpp as the kernels use it (row_shr:n, row_bcast:15 / :31; bound_ctrl off): a lane whose row / bank is masked out, or whose
// source lies outside its row, keeps `old`.  Every lane of the wave takes part (the callers run it under full EXEC).
HIPEMU_WOP int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl)
{
    const int l = hipemu::tl_lane->lin & 63, row = l >> 4, in_row = l & 15;
    int from = -1;
    if (ctrl >= 0x111 && ctrl <= 0x11f) { const int n = ctrl - 0x110; if (in_row >= n) from = l - n; }
    else if (ctrl == 0x142) { if (row >= 1) from = 16 * (row - 1) + 15; }
    else if (ctrl == 0x143) { if (row >= 2) from = 31; }
    else __builtin_trap();
    const int got = hipemu::from_bits<int>(hipemu::wave_op(hipemu::K_SHFL, hipemu::to_bits(src), from < 0 ? l : from, 64));
    if (!((row_mask >> row) & 1) || !((bank_mask >> (in_row >> 2)) & 1)) return old;
    if (from < 0) return bound_ctrl ? 0 : old;
    return got;
}
// v_alignbyte_b32: ({hi, lo} >> 8 * (sh & 3)) & 0xffffffff
static inline unsigned __builtin_amdgcn_alignbyte(unsigned hi, unsigned lo, unsigned sh) { return (unsigned)(((((uint64_t)hi) << 32) | lo) >> (8 * (sh & 3u))); }
HIPEMU_WOP void hipemu_wave_sync() { (void)hipemu::wave_op(hipemu::K_SYNC, 0, 0, 64); }
#define __builtin_amdgcn_wave_barrier() hipemu_wave_sync()
#define __builtin_amdgcn_fence(order, scope) hipemu_wave_sync()
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_s_sleep(x) hipemu::lane_yield()
HIPEMU_WOP void __syncthreads() { hipemu::block_barrier(); }
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline void __threadfence_block() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }

static inline unsigned __builtin_amdgcn_mbcnt_lo(unsigned mask, unsigned base) { const int l = hipemu::tl_lane->lin & 63; return base + (unsigned)__builtin_popcount(l >= 32 ? mask : (mask & ((1u << l) - 1u))); }
static inline unsigned __builtin_amdgcn_mbcnt_hi(unsigned mask, unsigned base) { const int l = hipemu::tl_lane->lin & 63; return base + (l <= 32 ? 0u : (unsigned)__builtin_popcount(mask & ((1u << (l - 32)) - 1u))); }
static inline unsigned __builtin_amdgcn_perm(unsigned hi, unsigned lo, unsigned sel)
{
    // v_perm_b32: result byte i is picked by selector byte i out of the eight bytes { hi, lo } (0..3 = lo, 4..7 = hi); 12 = 0x00, 13..15 = 0xff
    const uint64_t src = ((uint64_t)hi << 32) | lo; unsigned r = 0;
    for (int i = 0; i < 4; ++i) {
        const unsigned s = (sel >> (8 * i)) & 0xff; unsigned b;
        if (s <= 7) b = (unsigned)(src >> (8 * s)) & 0xff;
        else if (s == 12) b = 0;
        else if (s >= 13) b = 0xff;
        else { const unsigned w = (s - 8) * 2 + 1; b = ((src >> (8 * w + 7)) & 1) ? 0xff : 0; }      // 8..11: sign of a 16-bit half
        r |= b << (8 * i);
    }
    return r;
}
static inline int __builtin_amdgcn_sbfe(int v, unsigned off, unsigned width) { off &= 31; width &= 31; if (!width) return 0; return (int)((unsigned)v << (32 - off - width)) >> (32 - width); }
static inline float __builtin_amdgcn_logf(float x) { return log2f(x); }

// ---- integer / conversion intrinsics ----
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __ffs(int x) { return __builtin_ffs(x); }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
static inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
static inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }
static inline long long __double_as_longlong(double x) { long long v; memcpy(&v, &x, 8); return v; }
static inline double __longlong_as_double(long long x) { double v; memcpy(&v, &x, 8); return v; }
static inline unsigned __float_as_uint(float x) { unsigned v; memcpy(&v, &x, 4); return v; }
static inline int __float_as_int(float x) { int v; memcpy(&v, &x, 4); return v; }
static inline float __uint_as_float(unsigned x) { float v; memcpy(&v, &x, 4); return v; }
static inline float __int_as_float(int x) { float v; memcpy(&v, &x, 4); return v; }
template <class T> static inline T __ldg(const T *p) { return *p; }

// ---- atomics (workgroups are sequential, but several host threads may each run their own launches) ----
template <class T, class U> static inline T atomicAdd(T *p, U v) { return __atomic_fetch_add(p, (T)v, __ATOMIC_RELAXED); }
static inline float atomicAdd(float *p, float v) { float o, n; do { o = *p; n = o + v; } while (!__atomic_compare_exchange(p, &o, &n, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)); return o; }
static inline double atomicAdd(double *p, double v) { double o, n; do { o = *p; n = o + v; } while (!__atomic_compare_exchange(p, &o, &n, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)); return o; }
template <class T, class U> static inline T atomicSub(T *p, U v) { return __atomic_fetch_sub(p, (T)v, __ATOMIC_RELAXED); }
template <class T, class U> static inline T atomicOr(T *p, U v) { return __atomic_fetch_or(p, (T)v, __ATOMIC_RELAXED); }
template <class T, class U> static inline T atomicAnd(T *p, U v) { return __atomic_fetch_and(p, (T)v, __ATOMIC_RELAXED); }
template <class T, class U> static inline T atomicExch(T *p, U v) { return __atomic_exchange_n(p, (T)v, __ATOMIC_RELAXED); }
template <class T, class U> static inline T atomicMax(T *p, U v) { T o = __atomic_load_n(p, __ATOMIC_RELAXED); while (o < (T)v && !__atomic_compare_exchange_n(p, &o, (T)v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} return o; }
template <class T, class U> static inline T atomicMin(T *p, U v) { T o = __atomic_load_n(p, __ATOMIC_RELAXED); while (o > (T)v && !__atomic_compare_exchange_n(p, &o, (T)v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} return o; }
template <class T, class U, class V> static inline T atomicCAS(T *p, U cmp, V v) { T o = (T)cmp; __atomic_compare_exchange_n(p, &o, (T)v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED); return o; }
#define __HIP_MEMORY_SCOPE_SINGLETHREAD 1
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), __ATOMIC_SEQ_CST)
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), __ATOMIC_SEQ_CST)
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add((p), (v), __ATOMIC_SEQ_CST)

// ---- runtime API (synchronous; one "device") ----
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNoDevice = 100, hipErrorNotReady = 600 };
typedef struct hipemuStream_ *hipStream_t;
typedef struct hipemuEvent_ *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipStreamDefault = 0, hipStreamNonBlocking = 1, hipEventDefault = 0, hipEventDisableTiming = 2, hipHostMallocDefault = 0 };
struct hipDeviceProp_t { char name[256]; int multiProcessorCount; size_t totalGlobalMem; int warpSize; char gcnArchName[256]; };

hipError_t hipGetDeviceCount(int *n);
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int *d);
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int d);
hipError_t hipGetLastError();
const char *hipGetErrorString(hipError_t e);
hipError_t hipDeviceSynchronize();
hipError_t hipMallocRaw(void **p, size_t n);
hipError_t hipFree(void *p);
hipError_t hipHostMallocRaw(void **p, size_t n, unsigned flags);
hipError_t hipHostFree(void *p);
template <class T> static inline hipError_t hipMalloc(T **p, size_t n) { return hipMallocRaw((void **)p, n); }
template <class T> static inline hipError_t hipHostMalloc(T **p, size_t n, unsigned flags = 0) { return hipHostMallocRaw((void **)p, n, flags); }
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind k, hipStream_t st = nullptr);
hipError_t hipMemset(void *d, int v, size_t n);
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t st = nullptr);
hipError_t hipStreamCreate(hipStream_t *s);
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags = 0);
hipError_t hipEventCreate(hipEvent_t *e);
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s = nullptr);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventQuery(hipEvent_t e);
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b);
template <class K> static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *n, K, int, size_t) { *n = 1; return hipSuccess; }
