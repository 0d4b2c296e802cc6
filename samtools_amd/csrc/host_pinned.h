// host_pinned.h -- page-locked host memory for the staging pools and the fetched text (BASELINE.json north_star: "stages
// pre-decoded BAM records into pinned ... buffers").  hipMemcpyAsync from pageable memory goes through the runtime's own
// bounce buffer (~10 GB/s and synchronous); from pinned memory it is a direct DMA at PCIe rate.  The vectors are reused
// across windows, so the (slow) hipHostMalloc calls happen only while the pools grow.  Without a usable device (host-only
// tools and tests: sta_io_scan) the allocator falls back to malloc.
#pragma once
#include <cstddef>
#include <cstdint>
#include <new>
#include <vector>

namespace sta {

void *pinned_alloc(size_t bytes);      // never returns nullptr (throws std::bad_alloc)
// Page-locking costs ~0.3 ms per MB (a window's staging pools: ~15 ms per pipeline slot) and needs the HIP runtime, which takes the
// first ~0.2 s of a process to come up.  The drivers therefore say when the runtime exists (until then an allocation is plain malloc
// and nobody waits for the runtime: the producer stages its first windows while the engine is still being created), and whether the
// input is large enough for page-locked staging pools to pay at all (round 6; profiles/r06_sessionG_e2e_window_trace.log).
void pinned_runtime_is_up();
void pinned_set_policy(bool page_lock);     // false: every pinned_alloc is plain malloc (small inputs); default true
bool pinned_policy();
void pinned_free(void *p) noexcept;

template <class T> struct PinnedAlloc {
    typedef T value_type;
    PinnedAlloc() noexcept {}
    template <class U> PinnedAlloc(const PinnedAlloc<U> &) noexcept {}
    T *allocate(size_t n) { return static_cast<T *>(pinned_alloc(n * sizeof(T))); }
    void deallocate(T *p, size_t) noexcept { pinned_free(p); }
    // resize(n) leaves new elements of these plain-data vectors uninitialised (they are overwritten right away: bulk copies,
    // device-to-host fetches) instead of zero-filling them first; resize(n, v) still fills
    template <class U> void construct(U *p) noexcept { ::new (static_cast<void *>(p)) U; }
    template <class U, class A0, class... A> void construct(U *p, A0 &&a0, A &&... a) { ::new (static_cast<void *>(p)) U(static_cast<A0 &&>(a0), static_cast<A &&>(a)...); }
    template <class U> bool operator==(const PinnedAlloc<U> &) const noexcept { return true; }
    template <class U> bool operator!=(const PinnedAlloc<U> &) const noexcept { return false; }
};
template <class T> using pvector = std::vector<T, PinnedAlloc<T>>;

}  // namespace sta
