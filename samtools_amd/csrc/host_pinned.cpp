// host_pinned.cpp -- see host_pinned.h
#include "host_pinned.h"
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdlib>
#include <new>

namespace sta {

namespace {
constexpr size_t HDR = 64;                  // keeps the payload 64-byte aligned; first word = how the block was obtained
constexpr uint64_t KIND_PINNED = 0x50494e4e45440001ull, KIND_MALLOC = 0x4d414c4c4f430001ull;
constexpr size_t PIN_MIN = 1 << 16;         // small blocks are not worth a page-locking call

std::atomic<int> g_runtime_up{0}, g_policy{1};

bool device_present()
{
    static const bool yes = [] { int n = 0; bool ok = hipGetDeviceCount(&n) == hipSuccess && n > 0; (void)hipGetLastError(); return ok && !getenv("STA_NO_PINNED"); }();
    return yes;
}
}  // namespace

void *pinned_alloc(size_t bytes)
{
    void *raw = nullptr;
    uint64_t kind = KIND_MALLOC;
    if (bytes >= PIN_MIN && g_policy.load(std::memory_order_relaxed) && g_runtime_up.load(std::memory_order_acquire) && device_present()) {
        if (hipHostMalloc(&raw, bytes + HDR, hipHostMallocDefault) == hipSuccess && raw) kind = KIND_PINNED;
        else { raw = nullptr; (void)hipGetLastError(); }
    }
    if (!raw) { raw = malloc(bytes + HDR); if (!raw) throw std::bad_alloc(); }
    *static_cast<uint64_t *>(raw) = kind;
    return static_cast<char *>(raw) + HDR;
}

void pinned_runtime_is_up() { g_runtime_up.store(1, std::memory_order_release); }
void pinned_set_policy(bool page_lock) { const char *e = getenv("STA_PIN"); g_policy.store(e ? (atoi(e) != 0) : (page_lock ? 1 : 0), std::memory_order_relaxed); }
bool pinned_policy() { return g_policy.load(std::memory_order_relaxed) != 0; }

void pinned_free(void *p) noexcept
{
    if (!p) return;
    void *raw = static_cast<char *>(p) - HDR;
    if (*static_cast<uint64_t *>(raw) == KIND_PINNED) (void)hipHostFree(raw); else free(raw);
}

}  // namespace sta
