// tests/cpu/pinned_stub.cpp -- TEST INFRASTRUCTURE: host_pinned.h without a HIP runtime (plain malloc), for the host-only harnesses
// that compile host_io.cpp on its own (its group buffers are page-locked vectors in the product).
#include "../../samtools_amd/csrc/host_pinned.h"
#include <cstdlib>
#include <new>
namespace sta {
void *pinned_alloc(size_t bytes) { void *p = malloc(bytes ? bytes : 1); if (!p) throw std::bad_alloc(); return p; }
void pinned_free(void *p) noexcept { free(p); }
}
