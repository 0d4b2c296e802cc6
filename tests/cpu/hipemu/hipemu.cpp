// hipemu.cpp -- TEST INFRASTRUCTURE ONLY (see hipemu.h): the fiber scheduler that stands in for a workgroup, and a synchronous
// stand-in for the few HIP runtime calls the library makes.
#include "hipemu.h"
#include <sys/mman.h>
#include <time.h>
#include <sched.h>
#include <vector>
#include <algorithm>
#include <mutex>
#include <string>
#include <unordered_map>
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#include <pthread.h>
#include <dlfcn.h>

#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define HIPEMU_ASAN 1
extern "C" void __sanitizer_start_switch_fiber(void **fake_stack_save, const void *bottom, size_t size);
extern "C" void __sanitizer_finish_switch_fiber(void *fake_stack_save, const void **bottom_old, size_t *size_old);
extern "C" void __asan_unpoison_memory_region(void const volatile *addr, size_t size);
#endif
#endif

// dynamic LDS: the kernels declare these as `extern __shared__ T name[]` (thread_local here, one OS thread runs one workgroup at a time)
#define HIPEMU_LDS_BYTES (160 * 1024)
alignas(16) thread_local uint8_t baq_state[HIPEMU_LDS_BYTES];
alignas(16) thread_local unsigned char lds[HIPEMU_LDS_BYTES];
alignas(16) thread_local char lds_dtext[HIPEMU_LDS_BYTES];
alignas(16) thread_local uint8_t tile[HIPEMU_LDS_BYTES];
alignas(16) thread_local char lds_text[HIPEMU_LDS_BYTES];

// void hipemu_switch(void **save_sp, void *new_sp): callee-saved registers on the old stack, stack pointers exchanged
extern "C" void hipemu_switch(void **save_sp, void *new_sp);
asm(R"(
    .text
    .globl hipemu_switch
    .type hipemu_switch, @function
hipemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size hipemu_switch, .-hipemu_switch
)");

thread_local hipError_t tl_last = hipSuccess;       // what hipGetLastError() hands out next

namespace hipemu {

enum { S_READY = 0, S_WAVE, S_BLOCK, S_DONE };

thread_local Lane *tl_lane = nullptr;
thread_local Block tl_block;
// where the running launch's kernel lies in the control-flow table, and the running lane: what observe() needs, in one thread-local
// object (one TLS look-up per traced block in the TRACE=1 flavour)
struct TraceCtx { Lane *lane; uintptr_t kfn_lo, kfn_span; const int32_t *addr2blk; const void *blks; const int *loop_pool; };
thread_local TraceCtx tl_trace = { nullptr, 1, 0, nullptr, nullptr, nullptr };

namespace {

// ---- the control-flow table (mkloops.py): per function its basic blocks (offsets into the library) with their number in the
// loop-contiguous reverse post-order and the loops around them ----
struct LoopTab {
    struct Fn { uintptr_t lo, hi; int first_block, n_blocks; };
    struct Blk { uintptr_t start; int rpo; int first_loop, n_loops; int header; };
    std::vector<Fn> fns;
    std::vector<Blk> blks;
    std::vector<int> loop_pool;          // the loops around a block, outermost first
    const Fn *fn_of(uintptr_t a) const
    {
        auto it = std::upper_bound(fns.begin(), fns.end(), a, [](uintptr_t v, const Fn &f) { return v < f.lo; });
        if (it == fns.begin()) return nullptr;
        --it;
        return a <= it->hi ? &*it : nullptr;
    }
};

struct Sched {
    void *sp = nullptr;
    const Launch *launch = nullptr;
    std::vector<Lane> lanes;
    std::vector<char *> stacks;
    size_t stack_bytes = 0;
    bool spun = false;
    const LoopTab::Fn *kfn = nullptr; uintptr_t kfn_lo = 1, kfn_hi = 0;          // the launched kernel's function in the loop table
    // code address - kfn_lo -> 2 * index of its basic block + (the block is a loop header), per kernel, built at its first launch on this thread
    std::unordered_map<const LoopTab::Fn *, std::vector<int32_t>> addr2blk_of;
#ifdef HIPEMU_ASAN
    const void *sched_bottom = nullptr; size_t sched_size = 0;
#endif
};
thread_local Sched tl_s;

long env_long(const char *k, long d) { const char *v = getenv(k); return v && *v ? atol(v) : d; }
const bool g_trace = env_long("HIPEMU_TRACE", 0) != 0;
const bool g_poison_lds = env_long("HIPEMU_POISON", 1) != 0;
const bool g_diverge_log = env_long("HIPEMU_LOG_DIVERGENCE", 0) != 0;

// a code address as an offset into the library (what llvm-symbolizer -e <library> wants)
const uintptr_t g_base = [] { Dl_info di; return dladdr((const void *)&env_long, &di) && di.dli_fbase ? (uintptr_t)di.dli_fbase : (uintptr_t)0; }();
uintptr_t off(const void *p) { return (uintptr_t)p - g_base; }

[[noreturn]] void die(const char *what)
{
    fprintf(stderr, "hipemu: %s (kernel %s, workgroup %u)\n", what, tl_s.launch ? tl_s.launch->name : "?", tl_block.bid.x);
    abort();
}

inline void to_sched(Lane *me)
{
    { const uintptr_t here = (uintptr_t)__builtin_frame_address(0); if (here < me->low_sp) me->low_sp = here; }
#ifdef HIPEMU_ASAN
    void *fake = nullptr;
    __sanitizer_start_switch_fiber(me->state == S_DONE ? nullptr : &fake, tl_s.sched_bottom, tl_s.sched_size);
    hipemu_switch(&me->sp, tl_s.sp);
    __sanitizer_finish_switch_fiber(fake, &tl_s.sched_bottom, &tl_s.sched_size);
#else
    hipemu_switch(&me->sp, tl_s.sp);
#endif
}

void fiber_entry()
{
#ifdef HIPEMU_ASAN
    __sanitizer_finish_switch_fiber(nullptr, &tl_s.sched_bottom, &tl_s.sched_size);
#endif
    Lane *me = tl_lane;
    tl_s.launch->invoke(tl_s.launch->closure);
    me = tl_lane;
    me->state = S_DONE;
    to_sched(me);
    die("a finished lane was resumed");
}

inline void run_lane(Lane *l)
{
    tl_lane = l; tl_trace.lane = tl_trace.addr2blk ? l : nullptr;
#ifdef HIPEMU_ASAN
    void *fake = nullptr;
    __sanitizer_start_switch_fiber(&fake, l->stack, tl_s.stack_bytes);
    hipemu_switch(&tl_s.sp, l->sp);
    __sanitizer_finish_switch_fiber(fake, nullptr, nullptr);
#else
    hipemu_switch(&tl_s.sp, l->sp);
#endif
}

void prepare_lane(Lane *l, char *stack, size_t bytes)
{
    uintptr_t top = ((uintptr_t)stack + bytes) & ~(uintptr_t)15;
#ifdef HIPEMU_ASAN
    // the stack is used again: whatever red zones its last fiber's frames left behind are stale
    if (l->stack == stack && l->low_sp >= (uintptr_t)stack && l->low_sp < top) {
        uintptr_t lo = l->low_sp > (uintptr_t)stack + 8192 ? l->low_sp - 8192 : (uintptr_t)stack;
        __asan_unpoison_memory_region((void *)lo, top - lo);
    } else __asan_unpoison_memory_region(stack, bytes);
#endif
    l->stack = stack; l->low_sp = top;
    void **sp = (void **)top;
    *--sp = nullptr;                       // the "return address" of fiber_entry (never used)
    *--sp = (void *)&fiber_entry;          // popped by hipemu_switch's ret
    for (int i = 0; i < 6; ++i) *--sp = nullptr;
    l->sp = sp;
    l->state = S_READY;
    l->n_loops = 0; l->kpos = 0; l->rpo = 0; l->n_stack = 0; l->blk = nullptr;
}

// ---- the loop table, loaded ----
const LoopTab &loop_tab()
{
    static LoopTab T;
    static std::once_flag once;
    std::call_once(once, [] {
        Dl_info di;
        if (!dladdr((const void *)&loop_tab, &di) || !di.dli_fname) return;
        const std::string path = std::string(di.dli_fname) + ".loops";
        FILE *f = fopen(path.c_str(), "r");
        if (!f) { fprintf(stderr, "hipemu: %s is missing (tests/cpu/hipemu/mkloops.py writes it): diverged waves are ordered by code address alone\n", path.c_str()); return; }
        char line[8192];
        while (fgets(line, sizeof(line), f)) {
            unsigned long lo = 0, hi = 0;
            if (line[0] == 'F' && sscanf(line + 1, "%lx %lx", &lo, &hi) == 2) { LoopTab::Fn fn = { lo, hi, (int)T.blks.size(), 0 }; T.fns.push_back(fn); }
            else if (line[0] == 'B' && !T.fns.empty()) {
                char *q = line + 1;
                LoopTab::Blk b; b.start = strtoul(q, &q, 16); b.rpo = (int)strtol(q, &q, 10); b.header = (int)strtol(q, &q, 10); b.first_loop = (int)T.loop_pool.size(); b.n_loops = 0;
                for (;;) { while (*q == ' ') ++q; if (*q < '0' || *q > '9') break; T.loop_pool.push_back((int)strtol(q, &q, 10)); b.n_loops++; }
                T.blks.push_back(b); T.fns.back().n_blocks++;
            }
        }
        fclose(f);
    });
    return T;
}

// The lane is at code address `o` of the kernel's function (a stop at a wave operation or a barrier, or -- kernels are built with
// -fsanitize-coverage=bb,trace-pc -- the entry of any basic block).  Because every block entry is seen, a loop's back edge is seen when
// it is taken: the lane arrives, inside a loop it was already in, at a block that is not later in reverse post-order than the one it
// came from; it is the innermost such loop that went round (going round an outer one passes blocks outside the inner one first).
inline void observe(Lane *me, uintptr_t o)
{
    const TraceCtx &C = tl_trace;
    const LoopTab::Blk *b = (const LoopTab::Blk *)C.blks + (C.addr2blk[o - C.kfn_lo] >> 1);
    if (b == (const LoopTab::Blk *)me->blk) {          // the same block again: further down it, or once round a loop that is this block
        if (o <= me->kpos && me->n_loops > 0) me->loops[me->n_loops - 1].count++;
        me->kpos = o;
        return;
    }
    const int *ids = C.loop_pool + b->first_loop;
    const int nb = b->n_loops < 12 ? b->n_loops : 12;
    int common = 0;
    while (common < me->n_loops && common < nb && me->loops[common].id == ids[common]) ++common;
    if (common > 0 && b->rpo <= me->rpo) me->loops[common - 1].count++;
    for (int i = common; i < nb; ++i) { me->loops[i].id = ids[i]; me->loops[i].count = 0; }
    me->n_loops = nb; me->rpo = b->rpo; me->kpos = o; me->blk = b;
}

// the lane has stopped at a wave operation or a barrier
void note_position(Lane *me, const void *const *ra_outer_first, int n)
{
    Sched &S = tl_s;
    uintptr_t kpos = 0;
    for (int i = 0; i < n; ++i) { const uintptr_t o = off(ra_outer_first[i]); if (o >= S.kfn_lo && o <= S.kfn_hi) { kpos = o; break; } }
    if (!S.kfn || !kpos) { me->kpos = n ? off(ra_outer_first[n - 1]) : 0; me->rpo = 0; me->n_loops = 0; return; }
    observe(me, kpos);
}

// one wave: serve the group of lanes that wait at the lowest site
void resolve_wave(Lane *w, int cnt)
{
    // Which group first, when lanes of the wave wait at different sites?  The hardware runs an inner divergent region to its end while
    // the lanes that skipped it stay masked where control flow joins again.  The emulator gets the same order from the control-flow graph
    // of the code it runs (mkloops.py: basic blocks, natural loops, reverse post-order without the back edges): a lane is earlier when it
    // has gone round a loop both are in fewer times; in the same iteration, when its block comes first in reverse post-order (a block
    // that can be reached from another one without a back edge has the higher number: the lanes at the other one are on their way to it).
    auto earlier = [](const Lane &a, const Lane &b) {
        for (int i = 0; i < a.n_loops && i < b.n_loops && a.loops[i].id == b.loops[i].id; ++i)
            if (a.loops[i].count != b.loops[i].count) return a.loops[i].count < b.loops[i].count;
        if (a.rpo != b.rpo) return a.rpo < b.rpo;
        if (a.kpos != b.kpos) return a.kpos < b.kpos;
        const int n = a.n_stack < b.n_stack ? a.n_stack : b.n_stack;
        for (int k = 0; k < n; ++k) if (a.stack_sites[k] != b.stack_sites[k]) return (uintptr_t)a.stack_sites[k] < (uintptr_t)b.stack_sites[k];
        return a.n_stack > b.n_stack;
    };
    const Lane *pick = nullptr; int n_wait = 0; bool mixed = false;
    for (int i = 0; i < cnt; ++i) if (w[i].state == S_WAVE) {
        ++n_wait;
        if (!pick) { pick = &w[i]; continue; }
        if (w[i].site == pick->site && w[i].n_stack == pick->n_stack && !memcmp(w[i].stack_sites, pick->stack_sites, sizeof(void *) * (size_t)pick->n_stack)) continue;
        mixed = true;
        if (earlier(w[i], *pick)) pick = &w[i];
    }
    if (!n_wait) return;
    const void *site = pick->site;
    if (mixed && g_diverge_log) {
        fprintf(stderr, "hipemu: divergent wave operations in %s:", tl_s.launch->name);
        for (int i = 0; i < cnt; ++i) if (w[i].state == S_WAVE && (i == 0 || w[i].site != w[i - 1].site || w[i - 1].state != S_WAVE)) {
            fprintf(stderr, " lane %d.. 0x%zx/%d rpo %d [", i, off(w[i].site), w[i].kind, w[i].rpo);
            for (int k = 0; k < w[i].n_loops; ++k) fprintf(stderr, "%s%d:%d", k ? " " : "", w[i].loops[k].id, w[i].loops[k].count);
            fprintf(stderr, "]");
        }
        fprintf(stderr, " -> 0x%zx\n", off(site));
    }
    uint64_t members = 0; int kind = 0;
    for (int i = 0; i < cnt; ++i)
        if (w[i].state == S_WAVE && w[i].site == site && w[i].n_stack == pick->n_stack && !memcmp(w[i].stack_sites, pick->stack_sites, sizeof(void *) * (size_t)pick->n_stack)) { members |= 1ull << i; kind = w[i].kind; }
    const int first = __builtin_ctzll(members);
    uint64_t ballot = 0;
    if (kind == K_BALLOT) for (int i = 0; i < cnt; ++i) if ((members >> i & 1) && w[i].val) ballot |= 1ull << i;
    uint64_t res[64];
    for (int i = 0; i < cnt; ++i) {
        if (!(members >> i & 1)) continue;
        Lane &l = w[i];
        if (l.kind != kind) die("lanes of one wave meet at one site with different operations");
        const int width = l.width > 0 && l.width <= 64 ? l.width : 64;
        const int seg = i & ~(width - 1);
        int src = i; bool own = false;
        switch (kind) {
        case K_BALLOT: res[i] = ballot; continue;
        case K_SYNC: res[i] = 0; continue;
        case K_SHFL: src = seg + (l.arg & (width - 1)); break;
        case K_SHFL_UP: src = i - l.arg; if (src < seg) own = true; break;
        case K_SHFL_DOWN: src = i + l.arg; if (src >= seg + width) own = true; break;
        case K_SHFL_XOR: src = i ^ l.arg; if (src >= seg + width || src < seg) own = true; break;
        case K_READLANE: src = l.arg & 63; break;
        case K_READFIRST: src = first; break;
        default: die("unknown wave operation");
        }
        if (own) res[i] = l.val;
        else if (src >= 0 && src < cnt && (members >> src & 1)) res[i] = w[src].val;
        else res[i] = 0;           // an inactive source lane: the hardware returns whatever that register holds; nothing may depend on it
    }
    for (int i = 0; i < cnt; ++i) if (members >> i & 1) { w[i].res = res[i]; w[i].state = S_READY; }
}

void run_block(const Launch &L, unsigned bx, unsigned by, unsigned bz)
{
    Sched &S = tl_s;
    const int n = (int)(L.block.x * L.block.y * L.block.z);
    tl_block.bid = dim3(bx, by, bz); tl_block.bdim = L.block; tl_block.gdim = L.grid;
    if (g_poison_lds && L.shmem) {          // LDS holds whatever the last workgroup left there: nothing may depend on it
        memset(baq_state, 0x5A, L.shmem); memset(lds, 0x5A, L.shmem); memset(lds_dtext, 0x5A, L.shmem); memset(tile, 0x5A, L.shmem); memset(lds_text, 0x5A, L.shmem);
    }
    if ((int)S.lanes.size() < n) S.lanes.resize((size_t)n);
    while ((int)S.stacks.size() < n) {
        char *p = (char *)mmap(nullptr, S.stack_bytes + 4096, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (p == MAP_FAILED) die("cannot map a fiber stack");
        mprotect(p, 4096, PROT_NONE);          // guard page under the stack
        S.stacks.push_back(p + 4096);
    }
    for (int i = 0; i < n; ++i) {
        Lane &l = S.lanes[(size_t)i];
        l.lin = i;
        l.tid = dim3((unsigned)i % L.block.x, ((unsigned)i / L.block.x) % L.block.y, (unsigned)i / (L.block.x * L.block.y));
        prepare_lane(&l, S.stacks[(size_t)i], S.stack_bytes);
    }
    const int nw = (n + 63) / 64;
    long idle_passes = 0;
    for (;;) {
        bool changed = false; int live = 0, at_barrier = 0, spinning = 0;
        for (int w = 0; w < nw; ++w) {
            Lane *wl = &S.lanes[(size_t)w * 64]; const int cnt = n - w * 64 < 64 ? n - w * 64 : 64;
            for (;;) {
                bool spun_any = false, ran = false;
                for (int i = 0; i < cnt; ++i) if (wl[i].state == S_READY) {
                    S.spun = false;
                    run_lane(&wl[i]);
                    if (S.spun) spun_any = true; else { ran = true; changed = true; }
                }
                if (ran) continue;                 // states moved: look again (a lane that came back READY without spinning does not exist)
                if (spun_any) break;               // a spinning lane holds its wave: the others' wave operations wait for it (EXEC)
                bool waiting = false;
                for (int i = 0; i < cnt; ++i) if (wl[i].state == S_WAVE) { waiting = true; break; }
                if (!waiting) break;
                resolve_wave(wl, cnt); changed = true;
            }
            for (int i = 0; i < cnt; ++i) { const int st = wl[i].state; if (st != S_DONE) ++live; if (st == S_BLOCK) ++at_barrier; if (st == S_READY) ++spinning; }
        }
        if (!live) break;
        if (at_barrier == live) { for (int i = 0; i < n; ++i) if (S.lanes[(size_t)i].state == S_BLOCK) S.lanes[(size_t)i].state = S_READY; continue; }
        if (!spinning) die("deadlock: lanes wait at a barrier that the rest of the workgroup cannot reach");
        if (changed) idle_passes = 0;
        else { sched_yield(); if (++idle_passes > 2000000) die("deadlock: a lane spins on something no earlier workgroup will write"); }
    }
}

// HIPEMU_WATCHDOG=<seconds>: when a process is still running after that long, every thread that is inside an emulated kernel says
// where (a hang inside a kernel cannot be looked at with a debugger in this container)
pthread_mutex_t g_reg_mu = PTHREAD_MUTEX_INITIALIZER;
std::vector<pthread_t> g_in_kernel;
void on_probe(int)
{
    void *bt[48]; const int n = backtrace(bt, 48); backtrace_symbols_fd(bt, n, 2);
    Sched &S = tl_s;
    if (S.launch) {
        fprintf(stderr, "hipemu: watchdog: kernel %s, workgroup %u of %u, running lane %d\n", S.launch->name, tl_block.bid.x, S.launch->grid.x, tl_lane ? tl_lane->lin : -1);
        const int nl = (int)(S.launch->block.x * S.launch->block.y * S.launch->block.z);
        for (int i = 0; i < nl; ++i) if (S.lanes[(size_t)i].state != S_DONE) fprintf(stderr, "  lane %d state %d site 0x%zx kind %d\n", i, S.lanes[(size_t)i].state, off(S.lanes[(size_t)i].site), S.lanes[(size_t)i].kind);
    }
    sleep(1);
    _exit(97);
}
void *watchdog(void *arg)
{
    sleep((unsigned)(uintptr_t)arg);
    fprintf(stderr, "hipemu: watchdog fired\n");
    pthread_mutex_lock(&g_reg_mu);
    if (g_in_kernel.empty()) { fprintf(stderr, "hipemu: no thread is inside a kernel\n"); _exit(98); }
    for (pthread_t t : g_in_kernel) pthread_kill(t, SIGUSR2);
    pthread_mutex_unlock(&g_reg_mu);
    sleep(5); _exit(97);
}
const long g_watchdog = [] {
    const long t = env_long("HIPEMU_WATCHDOG", 0);
    if (t > 0) { signal(SIGUSR2, on_probe); pthread_t th; pthread_create(&th, nullptr, watchdog, (void *)(uintptr_t)t); pthread_detach(th); }
    return t;
}();
void reg_enter() { if (!g_watchdog) return; pthread_mutex_lock(&g_reg_mu); g_in_kernel.push_back(pthread_self()); pthread_mutex_unlock(&g_reg_mu); }
void reg_leave()
{
    if (!g_watchdog) return;
    pthread_mutex_lock(&g_reg_mu);
    for (size_t i = 0; i < g_in_kernel.size(); ++i) if (pthread_equal(g_in_kernel[i], pthread_self())) { g_in_kernel.erase(g_in_kernel.begin() + (long)i); break; }
    pthread_mutex_unlock(&g_reg_mu);
}

}  // namespace

uint64_t wave_op(int kind, uint64_t val, int arg, int width)
{
    Lane *me = tl_lane;
    me->site = __builtin_return_address(0); me->kind = kind; me->val = val; me->arg = arg; me->width = width;
    {   // the frame chain (the build keeps frame pointers; a fiber starts with a null frame pointer)
        const void *ra[24]; int n = 0;
        void **fp = (void **)__builtin_frame_address(0);
        while (fp && n < 24) { ra[n++] = fp[1]; fp = (void **)fp[0]; }
        if (n > 0 && ra[n - 1] == nullptr) --n;
        me->n_stack = n;
        for (int i = 0; i < n; ++i) me->stack_sites[i] = ra[n - 1 - i];
        note_position(me, me->stack_sites, n);
    }
    me->state = S_WAVE;
    to_sched(me);
    return me->res;
}

void block_barrier()
{
    Lane *me = tl_lane;
    {   // a barrier is a position in the loop nest like any other (the more positions are seen, the fewer back edges go unnoticed)
        const void *ra[24], *fw[24]; int n = 0;
        void **fp = (void **)__builtin_frame_address(0);
        while (fp && n < 24) { ra[n++] = fp[1]; fp = (void **)fp[0]; }
        if (n > 0 && ra[n - 1] == nullptr) --n;
        for (int i = 0; i < n; ++i) fw[i] = ra[n - 1 - i];
        note_position(me, fw, n);
    }
    me->state = S_BLOCK;
    to_sched(me);
}

void lane_yield()
{
    Lane *me = tl_lane;
    tl_s.spun = true;
    to_sched(me);
}

void run(const Launch &L)
{
    Sched &S = tl_s;
    if (S.launch) die("a kernel launched from inside a kernel");
    if (!S.stack_bytes) S.stack_bytes = (size_t)env_long("HIPEMU_STACK_KB", 256) * 1024;
    const size_t n = (size_t)L.block.x * L.block.y * L.block.z;
    if (!n || n > 1024) die("workgroup size out of range");
    if (g_trace) fprintf(stderr, "hipemu: %s grid %u x %u x %u, workgroup %u, lds %zu\n", L.name, L.grid.x, L.grid.y, L.grid.z, L.block.x, L.shmem);
    // what the HIP runtime refuses with hipErrorInvalidConfiguration: an empty grid (the LDS limit is checked below against the CU's 160 KB).
    // The device build returns the error from hipGetLastError() and runs nothing; so does this one (found on the device by hunt5 / hunt6,
    // round 6: a consensus window whose reads hold no base launched a grid of 0 workgroups -- the emulation had run "nothing" silently).
    if ((size_t)L.grid.x * L.grid.y * L.grid.z == 0) {
        if (const char *lf = getenv("HIPEMU_STRICT_FILE")) { if (FILE *fh = fopen(lf, "a")) { fprintf(fh, "%s grid %u x %u x %u\n", L.name, L.grid.x, L.grid.y, L.grid.z); fclose(fh); } }
        if (g_trace || getenv("HIPEMU_STRICT")) fprintf(stderr, "hipemu: %s: invalid configuration (grid %u x %u x %u, lds %zu)\n", L.name, L.grid.x, L.grid.y, L.grid.z, L.shmem);
        ::tl_last = (hipError_t)9;       // hipErrorInvalidConfiguration
        return;
    }
    if (L.shmem > HIPEMU_LDS_BYTES) die("more dynamic LDS than a CU has");
    S.launch = &L; reg_enter();
    {
        const uintptr_t k = off(L.kernel);
        const LoopTab::Fn *fn = loop_tab().fn_of(k);
        S.kfn = fn; S.kfn_lo = fn ? fn->lo : 1; S.kfn_hi = fn ? fn->hi : 0;
        if (fn) {
            std::vector<int32_t> &v = S.addr2blk_of[fn];
            if (v.empty()) {
                v.resize(fn->hi - fn->lo + 1);
                const LoopTab &T = loop_tab();
                for (int k = 0; k < fn->n_blocks; ++k) {
                    const uintptr_t a0 = T.blks[(size_t)(fn->first_block + k)].start, a1 = k + 1 < fn->n_blocks ? T.blks[(size_t)(fn->first_block + k + 1)].start : fn->hi + 1;
                    for (uintptr_t a = a0; a < a1; ++a) v[a - fn->lo] = (fn->first_block + k) * 2 + (T.blks[(size_t)(fn->first_block + k)].header ? 1 : 0);
                }
            }
            const LoopTab &T = loop_tab();
            tl_trace.kfn_lo = fn->lo; tl_trace.kfn_span = fn->hi - fn->lo; tl_trace.addr2blk = v.data(); tl_trace.blks = T.blks.data(); tl_trace.loop_pool = T.loop_pool.data();
        } else { tl_trace.kfn_lo = 1; tl_trace.kfn_span = 0; tl_trace.addr2blk = nullptr; }
    }
    Lane *outer_lane = tl_lane; Block outer_block = tl_block;
    for (unsigned z = 0; z < L.grid.z; ++z) for (unsigned y = 0; y < L.grid.y; ++y) for (unsigned x = 0; x < L.grid.x; ++x) run_block(L, x, y, z);
    tl_lane = outer_lane; tl_block = outer_block; tl_trace.lane = nullptr;
    S.launch = nullptr; reg_leave();
}

}  // namespace hipemu

// every basic block of the kernels' translation units reports here (-fsanitize-coverage=bb,trace-pc); host code of those units, and
// device functions that were not inlined into their kernel, are not tracked
extern "C" void __sanitizer_cov_trace_pc()
{
    const hipemu::TraceCtx &C = hipemu::tl_trace;
    if (!C.lane) return;
    const uintptr_t d = (uintptr_t)__builtin_return_address(0) - hipemu::g_base - C.kfn_lo;
    if (d > C.kfn_span) return;
    // only the entries of loop headers matter: that is where a back edge lands (a block that is not a header changes nothing that the
    // lane's next stop does not see for itself)
    if (!(C.addr2blk[d] & 1)) return;
    hipemu::observe(C.lane, d + C.kfn_lo);
}

// ---- runtime API ----
namespace {
const bool g_poison = hipemu::env_long("HIPEMU_POISON", 1) != 0;
struct Ev { double t; };
double now_ms() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
}

hipError_t hipGetDeviceCount(int *n) { if (getenv("HIPEMU_NO_DEVICE")) { *n = 0; return hipErrorNoDevice; } *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidValue; }
hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int)
{
    memset(p, 0, sizeof(*p));
    snprintf(p->name, sizeof(p->name), "hipemu (CPU emulation, test infrastructure)"); snprintf(p->gcnArchName, sizeof(p->gcnArchName), "hipemu");
    p->multiProcessorCount = (int)hipemu::env_long("HIPEMU_CUS", 4); p->totalGlobalMem = (size_t)8 << 30; p->warpSize = 64;
    return hipSuccess;
}
hipError_t hipGetLastError() { hipError_t e = tl_last; tl_last = hipSuccess; return e; }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : e == hipErrorOutOfMemory ? "out of memory" : (int)e == 9 ? "invalid configuration argument" : "hipemu error"; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipMallocRaw(void **p, size_t n)
{
    void *q = nullptr;
    if (posix_memalign(&q, 256, n ? n : 1)) { *p = nullptr; return tl_last = hipErrorOutOfMemory; }
    if (g_poison) memset(q, 0xA5, n);      // device memory is not zeroed: nothing may depend on what a fresh allocation holds
    *p = q; return hipSuccess;
}
hipError_t hipFree(void *p) { free(p); return hipSuccess; }
hipError_t hipHostMallocRaw(void **p, size_t n, unsigned)
{
    void *q = nullptr;
    if (posix_memalign(&q, 4096, n ? n : 1)) { *p = nullptr; return tl_last = hipErrorOutOfMemory; }
    if (g_poison) memset(q, 0xA5, n);
    *p = q; return hipSuccess;
}
hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { if (n) memmove(d, s, n); return hipSuccess; }
hipError_t hipMemset(void *d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { if (n) memset(d, v, n); return hipSuccess; }
hipError_t hipStreamCreate(hipStream_t *s) { *s = (hipStream_t)malloc(8); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (hipStream_t)malloc(8); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) { Ev *v = (Ev *)malloc(sizeof(Ev)); v->t = 0; *e = (hipEvent_t)v; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { ((Ev *)e)->t = now_ms(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(((Ev *)b)->t - ((Ev *)a)->t); return hipSuccess; }
