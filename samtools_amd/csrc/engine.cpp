// engine.cpp -- C-ABI engine: HBM staging, workspace, kernel pipeline orchestration.
// Implements include/samtools_amd.h (the bulk replacement for the bam_mplp_* / add_depth hot
// loops; see that header for the reference file:line map).  No CPU compute fallback exists here:
// without a usable HIP device every entry point returns STA_ERR_NO_DEVICE.
#include "sta_dev.h"
#include "glf_tables.h"
#include "cons_host.h"
#include "cons_window.h"
#include <memory>
#include <algorithm>
#include <string>
#include <vector>
#include <map>
#include <cstring>
#include <cstdio>
#include <climits>
#include <cctype>

namespace {

struct DevBuf {
    void *p = nullptr; size_t cap = 0;
    int ensure(size_t n)
    {
        if (n <= cap) return 0;
        if (p) { hipFree(p); p = nullptr; cap = 0; }
        size_t want = n + (n >> 3) + 256;
        if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; return -1; }
        cap = want;
        return 0;
    }
    void release() { if (p) hipFree(p); p = nullptr; cap = 0; }
};

struct FileBufs {
    // staged input copies (only used with STA_MEM_HOST)
    DevBuf pos, flag, mapq, aux, lq, cig_off, base_off8, mtid, mpos, isize, name_off, cigar, seq, qual, bq, names, xoff, xtext, moff, mqpos, mtoff, mtext;
    // device staging (sta_reads.raw_*): the uploaded BAM bytes, the records' offsets in them, the pools built for a verify run
    DevBuf raw, raw_off, raw_vfy;
    // workspace
    DevBuf qual_work, end, maxend, info, clip, chain, fix_y, fix_mate, fix_q, slist, clip_in, mate;
    void release()
    {
        DevBuf *all[] = { &pos, &flag, &mapq, &aux, &lq, &cig_off, &base_off8, &mtid, &mpos, &isize, &name_off, &cigar,
                          &seq, &qual, &bq, &names, &xoff, &xtext, &moff, &mqpos, &mtoff, &mtext, &raw, &raw_off, &raw_vfy, &qual_work, &end, &maxend, &info, &clip, &chain, &fix_y, &fix_mate, &fix_q, &slist, &clip_in, &mate };
        for (DevBuf *b : all) b->release();
    }
};

struct RefSeq { DevBuf buf; int64_t len = 0; bool external = false; const char *ext = nullptr; };

struct ProfEntry { uint64_t launches = 0; double ms = 0; };
struct ProfPending { std::string name; hipEvent_t a, b; };

}  // namespace

enum { PIN_FILES = 1024, PIN_BYTES = 16384 };

struct sta_engine {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t side = nullptr, side2 = nullptr;   // side streams: the list's BAQ groups (band width 8 / band width 7) run beside the main BAQ kernel
    hipEvent_t side_done = nullptr, side2_done = nullptr;
    hipEvent_t plan_words_ev = nullptr, plan_ready_ev = nullptr;      // mpileup_pipeline: the BAQ plan's words are on the host / what the list kernels read is ready
    std::string err;
    std::map<int32_t, RefSeq> refs;
    // current window
    bool staged = false;
    sta_window win{};
    std::string tname;
    std::vector<FileBufs> fb;
    std::vector<StaReadsDev> files_h;
    std::vector<int32_t> min_pos, max_pos_hint;
    sta_mate_resolver mate_fn = nullptr; void *mate_user = nullptr;     // sta_set_mate_resolver
    int32_t *pin_baq = nullptr; size_t pin_baq_words = 0;               // page-locked: per file the BAQ plan's list length + class-S histogram
    std::vector<char> late_copy;       // mpileup plan, per file: the working quality pool exists only if the window has overlap-eligible reads
    bool gen_xlen_on = false;          // this plan's generic measuring pass filled colinfo / gen_xlen for the emit
    DevBuf baq_list_tmp, files_d, tname_d, bed_d, line_len, colinfo, gen_xlen, wfirst, strip_rng, offs, scan_tmp, counters, table, out, diff, fused_status, maxcnt_scratch, baq_scratch, baq_scratch2, stage_bad, md_cap, cov_out, cov_hist, sc_pos, sc_delta, sc_tmp, sc_cov, glf_tab, glf_redo, md_nm, md_len, md_state, md_tag, md_seq;
    // consensus
    DevBuf cons_tab, cons_ws, cons_E, cons_Enm, cons_cols, cons_depth, cons_coloff, cons_seq, cons_qual, cons_qwork, cons_nm, cons_colpos, cons_gran;
    sta_cons_params cons_p{}; bool cons_tab_ok = false;
    cons::Win cons_w{}; uint64_t cons_ncols = 0, cons_nentries = 0, cons_nstored = 0; int64_t cons_W = 0; bool cons_text = false, cons_walk_all = false;
    int baq_slab_gib_cap = 0;       // 0 = default; 4 after a one-launch BAQ slab could not be allocated
    double glf_depcorr = -1.0;      // theta the coefficient block in glf_tab was computed for
    StaWinDev wd{};
    // plan state
    int planned = 0;   // 1 mpileup, 2 depth, 3 plp entries
    bool plp_mode = false;
    bool cov_mode = false;          // coverage / bedcov: the pipeline stops before the per-column text measuring pass
    int32_t cov_hist_bins = 0;      // coverage -m / -D: bins of the open histogram (sta_cov_hist_begin)
    sta_statcov_params sc{}; int32_t sc_ncov = 0;     // stats COV: the open distribution (sta_statcov_begin)
    sta_mplp_params mp{};
    sta_depth_params dp{};
    StaCounters ctr_h{};
    uint64_t out_bytes = 0;
    uint32_t lds_cap = 0;
    uint64_t n_raw_staged = 0;         // reads whose pools were cut out of raw BAM records on the device (sta_stage_stats)
    bool md_cap_valid = false;         // the last calmd plan computed sam_cap_mapq values (md_cap)
    bool stage_bad_pending = false;    // the device's verdict on the window's raw records is on its way to stage_bad_h (read behind the next synchronisation)
    unsigned long long stage_bad_h[2] = { 0, 0 };
    bool len_fused = false;            // the measuring kernel also produced offsets / totals (no scan, no column statistics)
    bool have_wfirst = false;          // the plan built the per-group read index of the tile kernels
    char *pin = nullptr;               // page-locked: [0, PIN_FILES) counters + text bytes read back, [PIN_FILES, PIN_BYTES) the file descriptors pushed
    DevBuf chunk_words; StaChunkState chunk_st;   // the preparation kernels' in-kernel prefix maximum (kernels_common.hip ChunkScan)
    void *last_out = nullptr;
    // profiling
    bool prof_on = false;
    std::string prof_only;             // sta_profile_only: the one name that is timed (empty: every launch)
    std::map<std::string, ProfEntry> prof;
    std::vector<ProfPending> pending;
    std::vector<hipEvent_t> ev_pool;
};

namespace {

// line-buffer bytes per wave of k_depth_fused: 64 rows at a time (experiment knob STA_DEPTH_LBUF)
uint32_t depth_lbuf()
{
    static const uint32_t v = [] { const char *e = getenv("STA_DEPTH_LBUF"); int x = e ? atoi(e) : 2048; return (uint32_t)(x < 512 ? 512 : x > 32768 ? 32768 : x); }();
    return v;
}

int fail(sta_engine *e, int code, const std::string &msg)
{
    if (e) e->err = msg;
    return code;
}
int hipfail(sta_engine *e, hipError_t r, const char *what)
{
    return fail(e, STA_ERR_HIP, std::string(what) + ": " + hipGetErrorString(r));
}
#define HIPCHK(call) do { hipError_t r_ = (call); if (r_ != hipSuccess) return hipfail(e, r_, #call); } while (0)

// Raw staging (k_bam_pools): the device's verdict on the window's records (offsets the host staged vs what the records say; records
// inside the uploaded bytes) used to be read back with its own synchronisation in sta_stage_window, which serialised the window
// producer with the device on the default BAM path.  It now travels behind the staging kernels and is looked at behind the
// first synchronisation any later call performs: nothing computed from a bad window leaves the engine.
static int stage_verdict(sta_engine *e)
{
    if (!e->stage_bad_pending) return STA_OK;
    e->stage_bad_pending = false;
    const unsigned long long *v = e->pin ? (const unsigned long long *)(e->pin + PIN_FILES - 32) : e->stage_bad_h;
    if (v[0]) return fail(e, STA_ERR_ARG, "raw staging: an alignment record does not match the offsets staged for it (or lies outside the staged bytes)");
    if (v[1]) return fail(e, STA_ERR_HIP, "raw staging (verify): the device-built pools differ from the host-built ones");
    return STA_OK;
}
#define SYNC_STREAM() do { HIPCHK(hipStreamSynchronize(e->stream)); if (int v_ = stage_verdict(e)) return v_; } while (0)
#define SYNC_S(st) do { HIPCHK(hipStreamSynchronize(st)); if (int v_ = stage_verdict(e)) return v_; } while (0)

hipEvent_t get_event(sta_engine *e)
{
    if (!e->ev_pool.empty()) { hipEvent_t ev = e->ev_pool.back(); e->ev_pool.pop_back(); return ev; }
    hipEvent_t ev; hipEventCreate(&ev); return ev;
}
struct ProfScope {
    sta_engine *e; const char *name; hipEvent_t a{}, b{}; hipStream_t st; bool active;
    ProfScope(sta_engine *e_, const char *n, hipStream_t on = nullptr, bool use_on = false) : e(e_), name(n), st(use_on ? on : e_->stream)
    {
        active = e->prof_on && (e->prof_only.empty() || e->prof_only == n);
        if (active) { a = get_event(e); b = get_event(e); hipEventRecord(a, st); }
    }
    ~ProfScope()
    {
        if (active) { hipEventRecord(b, st); e->pending.push_back(ProfPending{ name, a, b }); }
    }
};
void prof_drain(sta_engine *e)
{
    for (auto &p : e->pending) {
        float ms = 0;
        hipEventSynchronize(p.b);
        hipEventElapsedTime(&ms, p.a, p.b);
        ProfEntry &pe = e->prof[p.name];
        pe.launches++; pe.ms += ms;
        e->ev_pool.push_back(p.a); e->ev_pool.push_back(p.b);
    }
    e->pending.clear();
}

template <class T>
int upload(sta_engine *e, DevBuf &b, const T *src, size_t n, const T **dev_out, int mem)
{
    if (mem == STA_MEM_DEVICE) { *dev_out = src; return 0; }
    if (n == 0 || src == nullptr) { if (b.ensure(16)) return fail(e, STA_ERR_HIP, "hipMalloc failed"); *dev_out = src ? (const T *)b.p : nullptr; return 0; }
    if (b.ensure(n * sizeof(T) + 16)) return fail(e, STA_ERR_HIP, "hipMalloc failed");
    HIPCHK(hipMemcpyAsync(b.p, src, n * sizeof(T), hipMemcpyHostToDevice, e->stream));
    *dev_out = (const T *)b.p;
    return 0;
}

}  // namespace

extern "C" {

const char *sta_version(void) { return "samtools_amd 0.1 (MI355X/gfx950 mpileup+depth engine; samtools 1.23.1 semantics)"; }

int sta_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int sta_engine_create(sta_engine **out, int device, void *hip_stream)
{
    if (!out) return STA_ERR_ARG;
    *out = nullptr;
    int n = sta_device_count();
    if (n <= 0 || device < 0 || device >= n) return STA_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return STA_ERR_NO_DEVICE;
    sta_engine *e = new sta_engine();
    e->device = device;
    e->stream = (hipStream_t)hip_stream;   // nullptr = default stream
    // a page-locked block for the plan's small transfers (file descriptors in, counters out): copies from and to pageable memory are
    // staged by the runtime and block the caller; without it (allocation refused) the plan falls back to those
    if (hipHostMalloc(&e->pin, PIN_BYTES, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); e->pin = nullptr; }
    *out = e;
    return STA_OK;
}

void sta_engine_destroy(sta_engine *e)
{
    if (e && getenv("STA_STAGE_REPORT")) fprintf(stderr, "[sta] reads staged on the device out of raw BAM records: %llu\n", (unsigned long long)e->n_raw_staged);
    if (!e) return;
    hipSetDevice(e->device);
    hipStreamSynchronize(e->stream);
    for (auto &f : e->fb) f.release();
    for (auto &r : e->refs) r.second.buf.release();
    if (e->pin) hipHostFree(e->pin);
    if (e->pin_baq) hipHostFree(e->pin_baq);
    DevBuf *all[] = { &e->baq_list_tmp, &e->files_d, &e->tname_d, &e->bed_d, &e->line_len, &e->colinfo, &e->gen_xlen, &e->wfirst, &e->strip_rng, &e->offs, &e->scan_tmp, &e->counters, &e->table,
                      &e->out, &e->diff, &e->fused_status, &e->maxcnt_scratch, &e->baq_scratch, &e->baq_scratch2, &e->stage_bad, &e->md_cap, &e->chunk_words, &e->cov_out, &e->cov_hist, &e->sc_pos, &e->sc_delta, &e->sc_tmp, &e->sc_cov, &e->glf_tab, &e->glf_redo, &e->md_nm, &e->md_len, &e->md_state, &e->md_tag, &e->md_seq,
                      &e->cons_tab, &e->cons_ws, &e->cons_E, &e->cons_Enm, &e->cons_cols, &e->cons_depth, &e->cons_coloff, &e->cons_seq, &e->cons_qual, &e->cons_qwork, &e->cons_nm, &e->cons_colpos, &e->cons_gran };
    for (DevBuf *b : all) b->release();
    if (e->side) hipStreamDestroy(e->side);
    if (e->side_done) hipEventDestroy(e->side_done);
    if (e->side2) hipStreamDestroy(e->side2);
    if (e->side2_done) hipEventDestroy(e->side2_done);
    if (e->plan_words_ev) hipEventDestroy(e->plan_words_ev);
    if (e->plan_ready_ev) hipEventDestroy(e->plan_ready_ev);
    for (auto &p : e->pending) { hipEventDestroy(p.a); hipEventDestroy(p.b); }
    for (auto ev : e->ev_pool) hipEventDestroy(ev);
    delete e;
}

const char *sta_last_error(const sta_engine *e) { return e ? e->err.c_str() : "no engine"; }

int sta_set_reference(sta_engine *e, int32_t tid, const char *seq, int64_t len, int32_t mem)
{
    if (!e || len < 0) return STA_ERR_ARG;
    hipSetDevice(e->device);
    RefSeq &r = e->refs[tid];
    r.len = len;
    if (mem == STA_MEM_DEVICE) { r.external = true; r.ext = seq; return STA_OK; }
    r.external = false;
    if (r.buf.ensure((size_t)len + 16)) return fail(e, STA_ERR_HIP, "hipMalloc(reference) failed");
    if (len) HIPCHK(hipMemcpyAsync(r.buf.p, seq, (size_t)len, hipMemcpyHostToDevice, e->stream));
    SYNC_STREAM();
    return STA_OK;
}

void sta_clear_references(sta_engine *e)
{
    if (!e) return;
    hipSetDevice(e->device);
    hipStreamSynchronize(e->stream);
    for (auto &r : e->refs) r.second.buf.release();
    e->refs.clear();
}

int sta_stage_window(sta_engine *e, const sta_window *w)
{
    if (!e || !w || w->n_files < 0 || w->col_end < w->col_beg) return fail(e, STA_ERR_ARG, "bad window");
    hipSetDevice(e->device);
    e->staged = false; e->planned = 0;
    e->win = *w;
    e->tname = w->tname ? w->tname : "";
    if (e->fb.size() < (size_t)w->n_files) e->fb.resize((size_t)w->n_files);
    e->files_h.assign((size_t)w->n_files, StaReadsDev{});
    bool any_raw = false;
    if (e->stage_bad_pending) {
        // a window staged and never used: its verdict is not lost -- and is reported as what it is, the PREVIOUS window's
        HIPCHK(hipStreamSynchronize(e->stream));
        if (int v_ = stage_verdict(e)) { e->err = "the previously staged window (never planned): " + e->err; return v_; }
    }
    for (int f = 0; f < w->n_files; ++f) {
        const sta_reads &r = w->files[f];
        FileBufs &b = e->fb[(size_t)f];
        StaReadsDev &d = e->files_h[(size_t)f];
        size_t n = (size_t)r.n_reads;
        d.n = r.n_reads;
        d.n_bases_total = r.n_bases_total;
        int rc = 0, mem = w->mem;
        rc |= upload(e, b.pos, r.pos, n, &d.pos, mem);
        rc |= upload(e, b.flag, r.flag, n, &d.flag, mem);
        rc |= upload(e, b.mapq, r.mapq, n, &d.mapq, mem);
        rc |= upload(e, b.aux, r.aux, n, &d.aux, mem);
        rc |= upload(e, b.lq, r.l_qseq, n, &d.l_qseq, mem);
        rc |= upload(e, b.cig_off, r.cig_off, n + 1, &d.cig_off, mem);
        rc |= upload(e, b.base_off8, r.base_off8, n, &d.base_off8, mem);
        rc |= upload(e, b.mtid, r.mtid, n, &d.mtid, mem);
        rc |= upload(e, b.mpos, r.mpos, n, &d.mpos, mem);
        rc |= upload(e, b.isize, r.isize, n, &d.isize, mem);
        rc |= upload(e, b.name_off, r.name_off, n + 1, &d.name_off, mem);
        // device staging: the pools of reads [raw_first, n) come out of raw BAM records (kernels_stage.hip)
        const bool raw_mode = mem == STA_MEM_HOST && r.n_raw_pieces > 0 && r.raw_pieces && r.raw_rec_off && r.raw_first >= 0 && r.raw_first < r.n_reads && !r.bq;
        size_t cig0 = 0, bases0 = 0, names0 = 0;
        if (raw_mode && !r.raw_verify) {
            // only the host-written prefixes of the pools travel as pools
            cig0 = r.cig_off[r.raw_first]; bases0 = (size_t)r.base_off8[r.raw_first] << 3; names0 = r.name_off[r.raw_first];
            if (b.cigar.ensure((size_t)r.n_cigar_total * 4 + 16) || b.seq.ensure((size_t)(r.n_bases_total / 2) + 16) || b.qual.ensure((size_t)r.n_bases_total + 16)
                || b.names.ensure((size_t)r.n_name_bytes + 16)) return fail(e, STA_ERR_HIP, "hipMalloc failed");
            if (cig0) HIPCHK(hipMemcpyAsync(b.cigar.p, r.cigar, cig0 * 4, hipMemcpyHostToDevice, e->stream));
            if (bases0) { HIPCHK(hipMemcpyAsync(b.seq.p, r.seq, bases0 / 2, hipMemcpyHostToDevice, e->stream)); HIPCHK(hipMemcpyAsync(b.qual.p, r.qual, bases0, hipMemcpyHostToDevice, e->stream)); }
            if (names0) HIPCHK(hipMemcpyAsync(b.names.p, r.names, names0, hipMemcpyHostToDevice, e->stream));
            d.cigar = (const uint32_t *)b.cigar.p; d.seq = (const uint8_t *)b.seq.p; d.qual_in = (const uint8_t *)b.qual.p; d.names = (const char *)b.names.p;
            d.bq = nullptr;
        } else {
            rc |= upload(e, b.cigar, r.cigar, (size_t)r.n_cigar_total, &d.cigar, mem);
            rc |= upload(e, b.seq, r.seq, (size_t)(r.n_bases_total / 2), &d.seq, mem);
            rc |= upload(e, b.qual, r.qual, (size_t)r.n_bases_total, &d.qual_in, mem);
            if (r.bq) rc |= upload(e, b.bq, r.bq, (size_t)r.n_bases_total, &d.bq, mem); else d.bq = nullptr;
            rc |= upload(e, b.names, r.names, (size_t)r.n_name_bytes, &d.names, mem);
        }
        if (raw_mode && !rc) {
            const int64_t n_raw = r.n_reads - r.raw_first;
            uint64_t raw_bytes = 0;
            for (int32_t k = 0; k < r.n_raw_pieces; ++k) raw_bytes += r.raw_pieces[k].n_bytes;
            if (raw_bytes > 0xfffffff0ull) return fail(e, STA_ERR_ARG, "raw staging pieces beyond 4 GiB");
            if (b.raw.ensure((size_t)raw_bytes + 64) || b.raw_off.ensure((size_t)n_raw * 4 + 16) || e->stage_bad.ensure(64)) return fail(e, STA_ERR_HIP, "hipMalloc(raw staging) failed");
            uint64_t o = 0;
            for (int32_t k = 0; k < r.n_raw_pieces; ++k) {
                if (r.raw_pieces[k].n_bytes) HIPCHK(hipMemcpyAsync((char *)b.raw.p + o, r.raw_pieces[k].bytes, (size_t)r.raw_pieces[k].n_bytes, hipMemcpyHostToDevice, e->stream));
                o += r.raw_pieces[k].n_bytes;
            }
            HIPCHK(hipMemcpyAsync(b.raw_off.p, r.raw_rec_off, (size_t)n_raw * 4, hipMemcpyHostToDevice, e->stream));
            if (!any_raw) { HIPCHK(hipMemsetAsync(e->stage_bad.p, 0, 16, e->stream)); any_raw = true; }
            unsigned long long *bad = (unsigned long long *)e->stage_bad.p;
            uint32_t *o_cig; uint8_t *o_seq, *o_qual; char *o_names;
            if (r.raw_verify) {
                // build beside the (complete, host-written) pools and compare
                const size_t cb = ((size_t)r.n_cigar_total * 4 + 63) & ~(size_t)63, sb = ((size_t)(r.n_bases_total / 2) + 63) & ~(size_t)63, qb = ((size_t)r.n_bases_total + 63) & ~(size_t)63;
                if (b.raw_vfy.ensure(cb + sb + qb + (size_t)r.n_name_bytes + 64)) return fail(e, STA_ERR_HIP, "hipMalloc(raw staging) failed");
                o_cig = (uint32_t *)b.raw_vfy.p; o_seq = (uint8_t *)b.raw_vfy.p + cb; o_qual = o_seq + sb; o_names = (char *)(o_qual + qb);
                cig0 = r.cig_off[r.raw_first]; bases0 = (size_t)r.base_off8[r.raw_first] << 3; names0 = r.name_off[r.raw_first];
            } else { o_cig = (uint32_t *)b.cigar.p; o_seq = (uint8_t *)b.seq.p; o_qual = (uint8_t *)b.qual.p; o_names = (char *)b.names.p; }
            sta_launch_bam_pools(e->stream, (const uint8_t *)b.raw.p, (const uint32_t *)b.raw_off.p, raw_bytes, r.raw_first, n_raw, d, o_cig, o_seq, o_qual, o_names, bad);
            if (r.raw_verify) {
                sta_launch_stage_compare(e->stream, (const char *)o_cig + cig0 * 4, (const char *)d.cigar + cig0 * 4, ((size_t)r.n_cigar_total - cig0) * 4, bad + 1);
                sta_launch_stage_compare(e->stream, o_seq + bases0 / 2, d.seq + bases0 / 2, (size_t)(r.n_bases_total - bases0) / 2, bad + 1);
                sta_launch_stage_compare(e->stream, o_qual + bases0, d.qual_in + bases0, (size_t)(r.n_bases_total - bases0), bad + 1);
                sta_launch_stage_compare(e->stream, o_names + names0, d.names + names0, (size_t)r.n_name_bytes - names0, bad + 1);
            }
            e->n_raw_staged += (uint64_t)n_raw;
        }
        d.n_xcols = r.n_xcols > 0 && r.xcol_off && r.xcol_text ? r.n_xcols : 0;
        d.xcol_off = nullptr; d.xcol_text = nullptr;
        if (d.n_xcols) {
            rc |= upload(e, b.xoff, r.xcol_off, n * (size_t)d.n_xcols + 1, &d.xcol_off, mem);
            rc |= upload(e, b.xtext, r.xcol_text, (size_t)r.n_xcol_bytes, &d.xcol_text, mem);
        }
        d.mod_off = nullptr; d.mod_qpos = nullptr; d.mod_toff = nullptr; d.mod_text = nullptr;
        if (r.mod_off && r.mod_toff) {
            rc |= upload(e, b.moff, r.mod_off, n + 1, &d.mod_off, mem);
            rc |= upload(e, b.mqpos, r.mod_qpos, (size_t)r.n_mod_entries, &d.mod_qpos, mem);
            rc |= upload(e, b.mtoff, r.mod_toff, (size_t)r.n_mod_entries + 1, &d.mod_toff, mem);
            rc |= upload(e, b.mtext, r.mod_text, (size_t)r.n_mod_bytes, &d.mod_text, mem);
        }
        d.clip_in = nullptr; d.mate = nullptr;
        if (r.olap_clip && n) rc |= upload(e, b.clip_in, r.olap_clip, n, &d.clip_in, mem);
        // (STA_OLAP_DEVICE_TABLE=1, tests: the caller's partners are ignored and the window's own name table decides -- the path of callers that
        //  stage no partners; right whenever the window holds every record of its templates)
        static const bool own_table = getenv("STA_OLAP_DEVICE_TABLE") && atoi(getenv("STA_OLAP_DEVICE_TABLE")) != 0;
        if (r.olap_mate && n && !own_table) rc |= upload(e, b.mate, r.olap_mate, n, &d.mate, mem);
        if (rc) return rc;
        // workspace
        if (b.end.ensure(n * 4 + 16) || b.maxend.ensure(n * 4 + 16) || b.info.ensure(n * 4 + 16) || b.clip.ensure(n * 4 + 16)
            || b.chain.ensure(n * 4 + 16))
            return fail(e, STA_ERR_HIP, "hipMalloc(workspace) failed");
        d.end = (int32_t *)b.end.p; d.maxend = (int32_t *)b.maxend.p; d.info = (uint32_t *)b.info.p; d.clip = (int32_t *)b.clip.p; d.chain = (int32_t *)b.chain.p;
        d.s_ws = nullptr;        // (set by the plan when BAQ runs)
        d.qual = const_cast<uint8_t *>(d.qual_in);
    }
    if (any_raw) {
        static_assert(sizeof(StaCounters) + 8 <= PIN_FILES - 32, "the staging verdict's page-locked slot overlaps the counter block");
        HIPCHK(hipMemcpyAsync(e->pin ? (void *)(e->pin + PIN_FILES - 32) : (void *)e->stage_bad_h, e->stage_bad.p, 16, hipMemcpyDeviceToHost, e->stream));
        e->stage_bad_pending = true;
    }
    // window constants
    StaWinDev &wd = e->wd;
    wd = StaWinDev{};
    wd.col_beg = w->col_beg; wd.col_end = w->col_end; wd.origin = w->origin; wd.tid = w->tid; wd.tlen = w->tlen;
    wd.nfiles = w->n_files;
    if (e->tname_d.ensure(e->tname.size() + 16)) return fail(e, STA_ERR_HIP, "hipMalloc failed");
    if (!e->tname.empty()) HIPCHK(hipMemcpyAsync(e->tname_d.p, e->tname.data(), e->tname.size(), hipMemcpyHostToDevice, e->stream));
    wd.tname = (const char *)e->tname_d.p; wd.tname_len = (int32_t)e->tname.size();
    wd.has_bed = w->has_bed; wd.n_bed = w->has_bed ? w->n_bed : 0;
    if (w->has_bed) {
        size_t nb = (size_t)w->n_bed;
        if (e->bed_d.ensure(nb * 16 + 32)) return fail(e, STA_ERR_HIP, "hipMalloc failed");
        if (nb) {
            HIPCHK(hipMemcpyAsync(e->bed_d.p, w->bed_beg, nb * 8, hipMemcpyHostToDevice, e->stream));
            HIPCHK(hipMemcpyAsync((char *)e->bed_d.p + nb * 8, w->bed_end, nb * 8, hipMemcpyHostToDevice, e->stream));
        }
        wd.bed_beg = (const int64_t *)e->bed_d.p; wd.bed_end = wd.bed_beg + nb;
    }
    wd.has_reg = w->has_reg; wd.reg_beg = w->reg_beg; wd.reg_end = w->reg_end;
    wd.baq_plain = 0;
    auto it = e->refs.find(w->tid);
    if (it != e->refs.end()) { wd.ref = it->second.external ? it->second.ext : (const char *)it->second.buf.p; wd.ref_len = it->second.len; }
    else { wd.ref = nullptr; wd.ref_len = 0; }
    int64_t ncols = (int64_t)w->col_end - w->col_beg;
    if (e->line_len.ensure((size_t)(ncols + 1) * 4 + 16) || e->offs.ensure((size_t)(ncols + 2) * 8 + 16)
        || e->scan_tmp.ensure(sta_scan_tmp_bytes(ncols > 0 ? ncols : 1) + 64) || e->counters.ensure(sizeof(StaCounters)))
        return fail(e, STA_ERR_HIP, "hipMalloc(column workspace) failed");
    e->staged = true;
    return STA_OK;
}

static int push_files(sta_engine *e)
{
    size_t bytes = e->files_h.size() * sizeof(StaReadsDev);
    if (e->files_d.ensure(bytes + 16)) return fail(e, STA_ERR_HIP, "hipMalloc failed");
    const void *src = e->files_h.data();
    if (e->pin && bytes <= PIN_BYTES - PIN_FILES) {
        // (the previous plan's copy out of this block has completed: every plan ends with a synchronisation behind it)
        memcpy(e->pin + PIN_FILES, e->files_h.data(), bytes);
        src = e->pin + PIN_FILES;
    }
    if (bytes) HIPCHK(hipMemcpyAsync(e->files_d.p, src, bytes, hipMemcpyHostToDevice, e->stream));
    e->wd.files = (const StaReadsDev *)e->files_d.p;
    return STA_OK;
}

static int finish_plan(sta_engine *e, int64_t ncols, sta_plan_info *info)
{
    if (!e->len_fused) {
        {
            ProfScope ps(e, "len_scan");
            sta_launch_len_scan(e->stream, (const uint32_t *)e->line_len.p, (uint64_t *)e->offs.p, ncols, e->scan_tmp.p, e->scan_tmp.cap);
        }
        ProfScope ps(e, "col_stats");
        sta_launch_wave_bytes_max(e->stream, (const uint64_t *)e->offs.p, (const uint32_t *)e->line_len.p, ncols, (StaCounters *)e->counters.p);
    }
    uint64_t total = 0;
    static_assert(sizeof(StaCounters) + 8 <= PIN_FILES, "counter block outgrew its page-locked slot");
    StaCounters *ctr_dst = e->pin ? (StaCounters *)e->pin : &e->ctr_h;
    uint64_t *total_dst = e->pin ? (uint64_t *)(e->pin + sizeof(StaCounters)) : &total;
    HIPCHK(hipMemcpyAsync(ctr_dst, e->counters.p, sizeof(StaCounters), hipMemcpyDeviceToHost, e->stream));
    // (after the tile measuring kernel the offsets are tile-relative and k_tile_scan left the window's text bytes in the counter block)
    if (!e->len_fused) HIPCHK(hipMemcpyAsync(total_dst, (const uint64_t *)e->offs.p + (ncols > 0 ? ncols : 0), 8, hipMemcpyDeviceToHost, e->stream));
    SYNC_STREAM();
    if (e->pin) { e->ctr_h = *ctr_dst; total = *total_dst; }
    if (e->len_fused) total = e->ctr_h.out_bytes;
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return hipfail(e, le, "kernel launch");
    e->out_bytes = total;
    uint64_t mw = e->ctr_h.max_wave_bytes;
    uint32_t cap = (uint32_t)((mw + 255) & ~255ull);
    if (cap < 1024) cap = 1024;
    if (cap > 65536 - 64) cap = 65536 - 64;
    e->lds_cap = cap;
    if (info) {
        info->out_bytes = total; info->n_lines = e->ctr_h.n_lines; info->n_data_cols = e->ctr_h.n_data_cols;
        info->n_kept_reads = e->ctr_h.n_kept; info->piled_bases = e->ctr_h.piled_bases; info->n_maxcnt_dropped = e->ctr_h.n_dropped;
    }
    return STA_OK;
}

// the zero-initialised words of the preparation kernels' prefix maximum (allocated once per engine)
static int chunk_state(sta_engine *e)
{
    if (e->chunk_st.words) return STA_OK;
    if (e->chunk_words.ensure(STA_CHUNK_WORDS * 8)) return fail(e, STA_ERR_HIP, "hipMalloc(scan words) failed");
    HIPCHK(hipMemsetAsync(e->chunk_words.p, 0, STA_CHUNK_WORDS * 8, e->stream));
    e->chunk_st = StaChunkState{};
    e->chunk_st.words = (unsigned long long *)e->chunk_words.p;
    return STA_OK;
}

static int mpileup_pipeline(sta_engine *e, const sta_mplp_params *p, bool do_maxcnt)
{
    hipStream_t s = e->stream;
    StaCounters *ctr = (StaCounters *)e->counters.p;
    HIPCHK(hipMemsetAsync(ctr, 0, sizeof(StaCounters), s));
    e->len_fused = false;
    const int nf = (int)e->files_h.size();
    bool has_ref = e->wd.ref != nullptr;
    bool realn = (p->flag & STA_MPLP_REALN) && has_ref;
    bool redo = (p->flag & STA_MPLP_REDO_BAQ) != 0;
    bool olap = (p->flag & STA_MPLP_SMART_OVERLAPS) != 0;
    bool illum = (p->flag & STA_MPLP_ILLUMINA13) != 0;
    // Working quality pool: built up front (k_qual_prep) when something rewrites qualities whatever the reads look like (-6, BAQ,
    // BQ tags, -C); when only the mate-overlap pass could, the copy waits for k_prep_reads' count of eligible reads and is decided on
    // the device (k_olap_setup) -- a window without proper pairs is then read straight from the input pool.
    const bool capq = has_ref && p->capQ_thres > 10;
    std::vector<char> &late_copy = e->late_copy;      // per file: the working pool is k_olap_setup's business
    late_copy.assign((size_t)nf, 0);
    std::vector<int> deferred_qp;                     // files whose working pool is copied while the host waits for the BAQ plan's words
    for (int f = 0; f < nf; ++f) {
        StaReadsDev &d = e->files_h[(size_t)f];
        bool tag_bq = realn && !redo && d.bq != nullptr;
        const bool up_front = illum || realn || capq;
        if ((up_front || olap) && d.n_bases_total) {
            FileBufs &b = e->fb[(size_t)f];
            if (b.qual_work.ensure((size_t)d.n_bases_total + 32)) return fail(e, STA_ERR_HIP, "hipMalloc(qual) failed");
            d.qual = (uint8_t *)b.qual_work.p;
            if (up_front && realn && !tag_bq) deferred_qp.push_back(f);      // behind k_prep_reads, inside the plan's host round trip (below)
            else if (up_front) {
                // (with a BQ:Z pool k_prep_reads puts the bytes of turned-away records back: the pool copy has to be there first)
                StaReadsDev tmp = d;
                if (!tag_bq) tmp.bq = nullptr;
                ProfScope ps(e, "qual_prep");
                sta_launch_qual_prep(s, tmp, illum ? 1 : 0);
            } else late_copy[(size_t)f] = 1;
        } else d.qual = const_cast<uint8_t *>(d.qual_in);
        d.fix_y = nullptr; d.fix_mate = nullptr; d.fix_q = nullptr;
        if (olap && d.n) {
            FileBufs &b = e->fb[(size_t)f];
            if (b.fix_y.ensure((size_t)d.n * 4 + 16) || b.fix_mate.ensure((size_t)d.n * 4 + 16) || b.fix_q.ensure((size_t)d.n + 16))
                return fail(e, STA_ERR_HIP, "hipMalloc(overlap fix-up) failed");
            d.fix_y = (int32_t *)b.fix_y.p; d.fix_mate = (int32_t *)b.fix_mate.p; d.fix_q = (uint8_t *)b.fix_q.p;
        }
    }
    int rc = push_files(e);
    if (rc) return rc;
    rc = chunk_state(e);
    if (rc) return rc;
    // the text plan's tile kernels find their reads through a column -> read index built from the input positions: the first
    // preparation launch carries it
    const int64_t ncols = (int64_t)e->wd.col_end - e->wd.col_beg;
    e->have_wfirst = false;
    if (!e->cov_mode && !e->plp_mode && (sta_mplp_has_fast_path(*p) || sta_mplp_has_xfast_path(*p)) && sta_mplp_tile_ok(*p)) {
        bool small = true;
        for (int f = 0; f < nf; ++f) small = small && e->files_h[(size_t)f].n < 0xffffffffll;
        e->have_wfirst = small;
    }
    bool wf_done = false;
    if (e->have_wfirst && e->wfirst.ensure((size_t)((ncols > 0 ? ncols : 1) / 64 + 2) * (size_t)(nf > 0 ? nf : 1) * 4 + 16)) return fail(e, STA_ERR_HIP, "hipMalloc(read index) failed");
    for (int f = 0; f < nf; ++f) {
        // class-S BAQ: histogram / layout / cursors / list of the candidates (sta_dev.h StaReadsDev::s_ws); the padding of 241 lengths to
        // whole groups is at most 241 x 63 entries
        StaReadsDev &d = e->files_h[(size_t)f];
        d.s_ws = nullptr;
        if (!realn || !d.n) continue;
        FileBufs &b = e->fb[(size_t)f];
        if (b.slist.ensure(((size_t)d.n + 64 * (size_t)STA_SLIST_BINS + STA_SLIST_HEAD) * 4 + 64)) return fail(e, STA_ERR_HIP, "hipMalloc(class-S list) failed");
        d.s_ws = (int32_t *)b.slist.p;
        HIPCHK(hipMemsetAsync(d.s_ws, 0, (size_t)STA_SLIST_HEAD * 4, s));
    }
    {
        ProfScope ps(e, "prep_reads");
        wf_done = sta_launch_prep_reads(s, e->wd, e->files_h.data(), nf, *p, ctr, e->chunk_st, e->have_wfirst ? (uint32_t *)e->wfirst.p : nullptr);
    }
    // the BAQ list of every file into class order (band width 7 | 8 | general: kernels_baq.hip k_baq_list_partition): one small workgroup per
    // file, launched below while the host waits for the counters.  STA_BAQ_LIST_SORT=0: not at all (every list kernel over the whole list, as in round 5)
    bool list_sorted = false;
    std::vector<size_t> list_tmp_off((size_t)nf, 0);
    if (realn && !(getenv("STA_BAQ_LIST_SORT") && atoi(getenv("STA_BAQ_LIST_SORT")) == 0)) {
        size_t words = 0;
        for (int f = 0; f < nf; ++f) { list_tmp_off[(size_t)f] = words; words += (size_t)e->files_h[(size_t)f].n + 4; }
        if (e->baq_list_tmp.ensure(words * 4 + 64)) return fail(e, STA_ERR_HIP, "hipMalloc(BAQ list order) failed");
        list_sorted = true;
    }
    if (realn) {
        // geometry bounds of the reads that need BAQ (written by k_prep_reads; maxima over all files)
        StaCounters c{};
        StaCounters *c_dst = e->pin ? (StaCounters *)e->pin : &c;
        HIPCHK(hipMemcpyAsync(c_dst, ctr, sizeof(c), hipMemcpyDeviceToHost, s));
        // ... and with them, in the SAME round trip, what the plan used to fetch in two more: every file's list length
        // (chain[0]) and its class-S histogram of candidate lengths -- all three are k_prep_reads' output.  Page-locked, 1 + STA_SLIST_BINS
        // words per file.
        const size_t per_file = 1 + (size_t)STA_SLIST_BINS;
        if (e->pin_baq_words < (size_t)nf * per_file) {
            if (e->pin_baq) { hipHostFree(e->pin_baq); e->pin_baq = nullptr; e->pin_baq_words = 0; }
            if (hipHostMalloc((void **)&e->pin_baq, (size_t)nf * per_file * 4, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return fail(e, STA_ERR_HIP, "hipHostMalloc(BAQ plan words) failed"); }
            e->pin_baq_words = (size_t)nf * per_file;
        }
        for (int f = 0; f < nf; ++f) {
            StaReadsDev &d = e->files_h[(size_t)f];
            int32_t *w = e->pin_baq + (size_t)f * per_file;
            w[0] = 0;
            if (!d.n) continue;
            HIPCHK(hipMemcpyAsync(w, d.chain, 4, hipMemcpyDeviceToHost, s));
            if (d.s_ws) HIPCHK(hipMemcpyAsync(w + 1, d.s_ws, (size_t)STA_SLIST_BINS * 4, hipMemcpyDeviceToHost, s));
        }
        // The host waits for these words only (an event, not the stream): what does not depend on them -- the working copy of the quality
        // pool, the list's class order -- is launched behind the event and runs during the round trip (they were 0.06 ms in front of it).
        if (!e->plan_words_ev && hipEventCreateWithFlags(&e->plan_words_ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return fail(e, STA_ERR_HIP, "hipEventCreate failed"); }
        if (!e->plan_ready_ev && hipEventCreateWithFlags(&e->plan_ready_ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return fail(e, STA_ERR_HIP, "hipEventCreate failed"); }
        HIPCHK(hipEventRecord(e->plan_words_ev, s));
        for (int f : deferred_qp) {
            StaReadsDev tmp = e->files_h[(size_t)f];
            tmp.bq = nullptr;
            ProfScope ps(e, "qual_prep");
            sta_launch_qual_prep(s, tmp, illum ? 1 : 0);
        }
        deferred_qp.clear();
        if (list_sorted) {
            ProfScope ps(e, "baq_list_order");
            for (int f = 0; f < nf; ++f) if (e->files_h[(size_t)f].n) sta_launch_baq_list_partition(s, e->files_h[(size_t)f], (int32_t *)e->baq_list_tmp.p + list_tmp_off[(size_t)f]);
        }
        HIPCHK(hipEventRecord(e->plan_ready_ev, s));         // (the list kernels' side streams start behind this)
        HIPCHK(hipEventSynchronize(e->plan_words_ev));
        if (int v_ = stage_verdict(e)) return v_;
        if (e->pin) c = *c_dst;
        if (getenv("STA_DEBUG")) fprintf(stderr, "[sta] n_baq=%llu (fast %llu, lq<=%llu) slow: max_lq=%llu max_bw=%llu kept=%llu\n", c.n_baq, c.n_baq_fast, c.max_lq_fast, c.max_lq, c.max_bw, c.n_kept);
        if (getenv("STA_DEBUG")) fprintf(stderr, "[sta] bw8=%llu general=%llu class_s=%llu (lq<=%llu) bw7_list=%llu\n", c.n_baq_bw8, c.n_baq_general, c.n_baq_s, c.max_lq_s, c.n_baq_bw7l);
        for (int f = 0; f < nf && c.n_baq; ++f) {
            StaReadsDev &d = e->files_h[(size_t)f];
            if (!d.n) continue;
            const int32_t *plan_words = e->pin_baq + (size_t)f * per_file;
            const int32_t n_list = c.n_baq > c.n_baq_fast + c.n_baq_s ? plan_words[0] : 0;
            const bool has_main = c.n_baq_fast || c.n_baq_s;
            const bool has_list_band = c.n_baq_bw8 || c.n_baq_bw7l;
            if (has_main || has_list_band) {
                // band-in-registers kernels: groups of 64 reads, one scratch slot (forward rows) per group in flight
                int gpl = 0;
                size_t need = sta_baq_band_scratch_bytes(d.n, (int)c.max_lq_fast, &gpl, e->baq_slab_gib_cap);
                // class S: the histogram of candidate lengths comes back, every length gets its run of whole groups in the list
                int s_waves = 0;
                int64_t s_groups = 0;
                if (c.n_baq_s && d.s_ws) {
                    const int32_t *hist = plan_words + 1;
                    int32_t base[STA_SLIST_BINS];
                    int64_t at = 0;
                    for (int l = 0; l < STA_SLIST_BINS; ++l) { base[l] = (int32_t)at; at += ((int64_t)hist[l] + 63) / 64 * 64; }
                    s_groups = at / 64;
                    if (at > d.n + 64 * (int64_t)STA_SLIST_BINS) return fail(e, STA_ERR_HIP, "class-S histogram out of range");
                    HIPCHK(hipMemcpyAsync(d.s_ws + STA_SLIST_BINS, base, sizeof base, hipMemcpyHostToDevice, s));
                    if (at) HIPCHK(hipMemsetAsync(d.s_ws + STA_SLIST_HEAD, 0xff, (size_t)at * 4, s));
                    ProfScope ps(e, "baq_s_gather");
                    sta_launch_baq7s_gather(s, d);
                }
                const size_t need_s = s_groups ? sta_baq7s_scratch_bytes((int)c.max_lq_s, s_groups, &s_waves) : 0;
                if (!c.n_baq_fast) {
                    // nothing is taken in place (class S is on): the band slab only has to hold the list's groups when they run on this stream
                    need = sta_baq_band_scratch_bytes(has_list_band ? (int64_t)n_list : 0, (int)c.max_lq_fast, &gpl, e->baq_slab_gib_cap);
                }
                if (e->baq_scratch.ensure(std::max(need, need_s) + 64)) {
                    // not enough free HBM for the one-launch slab: this engine falls back to a 4 GiB slab (more, smaller launches)
                    (void)hipGetLastError();
                    e->baq_slab_gib_cap = 4;
                    need = sta_baq_band_scratch_bytes(c.n_baq_fast ? d.n : (int64_t)n_list, (int)c.max_lq_fast, &gpl, e->baq_slab_gib_cap);
                    if (e->baq_scratch.ensure(std::max(need, need_s) + 64)) return fail(e, STA_ERR_HIP, "hipMalloc(BAQ scratch) failed");
                }
                // The list's band kernels (band width 8, and band width 7 reads that class S does not take: a few dozen groups,
                // latency bound) run on a side stream beside the main kernel(s) when both exist; the side streams wait for plan_ready_ev
                // (the host only waited for the plan's words), and the main stream waits for them before the qualities are used
                const size_t slot_bytes = need / (size_t)(gpl > 0 ? gpl : 1);
                const int64_t groupsL = has_list_band ? ((int64_t)n_list + 63) / 64 : 0;
                bool side = has_main && groupsL > 0 && groupsL <= 4096 && !getenv("STA_BAQ_NO_SIDE_STREAM");
                char *side_scratch = nullptr;
                if (side) {
                    // (one region per list class: the two run at the same time, and their slot layouts differ)
                    size_t extra = (size_t)groupsL * slot_bytes * 2;
                    if (e->baq_scratch2.ensure(extra + 64)) { (void)hipGetLastError(); side = false; }
                    else side_scratch = (char *)e->baq_scratch2.p;
                    if (side && !e->side) {
                        if (hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&e->side_done, hipEventDisableTiming) != hipSuccess
                            || hipStreamCreateWithFlags(&e->side2, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&e->side2_done, hipEventDisableTiming) != hipSuccess) {
                            (void)hipGetLastError(); side = false;
                        }
                    }
                }
                // classes: 2 = band width 8 through the list, 1 = band width 7 through the list, 0 = band width 7 in place (round-3
                // kernels, only with STA_BAQ_CLASS_S=0), then class S.  On the side streams a list class is ONE launch (both passes):
                // its waves are placed before the persistent class-S kernel fills the chip and run to the end beside it.
                // both band classes in the list: it was put in class order behind k_prep_reads (above); each list kernel's workgroups outside its
                // own groups leave at once
                const int32_t *list_rng = nullptr;
                if (list_sorted && side && c.n_baq_bw8 && c.n_baq_bw7l && n_list > 0) list_rng = (const int32_t *)e->baq_list_tmp.p + list_tmp_off[(size_t)f] + n_list;
                for (int cls = 2; cls >= 0; --cls) {
                    int64_t items = cls == 0 ? (c.n_baq_fast ? d.n : 0) : cls == 1 ? (c.n_baq_bw7l ? (int64_t)n_list : 0) : (c.n_baq_bw8 ? (int64_t)n_list : 0);
                    int64_t ngroups = (items + 63) / 64;
                    const bool on_side = side && cls != 0;
                    static const char *const names[3][2] = { { "baq_fwd", "baq_bwd" }, { "baq7l_fwd", "baq7l_bwd" }, { "baq8_fwd", "baq8_bwd" } };
                    if (on_side) {
                        hipStream_t st = cls == 2 ? e->side : e->side2;
                        HIPCHK(hipStreamWaitEvent(st, e->plan_ready_ev, 0));      // the working qualities and the list's order
                        if (ngroups) {
                            ProfScope ps(e, cls == 2 ? "baq8_list" : "baq7_list", st, true);
                            sta_launch_baq_list(st, d, e->wd, side_scratch + (cls == 2 ? 0 : (size_t)groupsL * slot_bytes), (int)c.max_lq_fast, cls == 2 ? 8 : 7, ngroups,
                                                list_rng ? list_rng + (cls == 2 ? 2 : 0) : nullptr);
                        }
                        HIPCHK(hipEventRecord(cls == 2 ? e->side_done : e->side2_done, st));
                        continue;
                    }
                    int64_t step = gpl;
                    for (int64_t g0 = 0; g0 < ngroups; g0 += step) {
                        int64_t ng = ngroups - g0 < step ? ngroups - g0 : step;
                        for (int pass = 0; pass < 2; ++pass) {
                            ProfScope ps(e, names[cls][pass], s, true);
                            sta_launch_baq_band(s, d, e->wd, e->baq_scratch.p, (int)c.max_lq_fast, cls == 2 ? 8 : 7, g0, ng, cls != 0, pass);
                        }
                    }
                }
                if (s_groups) {
                    ProfScope ps(e, "baq_s");
                    sta_launch_baq7s(s, d, e->wd, e->baq_scratch.p, (int)c.max_lq_s, s_waves, s_groups);
                }
                if (side) { HIPCHK(hipStreamWaitEvent(s, e->side_done, 0)); HIPCHK(hipStreamWaitEvent(s, e->side2_done, 0)); }
            }
            if (c.n_baq_general && n_list) {
                size_t need = sta_baq_scratch_bytes(d.n, (int)c.max_lq, (int)c.max_bw);
                if (e->baq_scratch.ensure(need + 64)) return fail(e, STA_ERR_HIP, "hipMalloc(BAQ scratch) failed");
                ProfScope ps(e, "baq_general");
                sta_launch_baq(s, d, e->wd, redo ? 1 : 0, e->baq_scratch.p, need, (int)c.max_lq, (int)c.max_bw, n_list);
            }
        }
    }
    if (has_ref && p->capQ_thres > 10) {
        // -C: after BAQ (it reads the adjusted qualities), before anything that looks at KEEP or the mapping quality
        for (int f = 0; f < nf; ++f) {
            ProfScope ps(e, "cap_mapq");
            sta_launch_cap_mapq(s, e->files_h[(size_t)f], e->wd, p->capQ_thres, p->min_mq, ctr);
        }
    }
    if (do_maxcnt) {
        // exact replay of the -d cap, one file at a time (rare path)
        for (int f = 0; f < nf; ++f) {
            StaReadsDev &d = e->files_h[(size_t)f];
            if (!d.n) continue;
            // span of read ends relative to the smallest start: computed on the host from the staged window bounds
            int32_t lo = e->min_pos[(size_t)f], hi = e->max_pos_hint[(size_t)f];
            int64_t span = (int64_t)hi - lo + 2;
            if (e->maxcnt_scratch.ensure((size_t)(span + 4) * 4 + 32)) return fail(e, STA_ERR_HIP, "hipMalloc(maxcnt) failed");
            if (e->scan_tmp.ensure(sta_scan_tmp_bytes(d.n) + 64)) return fail(e, STA_ERR_HIP, "hipMalloc failed");
            ProfScope ps(e, "maxcnt_serial");
            // (R.maxend = prefix maxima BEFORE the cap, from k_prep_reads: the replay's candidate bounds -- unless -C dropped reads since)
            if (has_ref && p->capQ_thres > 10) sta_launch_maxend_scan(s, d, e->scan_tmp.p, e->scan_tmp.cap);
            sta_launch_maxcnt(s, d, p->max_depth, lo, (int32_t)span, (int32_t *)e->maxcnt_scratch.p, ctr);
        }
    }
    // R.maxend comes out of k_prep_reads; only -C and the -d replay change the set of kept reads afterwards
    const bool rescan = do_maxcnt || (has_ref && p->capQ_thres > 10);
    for (int f = 0; f < nf && rescan; ++f) {
        StaReadsDev &d = e->files_h[(size_t)f];
        if (!d.n) continue;
        if (e->scan_tmp.ensure(sta_scan_tmp_bytes(d.n) + 64)) return fail(e, STA_ERR_HIP, "hipMalloc failed");
        ProfScope ps(e, "maxend_scan");
        sta_launch_maxend_scan(s, d, e->scan_tmp.p, e->scan_tmp.cap);
    }
    // the -d detector rides in the tile measuring kernel when that one runs
    const bool want_detect = !do_maxcnt && p->max_depth > 0 && p->max_depth < INT_MAX;
    const bool detect_in_len = want_detect && e->have_wfirst && ncols > 0;
    if (want_detect && !detect_in_len) {
        for (int f = 0; f < nf; ++f) {
            ProfScope ps(e, "maxcnt_detect");
            sta_launch_maxcnt_detect(s, e->files_h[(size_t)f], p->max_depth, ctr);
        }
    }
    if (olap && e->mate_fn) {
        // the caller keeps the overlap hash, and only this plan's kernels know who reached bam_plp_push: one round trip (sta_set_mate_resolver)
        bool repush = false;
        for (int f = 0; f < nf; ++f) {
            StaReadsDev &d = e->files_h[(size_t)f];
            if (!d.n) continue;
            std::vector<uint32_t> st((size_t)d.n);
            std::vector<int32_t> mate((size_t)d.n, -1);
            HIPCHK(hipMemcpyAsync(st.data(), d.info, (size_t)d.n * 4, hipMemcpyDeviceToHost, s));
            SYNC_S(s);
            if (e->mate_fn(e->mate_user, f, st.data(), d.n, mate.data()) != 0) return fail(e, STA_ERR_ARG, "the mate resolver failed");
            FileBufs &b = e->fb[(size_t)f];
            if (b.mate.ensure((size_t)d.n * 4 + 16)) return fail(e, STA_ERR_HIP, "hipMalloc failed");
            HIPCHK(hipMemcpyAsync(b.mate.p, mate.data(), (size_t)d.n * 4, hipMemcpyHostToDevice, s));
            SYNC_S(s);                                 // (`mate` goes out of scope)
            d.mate = (const int32_t *)b.mate.p;
            repush = true;
        }
        if (repush) { int rc2 = push_files(e); if (rc2) return rc2; }
    }
    if (olap) {
        for (int f = 0; f < nf; ++f) {
            StaReadsDev &d = e->files_h[(size_t)f];
            if (!d.n) continue;
            size_t slots = d.mate ? 1 : sta_overlap_table_slots(d.n);
            if (e->table.ensure(sta_overlap_table_bytes(slots) + 64)) return fail(e, STA_ERR_HIP, "hipMalloc(name table) failed");
            ProfScope ps(e, "overlap");
            sta_launch_overlap_setup(s, d, (StaReadsDev *)e->files_d.p + f, late_copy[(size_t)f] != 0, e->table.p, slots, ctr);
            sta_launch_overlap(s, d, e->wd.origin, e->wd.tid, e->table.p, slots, (int32_t *)e->fb[(size_t)f].chain.p, ctr);
        }
    }
    if (e->cov_mode) return STA_OK;
    if (e->plp_mode) {
        ProfScope ps(e, "plp_count");
        sta_launch_plp_count(s, e->wd, (uint32_t *)e->line_len.p);
    } else {
        if (e->colinfo.ensure((size_t)(ncols > 0 ? ncols : 1) * (size_t)(nf > 0 ? nf : 1) * 8 + 16)) return fail(e, STA_ERR_HIP, "hipMalloc(column info) failed");
        if (e->have_wfirst) {
            if (e->fused_status.ensure(sta_mplp_len_status_bytes(ncols) + 16)) return fail(e, STA_ERR_HIP, "hipMalloc(look-back words) failed");
            if (!wf_done) {                      // (every file empty: no preparation launch carried the index)
                ProfScope ps(e, "wave_first");
                sta_launch_wave_first(s, e->wd, (uint32_t *)e->wfirst.p, e->fused_status.p);
            }
        }
        // the generic walker's measuring pass keeps what it learns about every string of every row for its emit (k_mplp_len_x)
        // ... and so does the read-major pair when the window prints extra columns (sta_mplp_has_xfast_path)
        const bool xfast = sta_mplp_has_xfast_path(*p) && e->have_wfirst;
        const bool tile_path = sta_mplp_has_fast_path(*p) && e->have_wfirst && sta_mplp_tile_ok(*p);
        const int gx = tile_path ? -1 : xfast ? sta_mplp_xfast_extras(*p) : sta_mplp_generic_extras(*p);
        e->gen_xlen_on = false;
        if (gx >= 0) {
            if (e->gen_xlen.ensure((size_t)(ncols > 0 ? ncols : 1) * (size_t)(nf > 0 ? nf : 1) * (size_t)(gx > 0 ? gx : 1) * 4 + 16)) return fail(e, STA_ERR_HIP, "hipMalloc(extra-column lengths) failed");
            e->gen_xlen_on = true;
        }
        ProfScope ps(e, "mplp_len");
        e->len_fused = sta_launch_mplp_len(s, e->wd, *p, (uint32_t *)e->line_len.p, (uint2 *)e->colinfo.p, ctr, e->have_wfirst ? (const uint32_t *)e->wfirst.p : nullptr,
                                           e->have_wfirst ? e->fused_status.p : nullptr, (uint64_t *)e->offs.p, detect_in_len ? p->max_depth : 0,
                                           e->gen_xlen_on ? (uint32_t *)e->gen_xlen.p : nullptr);
    }
    return STA_OK;
}

// Restores the engine's pipeline mode and reference view on every exit path of a plan that borrows mpileup_pipeline()
// (the binary-entry, coverage and glf plans run it without a reference and stop it early): an early HIP-error return
// must not leave a pooled engine in the wrong mode.
struct ModeGuard {
    sta_engine *e; const char *ref; int64_t len;
    ModeGuard(sta_engine *e_, bool plp, bool cov) : e(e_), ref(e_->wd.ref), len(e_->wd.ref_len)
    {
        e->plp_mode = plp; e->cov_mode = cov;
        e->wd.ref = nullptr; e->wd.ref_len = 0;      // the plain iterator has no contig-length filter and no BAQ
    }
    ~ModeGuard() { e->plp_mode = false; e->cov_mode = false; e->wd.ref = ref; e->wd.ref_len = len; }
};

// coordinate span of the staged reads, for the exact -d replay (rare path: only after the detector fired)
static int maxcnt_bounds(sta_engine *e)
{
    for (size_t f = 0; f < e->files_h.size(); ++f) {
        StaReadsDev &d = e->files_h[f];
        if (!d.n) continue;
        int32_t first = 0, lastmax = 0, lastpos = 0;
        HIPCHK(hipMemcpy(&first, d.pos, 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(&lastmax, d.maxend + (d.n - 1), 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(&lastpos, d.pos + (d.n - 1), 4, hipMemcpyDeviceToHost));
        e->min_pos[f] = first;
        e->max_pos_hint[f] = lastmax > lastpos ? lastmax : lastpos;
    }
    return STA_OK;
}

// pipeline without the text measuring pass (cov_mode), re-run with the exact -d replay when the detector asks for it
static int counting_pipeline(sta_engine *e, const sta_mplp_params *p)
{
    int rc = mpileup_pipeline(e, p, false);
    if (rc) return rc;
    SYNC_STREAM();
    HIPCHK(hipMemcpy(&e->ctr_h, e->counters.p, sizeof(StaCounters), hipMemcpyDeviceToHost));
    if (!e->ctr_h.maxcnt_flag) return STA_OK;
    rc = maxcnt_bounds(e);
    if (rc) return rc;
    return mpileup_pipeline(e, p, true);
}

// totals of a single-pass kernel (k_depth_fused): counters -> host, one synchronisation
static int fused_finish(sta_engine *e, sta_plan_info *info)
{
    StaCounters *ctr_dst = e->pin ? (StaCounters *)e->pin : &e->ctr_h;
    HIPCHK(hipMemcpyAsync(ctr_dst, e->counters.p, sizeof(StaCounters), hipMemcpyDeviceToHost, e->stream));
    SYNC_STREAM();
    if (e->pin) e->ctr_h = *ctr_dst;
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return hipfail(e, le, "kernel launch");
    e->out_bytes = e->ctr_h.out_bytes;
    if (info) {
        info->out_bytes = e->out_bytes; info->n_lines = e->ctr_h.n_lines; info->n_data_cols = e->ctr_h.n_data_cols;
        info->n_kept_reads = e->ctr_h.n_kept; info->piled_bases = e->ctr_h.piled_bases; info->n_maxcnt_dropped = e->ctr_h.n_dropped;
    }
    return STA_OK;
}

int sta_mpileup_plan(sta_engine *e, const sta_mplp_params *p, sta_plan_info *info)
{
    if (!e || !p) return STA_ERR_ARG;
    if (!e->staged) return fail(e, STA_ERR_ARG, "no staged window");
    hipSetDevice(e->device);
    {
        int need_x = ((p->flag & STA_MPLP_PRINT_RNEXT) ? 1 : 0) + (p->n_tags > 0 ? p->n_tags : 0);
        for (auto &d : e->files_h)
            if (need_x && d.n && d.n_xcols != need_x) return fail(e, STA_ERR_ARG, "RNEXT / tag columns requested but sta_reads.xcol_* does not hold them");
    }
    e->mp = *p;
    int64_t ncols = (int64_t)e->wd.col_end - e->wd.col_beg;
    // host-side bounds for the (rare) exact -d replay
    e->min_pos.assign(e->files_h.size(), 0); e->max_pos_hint.assign(e->files_h.size(), 0);
    int rc = mpileup_pipeline(e, p, false);
    if (rc) return rc;
    rc = finish_plan(e, ncols, info);
    if (rc) return rc;
    if (e->ctr_h.maxcnt_flag) {
        // The cap may trigger somewhere in this window: find the coordinate span of the reads, then
        // re-run the pipeline with the exact replay inserted.
        rc = maxcnt_bounds(e);
        if (rc) return rc;
        rc = mpileup_pipeline(e, p, true);
        if (rc) return rc;
        rc = finish_plan(e, ncols, info);
        if (rc) return rc;
    }
    e->planned = 1;
    return STA_OK;
}

static int emit_common(sta_engine *e, void *dev_out, uint64_t capacity, char **out)
{
    if (dev_out) {
        if (capacity < e->out_bytes) return fail(e, STA_ERR_ARG, "output buffer too small");
        *out = (char *)dev_out;
    } else {
        if (e->out.ensure((size_t)e->out_bytes + 64)) return fail(e, STA_ERR_HIP, "hipMalloc(output) failed");
        *out = (char *)e->out.p;
    }
    e->last_out = *out;
    return STA_OK;
}

int sta_mpileup_emit(sta_engine *e, void *dev_out, uint64_t capacity)
{
    if (!e) return STA_ERR_ARG;
    if (e->planned != 1) return fail(e, STA_ERR_ARG, "sta_mpileup_plan has not run for this window");
    hipSetDevice(e->device);
    char *out = nullptr;
    int rc = emit_common(e, dev_out, capacity, &out);
    if (rc) return rc;
    if (e->out_bytes == 0) return STA_OK;
    // Which kernel writes which columns (windows without extra columns):
    //  * deep windows (mean depth of the data columns >= 100): every strip through the read-major kernel k_mplp_emit_deep;
    //  * otherwise k_mplp_emit_tile with an LDS text slice that holds the largest wave's 64 rows -- up to 12 KiB; beyond that (a
    //    deep amplicon inside an ordinary window) the slice is 1.5 x the mean bytes of a wave's rows and the 64-column groups whose
    //    rows exceed it go through k_mplp_emit_deep (a second launch over the strips, most of which return at once: ~0.06 ms).
    // STA_EMIT_DEEP=0 / 1 forces the whole-window choice (tests).
    const int64_t ncols = (int64_t)e->wd.col_end - e->wd.col_beg;
    bool deep = e->ctr_h.n_data_cols > 0 && e->ctr_h.piled_bases / e->ctr_h.n_data_cols >= 100;
    if (const char *ev = getenv("STA_EMIT_DEEP")) deep = atoi(ev) != 0;
    const uint64_t mw = e->ctr_h.max_wave_bytes;
    static const uint64_t cap_pct = [] { const char *ev = getenv("STA_TILE_CAP_PCT"); const int v = ev ? atoi(ev) : 150; return (uint64_t)(v < 100 ? 100 : v > 400 ? 400 : v); }();      // experiment knob
    uint64_t tc = (mw + 255) & ~255ull;
    if (tc > 12288 || getenv("STA_TILE_CAP_PCT")) {
        tc = e->ctr_h.n_data_cols ? (e->out_bytes * 64 * cap_pct / 100) / e->ctr_h.n_data_cols : 1024;
        tc = (tc + 255) & ~255ull;
        if (tc > 12288) tc = 12288;
        if (tc > ((mw + 255) & ~255ull)) tc = (mw + 255) & ~255ull;
    }
    if (tc < 1024) tc = 1024;
    const uint32_t tile_cap = (uint32_t)tc;
    int deep_mode = deep ? 1 : 0;
    if (!e->len_fused) deep_mode = 0;                     // the generic walker writes every row itself
    else if (sta_mplp_has_xfast_path(e->mp)) deep_mode = 1;      // extra columns: the read-major kernel at every depth
    else if (!deep && mw > tile_cap) deep_mode = 2;
    if (deep_mode) {
        if (e->strip_rng.ensure((size_t)sta_mplp_deep_strips(ncols > 0 ? ncols : 1) * (size_t)(e->wd.nfiles > 0 ? e->wd.nfiles : 1) * 16 + 16))
            return fail(e, STA_ERR_HIP, "hipMalloc(strip ranges) failed");
    }
    ProfScope ps(e, deep_mode == 1 ? "mplp_emit_deep" : "mplp_emit");
    sta_launch_mplp_emit(e->stream, e->wd, e->mp, (const uint64_t *)e->offs.p, (const uint2 *)e->colinfo.p, out, e->lds_cap, deep_mode ? (int64_t *)e->strip_rng.p : nullptr,
                         tile_cap, deep_mode, e->have_wfirst ? (const uint32_t *)e->wfirst.p : nullptr,
                         e->len_fused ? sta_mplp_tile_base(e->fused_status.p, ncols) : nullptr, e->len_fused,
                         e->gen_xlen_on ? (const uint32_t *)e->gen_xlen.p : nullptr);
    return STA_OK;
}

/* plan + emit in one call: the whole window's text into dev_out (NULL: engine buffer, read with sta_fetch_output) */
int sta_mpileup_run(sta_engine *e, const sta_mplp_params *p, void *dev_out, uint64_t capacity, sta_plan_info *info)
{
    sta_plan_info tmp;
    int rc = sta_mpileup_plan(e, p, info ? info : &tmp);
    if (rc) return rc;
    return sta_mpileup_emit(e, dev_out, capacity);
}

/* ---- binary per-column entries (bam_plp_* surface) ---- */
int sta_plp_plan(sta_engine *e, int32_t max_depth, int32_t overlaps, sta_plan_info *info)
{
    if (!e) return STA_ERR_ARG;
    if (!e->staged) return fail(e, STA_ERR_ARG, "no staged window");
    if (e->files_h.size() != 1) return fail(e, STA_ERR_ARG, "the pileup-entry path takes one input file per window");
    hipSetDevice(e->device);
    sta_mplp_params p; memset(&p, 0, sizeof p);
    p.max_depth = max_depth;
    p.flag = overlaps ? STA_MPLP_SMART_OVERLAPS : 0;     // bam_plp_push drops unmapped reads only; callers filter in their callback
    e->mp = p;
    int64_t ncols = (int64_t)e->wd.col_end - e->wd.col_beg;
    e->min_pos.assign(e->files_h.size(), 0); e->max_pos_hint.assign(e->files_h.size(), 0);
    int rc;
    {
        // no reference here: one set for another use of this engine must not trigger the "read beyond the FASTA" filter
        ModeGuard guard(e, true, false);
        rc = mpileup_pipeline(e, &p, false);
        if (!rc) rc = finish_plan(e, ncols, info);
        if (!rc && e->ctr_h.maxcnt_flag) {
            rc = maxcnt_bounds(e);
            if (!rc) rc = mpileup_pipeline(e, &p, true);
            if (!rc) rc = finish_plan(e, ncols, info);
        }
    }
    if (rc) return rc;
    e->out_bytes *= 16;                                   // offsets were scanned in entries
    if (info) info->out_bytes = e->out_bytes;
    e->planned = 3;
    return STA_OK;
}

int sta_plp_emit(sta_engine *e, void *dev_entries, uint64_t capacity)
{
    if (!e) return STA_ERR_ARG;
    if (e->planned != 3) return fail(e, STA_ERR_ARG, "sta_plp_plan has not run for this window");
    hipSetDevice(e->device);
    char *out = nullptr;
    int rc = emit_common(e, dev_entries, capacity, &out);
    if (rc) return rc;
    if (e->out_bytes == 0) return STA_OK;
    ProfScope ps(e, "plp_fill");
    sta_launch_plp_fill(e->stream, e->wd, (const uint64_t *)e->offs.p, out);
    return STA_OK;
}

uint64_t sta_stage_raw_reads(sta_engine *e) { return e ? e->n_raw_staged : 0; }

int sta_fetch_col_offsets(sta_engine *e, uint64_t *host_offs, uint64_t n)
{
    if (!e || !host_offs) return STA_ERR_ARG;
    if (!e->planned) return fail(e, STA_ERR_ARG, "no planned window");
    hipSetDevice(e->device);
    uint64_t ncols = (uint64_t)((int64_t)e->wd.col_end - e->wd.col_beg);
    if (n > ncols + 1) return fail(e, STA_ERR_ARG, "more offsets requested than columns + 1");
    HIPCHK(hipMemcpyAsync(host_offs, e->offs.p, (size_t)n * 8, hipMemcpyDeviceToHost, e->stream));
    if (e->len_fused) {
        // the tile path keeps row offsets relative to each column's measuring tile of 1 024 columns, with the tiles' bases (and, behind
        // them, the window total) in the look-back buffer: what the header promises are offsets inside the whole window
        const uint64_t tiles = (ncols + 1023) / 1024;
        std::vector<uint64_t> tb((size_t)tiles + 1);
        HIPCHK(hipMemcpyAsync(tb.data(), sta_mplp_tile_base(e->fused_status.p, ncols), (size_t)(tiles + 1) * 8, hipMemcpyDeviceToHost, e->stream));
        SYNC_STREAM();
        for (uint64_t c = 0; c < n; ++c) {
            if (c == ncols) host_offs[c] = tb[(size_t)tiles];          // the window's total bytes
            else host_offs[c] += tb[(size_t)(c >> 10)];
        }
        return STA_OK;
    }
    SYNC_STREAM();
    return STA_OK;
}

int sta_set_mate_resolver(sta_engine *e, sta_mate_resolver fn, void *user)
{
    if (!e) return STA_ERR_ARG;
    e->mate_fn = fn; e->mate_user = fn ? user : nullptr;
    return STA_OK;
}

int sta_fetch_read_state(sta_engine *e, int32_t file, uint32_t *host_info, uint8_t *host_qual)
{
    if (!e || file < 0 || (size_t)file >= e->files_h.size()) return STA_ERR_ARG;
    if (!e->planned) return fail(e, STA_ERR_ARG, "no planned window");
    hipSetDevice(e->device);
    StaReadsDev &d = e->files_h[(size_t)file];
    if (host_info && d.n) HIPCHK(hipMemcpyAsync(host_info, d.info, (size_t)d.n * 4, hipMemcpyDeviceToHost, e->stream));
    // (a working pool left to k_olap_setup was never written when the window had no eligible read: the qualities in use are the input's)
    const bool unused_pool = (size_t)file < e->late_copy.size() && e->late_copy[(size_t)file] && e->ctr_h.n_olap_el == 0;
    if (host_qual && d.n_bases_total) HIPCHK(hipMemcpyAsync(host_qual, unused_pool ? d.qual_in : d.qual, (size_t)d.n_bases_total, hipMemcpyDeviceToHost, e->stream));
    SYNC_STREAM();
    return STA_OK;
}

int sta_cov_plan(sta_engine *e, const sta_cov_params *cp, sta_cov_totals *totals, uint64_t *per_file, sta_plan_info *info)
{
    if (!e || !cp) return STA_ERR_ARG;
    if (!e->staged) return fail(e, STA_ERR_ARG, "no staged window");
    hipSetDevice(e->device);
    sta_mplp_params p; memset(&p, 0, sizeof p);
    p.max_depth = cp->max_depth; p.min_mq = cp->min_mq; p.rflag_require = cp->rflag_require; p.rflag_filter = cp->rflag_filter; p.min_qlen = cp->min_qlen;
    e->mp = p;
    e->min_pos.assign(e->files_h.size(), 0); e->max_pos_hint.assign(e->files_h.size(), 0);
    const size_t nf = e->files_h.size();
    if (e->cov_out.ensure((5 + nf * 2) * 8 + 64)) return fail(e, STA_ERR_HIP, "hipMalloc failed");
    {
        ModeGuard guard(e, false, true);
        int rc = counting_pipeline(e, &p);
        if (rc) return rc;
    }
    HIPCHK(hipMemsetAsync(e->cov_out.p, 0, (5 + nf * 2) * 8, e->stream));
    {
        ProfScope ps(e, "cov_cols");
        sta_launch_cov_cols(e->stream, e->wd, cp->mode, cp->min_baseQ, cp->min_depth, cp->skip_dn, (unsigned long long *)e->cov_out.p,
                            (unsigned long long *)e->cov_out.p + 5,
                            cp->hist_bins > 0 && e->cov_hist_bins >= cp->hist_bins ? (uint32_t *)e->cov_hist.p : nullptr, cp->hist_bins, cp->hist_depth,
                            cp->hist_beg, cp->hist_bin_width);
    }
    std::vector<uint64_t> host(5 + nf * 2);
    HIPCHK(hipMemcpyAsync(host.data(), e->cov_out.p, host.size() * 8, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemcpyAsync(&e->ctr_h, e->counters.p, sizeof(StaCounters), hipMemcpyDeviceToHost, e->stream));
    SYNC_STREAM();
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return hipfail(e, le, "coverage kernels");
    if (totals) { totals->n_covered_bases = host[0]; totals->summed_coverage = host[1]; totals->summed_baseQ = host[2]; totals->quality_bases = host[3]; totals->missing_qual = host[4]; }
    if (per_file) for (size_t i = 0; i < nf * 2; ++i) per_file[i] = host[5 + i];
    if (info) { memset(info, 0, sizeof *info); info->n_kept_reads = e->ctr_h.n_kept; info->piled_bases = e->ctr_h.piled_bases; info->n_maxcnt_dropped = e->ctr_h.n_dropped; }
    e->planned = 4;      // read state can be fetched; there is nothing to emit
    return STA_OK;
}

// coverage -m / -D: the per-contig histogram sta_cov_plan adds to (coverage.c:588, :599 memset between contigs)
int sta_cov_hist_begin(sta_engine *e, int32_t n_bins)
{
    if (!e || n_bins <= 0) return STA_ERR_ARG;
    hipSetDevice(e->device);
    if (e->cov_hist.ensure((size_t)n_bins * 4 + 64)) return fail(e, STA_ERR_HIP, "hipMalloc failed");
    HIPCHK(hipMemsetAsync(e->cov_hist.p, 0, (size_t)n_bins * 4, e->stream));
    SYNC_STREAM();
    e->cov_hist_bins = n_bins;
    return STA_OK;
}

int sta_cov_hist_fetch(sta_engine *e, uint32_t *hist, int32_t n_bins)
{
    if (!e || !hist || n_bins <= 0) return STA_ERR_ARG;
    if (n_bins > e->cov_hist_bins) return fail(e, STA_ERR_ARG, "histogram was opened with fewer bins");
    hipSetDevice(e->device);
    HIPCHK(hipMemcpyAsync(hist, e->cov_hist.p, (size_t)n_bins * 4, hipMemcpyDeviceToHost, e->stream));
    SYNC_STREAM();
    return STA_OK;
}

// `stats` coverage distribution (kernels_statcov.hip): the bins live on the device from begin to fetch
int sta_statcov_begin(sta_engine *e, const sta_statcov_params *p, int32_t *ncov_out)
{
    if (!e || !p || p->cov_step <= 0 || p->cov_max < p->cov_min) return STA_ERR_ARG;
    hipSetDevice(e->device);
    const int32_t ncov = 3 + (p->cov_max - p->cov_min) / p->cov_step;      // stats.c:2404 (the caller has applied :2398-2405 to the triple)
    if (e->sc_cov.ensure((size_t)ncov * 8 + 64)) return fail(e, STA_ERR_HIP, "hipMalloc failed");
    HIPCHK(hipMemsetAsync(e->sc_cov.p, 0, (size_t)ncov * 8, e->stream));
    SYNC_STREAM();
    e->sc = *p; e->sc_ncov = ncov;
    if (ncov_out) *ncov_out = ncov;
    return STA_OK;
}

int sta_statcov_add(sta_engine *e, const int64_t *pos, const int32_t *delta, int64_t n, int64_t carry_in, int mem)
{
    if (!e || !pos || !delta || n < 0) return STA_ERR_ARG;
    if (!e->sc_ncov) return fail(e, STA_ERR_ARG, "sta_statcov_begin has not been called");
    if (n < 2) return STA_OK;
    hipSetDevice(e->device);
    hipStream_t s = e->stream;
    const int64_t *dpos = pos; const int32_t *ddelta = delta;
    if (mem == STA_MEM_HOST) {
        if (e->sc_pos.ensure((size_t)n * 8 + 64) || e->sc_delta.ensure((size_t)n * 4 + 64)) return fail(e, STA_ERR_HIP, "hipMalloc failed");
        HIPCHK(hipMemcpyAsync(e->sc_pos.p, pos, (size_t)n * 8, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(e->sc_delta.p, delta, (size_t)n * 4, hipMemcpyHostToDevice, s));
        dpos = (const int64_t *)e->sc_pos.p; ddelta = (const int32_t *)e->sc_delta.p;
    }
    if (e->sc_tmp.ensure(sta_statcov_tmp_bytes(n) + 64)) return fail(e, STA_ERR_HIP, "hipMalloc failed");
    {
        ProfScope ps(e, "statcov");
        sta_launch_statcov(s, dpos, ddelta, n, (long long)carry_in, e->sc.cov_min, e->sc.cov_max, e->sc.cov_step, e->sc_ncov, (unsigned long long *)e->sc_cov.p, e->sc_tmp.p);
    }
    SYNC_S(s);         // the host buffers may be reused by the caller
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return hipfail(e, le, "stats coverage kernels");
    return STA_OK;
}

int sta_statcov_fetch(sta_engine *e, uint64_t *cov, int32_t ncov)
{
    if (!e || !cov || ncov <= 0) return STA_ERR_ARG;
    if (ncov != e->sc_ncov) return fail(e, STA_ERR_ARG, "bin count differs from the open distribution");
    hipSetDevice(e->device);
    HIPCHK(hipMemcpyAsync(cov, e->sc_cov.p, (size_t)ncov * 8, hipMemcpyDeviceToHost, e->stream));
    SYNC_STREAM();
    return STA_OK;
}

// row a14: bcf_call_glfgen over every column of the staged window (kernels_glf.hip)
int sta_glf_plan(sta_engine *e, const sta_glf_params *gp, sta_plan_info *info)
{
    if (!e || !gp) return STA_ERR_ARG;
    if (!e->staged) return fail(e, STA_ERR_ARG, "no staged window");
    hipSetDevice(e->device);
    const double theta = gp->theta <= 0. ? 0.83 : gp->theta;
    const double depcorr = 1. - theta;
    if (e->glf_depcorr != depcorr) {
        std::vector<double> t;
        sta::glf_tables(depcorr, t);
        if (e->glf_tab.ensure(t.size() * 8)) return fail(e, STA_ERR_HIP, "hipMalloc(glf tables) failed");
        HIPCHK(hipMemcpy(e->glf_tab.p, t.data(), t.size() * 8, hipMemcpyHostToDevice));
        e->glf_depcorr = depcorr;
    }
    sta_mplp_params p; memset(&p, 0, sizeof p);
    p.max_depth = gp->max_depth > 0 ? gp->max_depth : 8000;
    e->mp = p;
    const char *saved_ref = e->wd.ref; const int64_t saved_len = e->wd.ref_len;     // the column's reference base for k_glf_cols
    e->min_pos.assign(e->files_h.size(), 0); e->max_pos_hint.assign(e->files_h.size(), 0);
    const size_t nf = e->files_h.size();
    {
        ModeGuard guard(e, false, true);
        int rc = counting_pipeline(e, &p);
        if (rc) return rc;
    }
    const int64_t ncols = (int64_t)e->wd.col_end - e->wd.col_beg;
    const uint64_t bytes = (uint64_t)(ncols > 0 ? ncols : 0) * nf * sizeof(sta_glf_col);
    if (e->out.ensure(bytes + 64)) return fail(e, STA_ERR_HIP, "hipMalloc(output) failed");
    {
        ProfScope ps(e, "glf_cols");
        const double *t = (const double *)e->glf_tab.p;
        double mean_depth = 0;
        for (int f = 0; f < nf; ++f) mean_depth += (double)e->files_h[(size_t)f].n_bases_total;
        mean_depth = ncols > 0 ? mean_depth / (double)ncols / (double)(nf > 0 ? nf : 1) : 0;
        if (e->glf_redo.ensure(sta_glf_redo_bytes(e->wd) + 64)) return fail(e, STA_ERR_HIP, "hipMalloc(glf groups) failed");
        sta_launch_glf_cols(e->stream, e->wd, gp->min_baseQ, 60, saved_ref, saved_len, t + sta::GLF_FK_OFF, t + sta::GLF_BETA_OFF, t + sta::GLF_LHET_OFF, e->out.p, (uint8_t *)e->glf_redo.p, mean_depth);
    }
    HIPCHK(hipMemcpyAsync(&e->ctr_h, e->counters.p, sizeof(StaCounters), hipMemcpyDeviceToHost, e->stream));
    SYNC_STREAM();
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return hipfail(e, le, "glf kernels");
    e->out_bytes = bytes; e->last_out = e->out.p;
    if (info) { memset(info, 0, sizeof *info); info->out_bytes = bytes; info->n_kept_reads = e->ctr_h.n_kept; info->piled_bases = e->ctr_h.piled_bases; info->n_maxcnt_dropped = e->ctr_h.n_dropped; }
    e->planned = 4;
    return STA_OK;
}

// `samtools consensus` on file 0 of the staged window (kernels_cons.hip; steps in cons_window.h)
static int cons_run(sta_engine *e, const sta_cons_params *cp, sta_cons_info *info, bool walk_all);

int sta_consensus_run(sta_engine *e, const sta_cons_params *cp, sta_cons_info *info) { return cons_run(e, cp, info, false); }

// the iterator half only (pileup_loop / get_next_base): every read's entries, no callers.  The caller's seq_fetch has done the
// filtering, so only the unmapped flag is tested (consensus_pileup.c:341-349).
int sta_cons_entries_run(sta_engine *e, sta_cons_info *info)
{
    sta_cons_params p; memset(&p, 0, sizeof p);
    p.mode = STA_CONS_SIMPLE;
    return cons_run(e, &p, info, true);
}

static int cons_run(sta_engine *e, const sta_cons_params *cp, sta_cons_info *info, bool walk_all)
{
    if (!e || !cp) return STA_ERR_ARG;
    if (!e->staged) return fail(e, STA_ERR_ARG, "no staged window");
    if (e->files_h.size() != 1) return fail(e, STA_ERR_ARG, "consensus works on one input file");
    if (cp->mode < STA_CONS_SIMPLE || cp->mode > STA_CONS_MIXED) return fail(e, STA_ERR_ARG, "unknown consensus mode");
    hipSetDevice(e->device);
    hipStream_t s = e->stream;
    // lookup tables: rebuilt only when a parameter they depend on changed
    {
        sta_cons_params a = *cp, b = e->cons_p; a.want_pileup = b.want_pileup = 0;
        if (!e->cons_tab_ok || memcmp(&a, &b, sizeof a) != 0) {
            std::unique_ptr<cons::Tables> t(new cons::Tables);
            sta::cons_build_tables(*cp, *t);
            if (e->cons_tab.ensure(sizeof(cons::Tables))) return fail(e, STA_ERR_HIP, "hipMalloc(consensus tables) failed");
            HIPCHK(hipMemcpy(e->cons_tab.p, t.get(), sizeof(cons::Tables), hipMemcpyHostToDevice));
            e->cons_p = *cp; e->cons_tab_ok = true;
        }
    }
    const cons::Par o = sta::cons_par(*cp);
    const cons::Tables *tab = (const cons::Tables *)e->cons_tab.p;
    const StaReadsDev &d = e->files_h[0];
    const int64_t n = d.n;
    const int64_t W = (int64_t)e->wd.col_end - e->wd.col_beg;
    if (W < 0) return fail(e, STA_ERR_ARG, "empty window");
    const bool bayes_mq = o.mode != cons::MODE_SIMPLE && o.use_mqual;
    if (bayes_mq && d.n_xcols < 1 && n > 0) return fail(e, STA_ERR_ARG, "the Bayesian mode reads MD:Z from text column 0 of sta_reads.xcol_*");
    // workspace carve-up (one allocation): per-position and per-read arrays
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t at = off; off += (bytes + 255) & ~(size_t)255; return at; };
    const size_t o_ins = carve((size_t)(W + 1) * 4), o_len = carve((size_t)(W + 1) * 4), o_colbase = carve((size_t)(W + 2) * 8);
    const size_t o_last = carve((size_t)n * 4), o_tail = carve((size_t)n * 4), o_keep = carve((size_t)n * 4), o_cs = carve((size_t)n * 4),
                 o_ce = carve((size_t)n * 4), o_pmax = carve((size_t)n * 4), o_cnt = carve((size_t)n * 4), o_rowoff = carve((size_t)(n + 2) * 8), o_clist = carve((size_t)n * 4), o_meta = carve((size_t)(n + 1) * sizeof(cons::Meta)), o_ctr = carve(64);
    if (e->cons_ws.ensure(off + 256) || e->scan_tmp.ensure(sta_scan_tmp_bytes(std::max<int64_t>(std::max(W, n), 1) * 2) + 64)) return fail(e, STA_ERR_HIP, "hipMalloc(consensus workspace) failed");
    char *ws = (char *)e->cons_ws.p;
    cons::Win w; memset(&w, 0, sizeof w);
    w.n_reads = n; w.pos = d.pos; w.flag = d.flag; w.mapq = d.mapq; w.l_qseq = d.l_qseq; w.cig_off = d.cig_off; w.base_off8 = d.base_off8;
    w.cigar = d.cigar; w.seq = d.seq; w.qual_in = d.qual_in; w.n_xcols = d.n_xcols; w.xcol_off = d.xcol_off; w.xcol_text = d.xcol_text;
    w.col_beg = e->wd.col_beg; w.col_end = e->wd.col_end;
    w.ins = (uint32_t *)(ws + o_ins); w.colbase = (uint64_t *)(ws + o_colbase);
    w.r_last = (int32_t *)(ws + o_last); w.r_tail = (int32_t *)(ws + o_tail); w.r_keep = (uint32_t *)(ws + o_keep);
    w.cs = (int32_t *)(ws + o_cs); w.ce = (int32_t *)(ws + o_ce); w.pmax = (int32_t *)(ws + o_pmax); w.cnt = (uint32_t *)(ws + o_cnt);
    w.rowoff = (uint64_t *)(ws + o_rowoff); w.clist = (int32_t *)(ws + o_clist); w.meta = (cons::Meta *)(ws + o_meta); w.counters = (unsigned long long *)(ws + o_ctr);
    uint32_t *collen = (uint32_t *)(ws + o_len);
    w.qual = const_cast<uint8_t *>(d.qual_in);
    if (bayes_mq) {
        const size_t nb = (size_t)d.n_bases_total;
        if (e->cons_qwork.ensure(nb + 64) || e->cons_nm.ensure((nb + 8) * 4 + 64) || e->cons_gran.ensure((nb / 8 + 2) * 4 + 64)) return fail(e, STA_ERR_HIP, "hipMalloc(consensus per-base workspace) failed");
        if (nb) HIPCHK(hipMemcpyAsync(e->cons_qwork.p, d.qual_in, nb, hipMemcpyDeviceToDevice, s));
        w.qual = (uint8_t *)e->cons_qwork.p; w.nm = (int32_t *)e->cons_nm.p;
    }
    HIPCHK(hipMemsetAsync(w.ins, 0, (size_t)(W + 1) * 4, s));
    HIPCHK(hipMemsetAsync(w.counters, 0, 64, s));
    { ProfScope ps(e, "cons_read_a"); sta_launch_cons_read_a(s, w, o, tab, walk_all); }
    if (bayes_mq) { ProfScope ps(e, "cons_prepare"); sta_launch_cons_prepare(s, w, o, tab, (int32_t *)e->cons_gran.p, (int64_t)d.n_bases_total); }
    { ProfScope ps(e, "cons_scans");
      sta_launch_cons_collen(s, w.ins, collen, W);
      sta_launch_len_scan(s, collen, w.colbase, W, e->scan_tmp.p, e->scan_tmp.cap); }
    { ProfScope ps(e, "cons_read_b"); sta_launch_cons_read_b(s, w); }
    { ProfScope ps(e, "cons_scans");
      sta_launch_len_scan(s, w.cnt, w.rowoff, n, e->scan_tmp.p, e->scan_tmp.cap);
      if (n > 0) sta_launch_scan_max_i32(s, w.ce, w.pmax, n, e->scan_tmp.p); }
    uint64_t n_entries = 0, n_cols = 0; unsigned long long ctr[4] = { 0, 0, 0, 0 };
    HIPCHK(hipMemcpyAsync(&n_entries, w.rowoff + n, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&n_cols, w.colbase + W, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(ctr, w.counters, 32, hipMemcpyDeviceToHost, s));
    SYNC_S(s);
    if (ctr[1]) return fail(e, STA_ERR_ARG, "a CIGAR holds an operation outside MIDNSHP=X");
    if (n_cols > (uint64_t)INT32_MAX - 64 || n_entries > ((uint64_t)1 << 40)) return fail(e, STA_ERR_ARG, "consensus window too large: column indices are 32-bit (split the window)");
    if (e->cons_E.ensure((size_t)n_entries * 4 + 64) || ((bayes_mq || walk_all) && e->cons_Enm.ensure((size_t)n_entries * 4 + 64)) || e->cons_cols.ensure((size_t)n_cols * sizeof(sta_cons_col) + 64)
        || e->cons_depth.ensure((size_t)n_cols * 4 + 64) || e->cons_colpos.ensure((size_t)n_cols * 4 + 64))
        return fail(e, STA_ERR_HIP, "hipMalloc(consensus entries) failed");
    w.colpos = (int32_t *)e->cons_colpos.p;
    const uint64_t sum_depth = ctr[3];
    w.E = (uint32_t *)e->cons_E.p; w.Enm = bayes_mq || walk_all ? (uint32_t *)e->cons_Enm.p : nullptr;
    w.cols = (sta_cons_col *)e->cons_cols.p; w.depth = (uint32_t *)e->cons_depth.p;
    { ProfScope ps(e, "cons_colpos"); sta_launch_cons_colpos(s, w); }
    { ProfScope ps(e, "cons_walk"); sta_launch_cons_walk(s, w, o, (int64_t)ctr[2], walk_all); }
    if (!walk_all) { ProfScope ps(e, "cons_col"); sta_launch_cons_col(s, w, o, tab, (int64_t)n_cols); }
    e->cons_text = cp->want_pileup != 0 && !walk_all;
    if (e->cons_text) {
        if (e->cons_coloff.ensure((size_t)(n_cols + 2) * 8 + 64) || e->cons_seq.ensure((size_t)sum_depth + 64) || e->cons_qual.ensure((size_t)sum_depth + 64)
            || e->scan_tmp.ensure(sta_scan_tmp_bytes((int64_t)n_cols + 1) + 64))
            return fail(e, STA_ERR_HIP, "hipMalloc(consensus text) failed");
        w.col_off = (uint64_t *)e->cons_coloff.p; w.seq_chars = (char *)e->cons_seq.p; w.qual_chars = (char *)e->cons_qual.p;
        { ProfScope ps(e, "cons_scans"); sta_launch_len_scan(s, w.depth, w.col_off, (int64_t)n_cols, e->scan_tmp.p, e->scan_tmp.cap); }
        { ProfScope ps(e, "cons_text"); sta_launch_cons_text(s, w, o, (int64_t)n_cols); }
    }
    SYNC_S(s);
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return hipfail(e, le, "consensus kernels");
    prof_drain(e);
    e->cons_w = w; e->cons_ncols = n_cols; e->cons_nentries = sum_depth; e->cons_W = W; e->cons_walk_all = walk_all; e->cons_nstored = n_entries;
    if (info) { info->n_cols = n_cols; info->n_entries = sum_depth; info->n_kept_reads = ctr[0]; }
    e->planned = walk_all ? 7 : 8;
    return STA_OK;
}

int sta_fetch_cons_entries(sta_engine *e, int32_t *ins, int32_t *first_col, int32_t *last_col, uint64_t *entry_off, uint32_t *entries, uint32_t *seq_offs)
{
    if (!e) return STA_ERR_ARG;
    if (e->planned != 7) return fail(e, STA_ERR_ARG, "no consensus entries were computed");
    hipSetDevice(e->device);
    hipStream_t s = e->stream;
    const cons::Win &w = e->cons_w;
    const size_t n = (size_t)w.n_reads;
    if (ins && e->cons_W) HIPCHK(hipMemcpyAsync(ins, w.ins + 1, (size_t)e->cons_W * 4, hipMemcpyDeviceToHost, s));
    if (first_col && n) HIPCHK(hipMemcpyAsync(first_col, w.cs, n * 4, hipMemcpyDeviceToHost, s));
    if (last_col && n) HIPCHK(hipMemcpyAsync(last_col, w.ce, n * 4, hipMemcpyDeviceToHost, s));
    if (entry_off) HIPCHK(hipMemcpyAsync(entry_off, w.rowoff, (n + 1) * 8, hipMemcpyDeviceToHost, s));
    if (entries && e->cons_nstored) HIPCHK(hipMemcpyAsync(entries, w.E, (size_t)e->cons_nstored * 4, hipMemcpyDeviceToHost, s));
    if (seq_offs && e->cons_nstored) HIPCHK(hipMemcpyAsync(seq_offs, w.Enm, (size_t)e->cons_nstored * 4, hipMemcpyDeviceToHost, s));
    SYNC_S(s);
    return STA_OK;
}

int sta_fetch_consensus(sta_engine *e, int32_t *ins, sta_cons_col *cols, uint64_t *col_off, char *seq_chars, char *qual_chars)
{
    if (!e) return STA_ERR_ARG;
    if (e->planned != 8) return fail(e, STA_ERR_ARG, "no consensus window was run");
    if ((col_off || seq_chars || qual_chars) && !e->cons_text) return fail(e, STA_ERR_ARG, "the window was run without want_pileup");
    hipSetDevice(e->device);
    hipStream_t s = e->stream;
    const cons::Win &w = e->cons_w;
    if (ins && e->cons_W) HIPCHK(hipMemcpyAsync(ins, w.ins + 1, (size_t)e->cons_W * 4, hipMemcpyDeviceToHost, s));
    if (cols && e->cons_ncols) HIPCHK(hipMemcpyAsync(cols, w.cols, (size_t)e->cons_ncols * sizeof(sta_cons_col), hipMemcpyDeviceToHost, s));
    if (col_off) HIPCHK(hipMemcpyAsync(col_off, w.col_off, (size_t)(e->cons_ncols + 1) * 8, hipMemcpyDeviceToHost, s));
    if (seq_chars && e->cons_nentries) HIPCHK(hipMemcpyAsync(seq_chars, w.seq_chars, (size_t)e->cons_nentries, hipMemcpyDeviceToHost, s));
    if (qual_chars && e->cons_nentries) HIPCHK(hipMemcpyAsync(qual_chars, w.qual_chars, (size_t)e->cons_nentries, hipMemcpyDeviceToHost, s));
    SYNC_S(s);
    return STA_OK;
}

// 8(f) row 3: calmd's per-record arithmetic (kernels_md.hip) on file 0 of the staged window
int sta_calmd_plan(sta_engine *e, const sta_calmd_params *cp, sta_plan_info *info)
{
    if (!e || !cp) return STA_ERR_ARG;
    if (!e->staged) return fail(e, STA_ERR_ARG, "no staged window");
    if (e->files_h.size() != 1) return fail(e, STA_ERR_ARG, "calmd works on one input file");
    hipSetDevice(e->device);
    const bool realn = (cp->flag & STA_CALMD_REALN) != 0, apply = (cp->flag & STA_CALMD_APPLY) != 0;
    if (realn && !e->wd.ref) return fail(e, STA_ERR_ARG, "calmd -r needs the reference of the contig");
    sta_mplp_params p; memset(&p, 0, sizeof p);
    p.flag = (realn ? STA_MPLP_REALN : 0) | STA_MPLP_INT_CALMD;
    e->mp = p;
    e->min_pos.assign(1, 0); e->max_pos_hint.assign(1, 0);
    StaReadsDev &d = e->files_h[0];
    const uint8_t *saved_bq = d.bq;
    if (!apply) d.bq = nullptr;                  // without -A an existing BQ:Z is left alone (and still blocks a recomputation)
    e->wd.baq_plain = (cp->flag & STA_CALMD_EXTENDED) ? 0 : 1;
    e->cov_mode = true;
    int rc = mpileup_pipeline(e, &p, false);
    e->cov_mode = false;
    e->wd.baq_plain = 0;
    if (rc) { d.bq = saved_bq; return rc; }
    hipStream_t s = e->stream;
    const size_t n = (size_t)d.n, nb = (size_t)d.n_bases_total;
    if (e->md_nm.ensure(n * 4 + 16) || e->md_len.ensure(n * 4 + 16) || e->md_state.ensure(n + 16) || e->md_tag.ensure(nb + 16)
        || e->md_seq.ensure(nb / 2 + 16) || e->offs.ensure((n + 2) * 8 + 16) || e->scan_tmp.ensure(sta_scan_tmp_bytes((int64_t)n) + 64))
        return fail(e, STA_ERR_HIP, "hipMalloc failed");
    HIPCHK(hipMemsetAsync(e->md_state.p, 0, n + 1, s));
    HIPCHK(hipMemsetAsync(e->md_tag.p, 0, nb + 1, s));
    if (nb) HIPCHK(hipMemcpyAsync(e->md_seq.p, d.seq, nb / 2, hipMemcpyDeviceToDevice, s));
    if (!realn && d.qual == d.qual_in && ((cp->flag & STA_CALMD_BIN_QUAL) || cp->max_nm > 0) && nb) {
        // -q / -n rewrite qualities: they need the working copy the BAQ path would have made
        FileBufs &b = e->fb[0];
        if (b.qual_work.ensure(nb + 32)) return fail(e, STA_ERR_HIP, "hipMalloc(qual) failed");
        HIPCHK(hipMemcpyAsync(b.qual_work.p, d.qual_in, nb, hipMemcpyDeviceToDevice, s));
        d.qual = (uint8_t *)b.qual_work.p;
    }
    if (n) {
        if (realn) { ProfScope ps(e, "calmd_tag"); sta_launch_calmd_tag(s, d, apply ? 1 : 0, (uint8_t *)e->md_tag.p, (uint8_t *)e->md_state.p, saved_bq); }
        if (cp->capQ > 10) {
            // -C (bam_md.c:480-483): behind the BAQ step and its tag writer -- the qualities are what the record carries now -- and in front
            // of the MD step, whose -q / -n rewrite them
            if (!e->wd.ref) return fail(e, STA_ERR_ARG, "calmd -C needs the reference of the contig");
            if (e->md_cap.ensure(n * 2 + 16)) return fail(e, STA_ERR_HIP, "hipMalloc failed");
            ProfScope ps(e, "cap_mapq");
            sta_launch_cap_mapq_vals(s, d, e->wd, cp->capQ, (int16_t *)e->md_cap.p);
        }
        { ProfScope ps(e, "md_len"); sta_launch_md_len(s, d, e->wd, (int32_t *)e->md_nm.p, (uint32_t *)e->md_len.p, (uint8_t *)e->md_state.p); }
        { ProfScope ps(e, "len_scan"); sta_launch_len_scan(s, (const uint32_t *)e->md_len.p, (uint64_t *)e->offs.p, (int64_t)n, e->scan_tmp.p, e->scan_tmp.cap); }
    }
    uint64_t total = 0;
    if (n) HIPCHK(hipMemcpyAsync(&total, (uint64_t *)e->offs.p + n, 8, hipMemcpyDeviceToHost, s));
    SYNC_S(s);
    if (e->out.ensure((size_t)total + 64)) return fail(e, STA_ERR_HIP, "hipMalloc(output) failed");
    if (n) {
        ProfScope ps(e, "md_emit");
        sta_launch_md_emit(s, d, e->wd, (cp->flag & STA_CALMD_USE_EQUAL) ? 1 : 0, (cp->flag & STA_CALMD_BIN_QUAL) ? 1 : 0, cp->max_nm,
                           (const int32_t *)e->md_nm.p, (const uint64_t *)e->offs.p, (char *)e->out.p, (uint8_t *)e->md_seq.p);
    }
    d.bq = saved_bq;
    SYNC_S(s);
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return hipfail(e, le, "calmd kernels");
    e->out_bytes = total; e->last_out = e->out.p;
    if (info) { memset(info, 0, sizeof *info); info->out_bytes = total; }
    e->planned = 6;
    e->md_cap_valid = cp->capQ > 10;
    return STA_OK;
}

int sta_fetch_calmd_mapq_cap(sta_engine *e, int16_t *cap)
{
    if (!e || !cap) return STA_ERR_ARG;
    if (e->planned != 6 || !e->md_cap_valid) return fail(e, STA_ERR_ARG, "no calmd plan with capQ > 10");
    hipSetDevice(e->device);
    const size_t n = (size_t)e->files_h[0].n;
    if (n) HIPCHK(hipMemcpyAsync(cap, e->md_cap.p, n * 2, hipMemcpyDeviceToHost, e->stream));
    SYNC_STREAM();
    return STA_OK;
}

int sta_fetch_calmd(sta_engine *e, int32_t *nm, uint64_t *md_off, char *md_text, uint8_t *state, uint8_t *qual_pool, uint8_t *seq_pool, uint8_t *tag_pool)
{
    if (!e) return STA_ERR_ARG;
    if (e->planned != 6) return fail(e, STA_ERR_ARG, "no planned calmd window");
    hipSetDevice(e->device);
    StaReadsDev &d = e->files_h[0];
    const size_t n = (size_t)d.n, nb = (size_t)d.n_bases_total;
    hipStream_t s = e->stream;
    if (nm && n) HIPCHK(hipMemcpyAsync(nm, e->md_nm.p, n * 4, hipMemcpyDeviceToHost, s));
    if (md_off) { if (n) HIPCHK(hipMemcpyAsync(md_off, e->offs.p, (n + 1) * 8, hipMemcpyDeviceToHost, s)); else md_off[0] = 0; }
    if (md_text && e->out_bytes) HIPCHK(hipMemcpyAsync(md_text, e->out.p, (size_t)e->out_bytes, hipMemcpyDeviceToHost, s));
    if (state && n) HIPCHK(hipMemcpyAsync(state, e->md_state.p, n, hipMemcpyDeviceToHost, s));
    if (qual_pool && nb) HIPCHK(hipMemcpyAsync(qual_pool, d.qual, nb, hipMemcpyDeviceToHost, s));
    if (seq_pool && nb) HIPCHK(hipMemcpyAsync(seq_pool, e->md_seq.p, nb / 2, hipMemcpyDeviceToHost, s));
    if (tag_pool && nb) HIPCHK(hipMemcpyAsync(tag_pool, e->md_tag.p, nb, hipMemcpyDeviceToHost, s));
    SYNC_S(s);
    return STA_OK;
}

// bam_tview.c:194-212
int sta_glf_consensus(const sta_glf_col *c, char ref_base, char *call_char)
{
    if (!c) return -1;
    int qsum[4], a1, a2, tmp;
    double p[3], prior = 30;
    unsigned call;
    for (int i = 0; i < 4; ++i) qsum[i] = ((int)c->qsum[i]) << 2 | i;
    for (int i = 1; i < 4; ++i)
        for (int j = i; j > 0 && qsum[j] > qsum[j - 1]; --j) tmp = qsum[j], qsum[j] = qsum[j - 1], qsum[j - 1] = tmp;
    a1 = qsum[0] & 3; a2 = qsum[1] & 3;
    p[0] = c->p[a1 * 5 + a1]; p[1] = c->p[a1 * 5 + a2] + prior; p[2] = c->p[a2 * 5 + a2];
    const int rb = toupper((unsigned char)ref_base);
    if ("ACGT"[a1] != rb) p[0] += prior + 3;
    if ("ACGT"[a2] != rb) p[2] += prior + 3;
    if (p[0] < p[1] && p[0] < p[2]) call = (1u << a1) << 16 | (unsigned)(int)((p[1] < p[2] ? p[1] : p[2]) - p[0] + .499);
    else if (p[2] < p[1] && p[2] < p[0]) call = (1u << a2) << 16 | (unsigned)(int)((p[0] < p[1] ? p[0] : p[1]) - p[2] + .499);
    else call = (1u << a1 | 1u << a2) << 16 | (unsigned)(int)((p[0] < p[2] ? p[0] : p[2]) - p[1] + .499);
    if (call_char) *call_char = ",ACMGRSVTWYHKDBN"[call >> 16 & 0xf];
    return (int)(call & 0xffff);
}

int sta_bgzf_inflate(int32_t device, const uint8_t *comp, uint64_t comp_bytes, const sta_bgzf_block *blocks, int32_t n_blocks,
                     uint8_t *out, uint64_t out_bytes, uint32_t *status, double *kernel_ms)
{
    static_assert(sizeof(sta_bgzf_block) == sizeof(StaBgzfBlock), "the public block record is the kernel's");
    if (!comp || !blocks || !out || !status || n_blocks < 0) return STA_ERR_ARG;
    for (int32_t b = 0; b < n_blocks; ++b)
        if (blocks[b].comp_off + blocks[b].clen > comp_bytes || blocks[b].out_off + blocks[b].isize > out_bytes) return STA_ERR_ARG;
    if (hipSetDevice(device) != hipSuccess) return STA_ERR_NO_DEVICE;
    if (n_blocks == 0) return STA_OK;
    (void)hipGetLastError();       // (this thread's last error is sticky: whatever an earlier call of the process left behind is not this call's)
    uint8_t *d_comp = nullptr, *d_out = nullptr; StaBgzfBlock *d_blk = nullptr; uint32_t *d_st = nullptr;
    hipEvent_t a = nullptr, b = nullptr;
    int rc = STA_ERR_HIP;
    // (1 KiB of zeros behind the compressed bytes: the decoder's input window runs ahead of the bit reader)
    if (hipMalloc(&d_comp, comp_bytes + 1024) == hipSuccess && hipMalloc(&d_out, out_bytes + 64) == hipSuccess && hipMalloc(&d_blk, (size_t)n_blocks * sizeof(StaBgzfBlock)) == hipSuccess
        && hipMalloc(&d_st, (size_t)n_blocks * 4) == hipSuccess && hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess
        && hipMemset(d_comp + comp_bytes, 0, 1024) == hipSuccess && hipMemcpy(d_comp, comp, comp_bytes, hipMemcpyHostToDevice) == hipSuccess
        && hipMemcpy(d_blk, blocks, (size_t)n_blocks * sizeof(StaBgzfBlock), hipMemcpyHostToDevice) == hipSuccess && hipMemset(d_st, 0xff, (size_t)n_blocks * 4) == hipSuccess) {
        hipEventRecord(a, nullptr);
        sta_launch_bgzf_inflate(nullptr, d_comp, d_blk, n_blocks, d_out, d_st);
        hipEventRecord(b, nullptr);
        if (hipDeviceSynchronize() == hipSuccess && hipGetLastError() == hipSuccess && hipMemcpy(out, d_out, out_bytes, hipMemcpyDeviceToHost) == hipSuccess
            && hipMemcpy(status, d_st, (size_t)n_blocks * 4, hipMemcpyDeviceToHost) == hipSuccess) {
            float ms = 0; hipEventElapsedTime(&ms, a, b);
            if (kernel_ms) *kernel_ms = ms;
            rc = STA_OK;
        }
    }
    (void)hipGetLastError();
    if (a) hipEventDestroy(a);
    if (b) hipEventDestroy(b);
    hipFree(d_comp); hipFree(d_out); hipFree(d_blk); hipFree(d_st);
    return rc;
}

int sta_fetch_overlap_fixups(sta_engine *e, int32_t file, int32_t *fix_y, int32_t *fix_mate, uint8_t *fix_q)
{
    if (!e || file < 0 || (size_t)file >= e->files_h.size() || !fix_y || !fix_mate || !fix_q) return STA_ERR_ARG;
    if (!e->planned) return fail(e, STA_ERR_ARG, "no planned window");
    hipSetDevice(e->device);
    StaReadsDev &d = e->files_h[(size_t)file];
    if (!d.n) return STA_OK;
    if (!d.fix_y || e->ctr_h.n_olap_el == 0) { for (int64_t i = 0; i < d.n; ++i) fix_y[i] = -1; return STA_OK; }     // (no eligible read: the pass did not run)
    HIPCHK(hipMemcpyAsync(fix_y, d.fix_y, (size_t)d.n * 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemcpyAsync(fix_mate, d.fix_mate, (size_t)d.n * 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemcpyAsync(fix_q, d.fix_q, (size_t)d.n, hipMemcpyDeviceToHost, e->stream));
    SYNC_STREAM();
    return STA_OK;
}

// depth: read filters (+ -s mate clip) -> prefix max of read ends -> ONE kernel that counts every column, places the rows with a
// decoupled look-back and writes them (k_depth_fused).  out == NULL: the engine's own buffer.
static int depth_text(sta_engine *e, const sta_depth_params *p, char *out, uint64_t cap, sta_plan_info *info)
{
    hipStream_t s = e->stream;
    StaCounters *ctr = (StaCounters *)e->counters.p;
    HIPCHK(hipMemsetAsync(ctr, 0, sizeof(StaCounters), s));
    const int nf = (int)e->files_h.size();
    const int64_t ncols = (int64_t)e->wd.col_end - e->wd.col_beg;
    for (auto &d : e->files_h) d.qual = const_cast<uint8_t *>(d.qual_in);
    e->late_copy.clear();
    int rc = push_files(e);
    if (rc) return rc;
    if (e->fused_status.ensure(sta_depth_fused_status_bytes(ncols) + 64)) return fail(e, STA_ERR_HIP, "hipMalloc(look-back status) failed");
    bool status_zeroed;
    rc = chunk_state(e);
    if (rc) return rc;
    {
        ProfScope ps(e, "prep_reads_depth");
        status_zeroed = sta_launch_prep_reads_depth(s, e->wd, e->files_h.data(), nf, *p, ctr, e->chunk_st, e->fused_status.p, sta_depth_fused_status_bytes(ncols));
    }
    if (p->remove_overlaps) {
        for (int f = 0; f < nf; ++f) {
            StaReadsDev &d = e->files_h[(size_t)f];
            if (!d.n || d.clip_in) continue;          // (the caller kept the name hash itself: k_prep_reads_depth took its clip columns)
            size_t slots = sta_overlap_table_slots(d.n);
            if (e->table.ensure(sta_overlap_table_bytes(slots) + 64)) return fail(e, STA_ERR_HIP, "hipMalloc(name table) failed");
            ProfScope ps(e, "depth_pair");
            sta_launch_depth_pair(s, d, e->wd.origin, e->wd.tid, e->table.p, slots, (int32_t *)e->fb[(size_t)f].chain.p, ctr);
        }
    }
    // (R.maxend comes out of k_prep_reads_depth: nothing changes the set of kept reads or their ends afterwards)
    const size_t drows = (size_t)(nf + 1) * (size_t)(ncols + 1);
    if (e->diff.ensure(drows * 4 + 64)) return fail(e, STA_ERR_HIP, "hipMalloc(depth rows) failed");
    const bool own = out == nullptr;
    if (own) {
        const uint64_t guess = (uint64_t)(ncols > 0 ? ncols : 0) * (uint64_t)(e->tname.size() + 14 + 12 * (size_t)nf) + 4096;
        if (e->out.ensure((size_t)guess)) return fail(e, STA_ERR_HIP, "hipMalloc(output) failed");
        out = (char *)e->out.p; cap = e->out.cap;
    }
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (ncols > 0) {
            if (attempt) HIPCHK(hipMemsetAsync(&ctr->out_bytes, 0, 16, s));       // (the whole counter block was cleared before the first one)
            ProfScope ps(e, "depth_fused");
            sta_launch_depth_fused(s, e->wd, *p, e->fused_status.p, (int32_t *)e->diff.p, out, cap, ctr, depth_lbuf(), status_zeroed && !attempt, (uint32_t *)e->line_len.p);
        }
        rc = fused_finish(e, info);
        if (rc) return rc;
        if (!e->ctr_h.overflow) break;
        if (!own || attempt) return fail(e, STA_ERR_ARG, "output buffer too small (" + std::to_string(e->out_bytes) + " bytes needed)");
        if (e->out.ensure((size_t)e->out_bytes + 64)) return fail(e, STA_ERR_HIP, "hipMalloc(output) failed");
        out = (char *)e->out.p; cap = e->out.cap;
    }
    e->last_out = out;
    return STA_OK;
}

int sta_depth_plan(sta_engine *e, const sta_depth_params *p, sta_plan_info *info)
{
    if (!e || !p) return STA_ERR_ARG;
    if (!e->staged) return fail(e, STA_ERR_ARG, "no staged window");
    hipSetDevice(e->device);
    e->dp = *p;
    int rc = depth_text(e, p, nullptr, 0, info);
    if (rc) return rc;
    e->planned = 2;
    return STA_OK;
}

int sta_depth_emit(sta_engine *e, void *dev_out, uint64_t capacity)
{
    if (!e) return STA_ERR_ARG;
    if (e->planned != 2) return fail(e, STA_ERR_ARG, "sta_depth_plan has not run for this window");
    hipSetDevice(e->device);
    // the rows were written by the plan (engine buffer): a caller's buffer gets a device-to-device copy
    if (dev_out && dev_out != e->last_out) {
        if (capacity < e->out_bytes) return fail(e, STA_ERR_ARG, "output buffer too small");
        if (e->out_bytes) HIPCHK(hipMemcpyAsync(dev_out, e->last_out, (size_t)e->out_bytes, hipMemcpyDeviceToDevice, e->stream));
        e->last_out = dev_out;
    }
    return STA_OK;
}

/* plan + emit in one call; see sta_mpileup_run */
int sta_depth_run(sta_engine *e, const sta_depth_params *p, void *dev_out, uint64_t capacity, sta_plan_info *info)
{
    if (!e || !p) return STA_ERR_ARG;
    if (!e->staged) return fail(e, STA_ERR_ARG, "no staged window");
    hipSetDevice(e->device);
    e->dp = *p;
    int rc = depth_text(e, p, (char *)dev_out, capacity, info);
    if (rc) { e->planned = 0; return rc; }
    e->planned = 2;
    return STA_OK;
}

const int32_t *sta_depth_counts_dev(sta_engine *e) { return e && e->planned == 2 ? (const int32_t *)e->diff.p : nullptr; }

int sta_fetch_output(sta_engine *e, char *host_out, uint64_t n)
{
    if (!e || (!host_out && n)) return STA_ERR_ARG;
    hipSetDevice(e->device);
    if (n > e->out_bytes) return fail(e, STA_ERR_ARG, "fetch larger than output");
    if (n) HIPCHK(hipMemcpyAsync(host_out, e->last_out, (size_t)n, hipMemcpyDeviceToHost, e->stream));
    SYNC_STREAM();
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return hipfail(e, le, "emit kernel");
    return STA_OK;
}

int sta_fetch_output_at(sta_engine *e, char *host_out, uint64_t offset, uint64_t n)
{
    if (!e || (!host_out && n)) return STA_ERR_ARG;
    hipSetDevice(e->device);
    if (offset > e->out_bytes || n > e->out_bytes - offset) return fail(e, STA_ERR_ARG, "fetch beyond the output");
    if (n) HIPCHK(hipMemcpyAsync(host_out, (const char *)e->last_out + offset, (size_t)n, hipMemcpyDeviceToHost, e->stream));
    SYNC_STREAM();
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return hipfail(e, le, "emit kernel");
    return STA_OK;
}

int sta_sync(sta_engine *e)
{
    if (!e) return STA_ERR_ARG;
    hipSetDevice(e->device);
    SYNC_STREAM();
    return STA_OK;
}

void sta_profile_enable(sta_engine *e, int on) { if (e) e->prof_on = on != 0; }
void sta_profile_only(sta_engine *e, const char *name) { if (e) e->prof_only = name ? name : ""; }
void sta_profile_reset(sta_engine *e) { if (e) { prof_drain(e); e->prof.clear(); } }
int sta_profile_get(sta_engine *e, sta_kernel_time *out, int cap)
{
    if (!e) return 0;
    hipSetDevice(e->device);
    hipStreamSynchronize(e->stream);
    prof_drain(e);
    int i = 0;
    for (auto &kv : e->prof) {
        if (i < cap && out) {
            std::memset(&out[i], 0, sizeof(out[i]));
            std::snprintf(out[i].name, sizeof(out[i].name), "%s", kv.first.c_str());
            out[i].launches = kv.second.launches; out[i].total_ms = kv.second.ms;
        }
        ++i;
    }
    return i;
}

}  // extern "C"
