// driver_consensus.cpp -- `samtools-amd consensus`: the command line of `samtools consensus` (bam_consensus.c:3082-3593) over
// the window engine.  The columns and their calls come from the device (sta_consensus_run); what stays on the host is what
// the reference does once per column after consensus_base(): the FASTA / FASTQ assembly with its gap filling
// (basic_fasta, bam_consensus.c:2323-2455; dump_fastq :2054-2075), the pileup rows (basic_pileup :2191-2317, empty_pileup2
// :2107-2131) and the per-region loop of the serial driver (:2898-3075).  The threaded driver (:2626-2890) produces the same
// text and is not mirrored; -X presets and the named calibration tables are built (cons_qcal_tables.inc: the reference's arrays
// extracted as data by scripts/gen_qcal_tables.py).
#include "cons_host.h"
#include "host_io.h"
#include "host_bgzf.h"
#include "host_chunk.h"
#include "host_pump.h"
#include "host_stage.h"
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <getopt.h>
#include <strings.h>
#include <zlib.h>

namespace sta {
namespace {

enum { FMT_FASTQ, FMT_FASTA, FMT_PILEUP };

struct Opts {
    sta_cons_params p;
    int fmt = FMT_FASTA, line_len = 70, all_bases = 0, show_del = 0, show_ins = 1, mark_ins = 0, ref_qual = 0;
    const char *ref_fn = nullptr, *reg = nullptr, *bed_fn = nullptr;
    FILE *out = stdout;
};

struct Job {                     // the per-region job context of the reference (ctx, bam_consensus.c:263-293)
    const Opts *o; const Header *h; const Fasta *fa;
    std::string seq, qual, row;
    int64_t last_pos = -1; int last_tid = -1;
    const std::string *ref = nullptr; int ref_tid = -1;
    bool has_iter = false; int iter_tid = 0; int64_t iter_beg = 0, iter_end = 0;

    int64_t update_ref(int tid)
    {
        if (!o->ref_fn) return 0;
        if (tid == ref_tid && ref) return (int64_t)ref->size();
        ref = nullptr; ref_tid = tid;
        if (tid < 0 || tid >= h->nref()) return -1;
        ref = fa->fetch(h->names[(size_t)tid]);
        return ref ? (int64_t)ref->size() : -1;
    }
    void empty_rows(int tid, int64_t start, int64_t end)
    {
        const std::string *rs = nullptr;
        if (o->ref_fn && update_ref(tid) > 0) rs = ref;
        for (int64_t i = start; i < end; ++i)
            fprintf(o->out, "%s\t%lld\t0\t0\t%c\t0\t*\t*\n", h->names[(size_t)tid].c_str(), (long long)(i + 1), rs && i < (int64_t)rs->size() ? (*rs)[(size_t)i] : 'N');
    }
    void fill_flat(int64_t from, int64_t n)
    {
        for (int64_t i = 0; i < n; ++i) {
            seq += ref && from + i < (int64_t)ref->size() ? (*ref)[(size_t)(from + i)] : 'N';
            qual += (char)((ref ? o->ref_qual : 0) + '!');
        }
    }
    void dump(const std::string &name)
    {
        if (seq.empty()) return;
        fprintf(o->out, "%c%s\n", ">@"[o->fmt == FMT_FASTQ], name.c_str());
        const size_t ll = (size_t)o->line_len;
        for (size_t i = 0; i < seq.size(); i += ll) { fwrite(seq.data() + i, 1, std::min(ll, seq.size() - i), o->out); fputc('\n', o->out); }
        if (o->fmt != FMT_FASTQ) return;
        fputs("+\n", o->out);
        for (size_t i = 0; i < seq.size(); i += ll) { fwrite(qual.data() + i, 1, std::min(ll, seq.size() - i), o->out); fputc('\n', o->out); }
    }

    // one column of `-f pileup`
    int column_pileup(int tid, int64_t pos, int nth, const sta_cons_col &col, const char *sc, const char *qc)
    {
        if (!o->show_ins && nth) return 0;
        if (has_iter && (iter_beg >= pos || iter_end < pos)) return 0;
        if (o->all_bases) {
            if (tid != last_tid && last_tid >= -1) {
                if (last_tid >= 0) {
                    int64_t len = h->lens[(size_t)last_tid];
                    if (has_iter && iter_end < len) len = iter_end;
                    empty_rows(last_tid, last_pos, len);
                }
                last_pos = has_iter ? iter_beg : 0;
            }
            if (!has_iter && tid > last_tid && o->all_bases > 1)
                while (++last_tid < tid) empty_rows(last_tid, 0, h->lens[(size_t)last_tid]);
            if (last_pos >= 0 && pos > last_pos + 1) empty_rows(tid, last_pos, pos - 1);
            else if (last_pos < 0) empty_rows(tid, has_iter ? iter_beg : 0, pos - 1);
        }
        if (!o->show_del && col.base == '*') return 0;
        char num[96];
        row.assign(h->names[(size_t)tid]);
        const int n = snprintf(num, sizeof num, "\t%lld\t%d\t%d\t%c\t%d\t", (long long)pos, nth, col.depth, (char)col.base, col.qual);
        row.append(num, (size_t)n);
        row.append(sc, (size_t)col.depth); row += '\t';
        row.append(qc, (size_t)col.depth); row += '\n';
        fwrite(row.data(), 1, row.size(), o->out);
        last_pos = pos; last_tid = tid;
        return 0;
    }

    // one column of the FASTA / FASTQ assembly
    int column_fasta(int tid, int64_t pos, int nth, const sta_cons_col &col)
    {
        if (!o->show_ins && nth) return 0;
        if (has_iter && (iter_beg >= pos || iter_end < pos)) return 0;
        while (tid != last_tid) {
            if (last_tid != -1) {
                if (o->all_bases) {
                    int64_t N = INT64_MAX;
                    if (has_iter) { last_pos = std::max(last_pos, iter_beg - 1); N = iter_end; }
                    N = std::min(N, h->lens[(size_t)last_tid]) - last_pos;
                    if (N > 0) {
                        if (ref && update_ref(last_tid) < 0) return -1;
                        fill_flat(last_pos, N);
                    }
                }
                dump(h->names[(size_t)last_tid]);
            }
            if (update_ref(tid) < 0) return -1;
            seq.clear(); qual.clear();
            if (!has_iter && o->all_bases > 1 && ++last_tid < tid) { last_pos = 0; continue; }
            last_tid = tid;
            last_pos = o->all_bases ? (has_iter ? iter_beg : 0) : pos - 1;
        }
        const int cb = col.base, cq = col.qual;
        if (!o->show_del && cb == '*') { last_pos = pos; last_tid = tid; return 0; }
        if (o->mark_ins && nth && cb != '*') { seq += '_'; qual += '_'; }
        if (pos > last_pos && (last_pos > 0 || o->all_bases)) {
            if (update_ref(tid) < 0) return -1;
            fill_flat(last_pos, pos - (last_pos + 1));
        }
        if ((nth && o->show_ins && cb != '*') || cb != '*' || (pos > last_pos && o->show_del)) {
            seq += (char)cb;
            qual += (char)(std::min(cq, '~' - '!') + '!');
        }
        last_pos = pos; last_tid = tid;
        return 0;
    }
};

// --regions-file: samtools keeps the BED in a khash keyed by contig name and walks its buckets (bedidx.c:593-646 via
// bam_consensus.c:2917-2925), so the contig order of the output is the table's: FNV-1a string hash, open addressing with
// triangular probing, growth at 77 % load.  Intervals of a contig come sorted by (beg, end) (bedidx.c:43-47,150).
struct BedOrder {
    struct Chr { std::string name; std::vector<std::pair<int64_t, int64_t>> iv; };
    std::vector<Chr> chr;
    std::vector<int> slot;       // bucket -> chr index, -1 empty
    int size = 0, upper = 0;
    static uint32_t fnv1a(const std::string &s) { uint32_t h = 2166136261u; for (unsigned char c : s) h = (h ^ c) * 16777619u; return h; }
    void place(std::vector<int> &tab, int key) const
    {
        const uint32_t mask = (uint32_t)tab.size() - 1;
        uint32_t i = fnv1a(chr[(size_t)key].name) & mask, step = 0;
        while (tab[i] >= 0) i = (i + (++step)) & mask;
        tab[i] = key;
    }
    void grow()
    {
        size_t nb = slot.size() + 1, v = nb - 1;
        v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; nb = std::max<size_t>(4, v + 1);
        std::vector<int> old(slot), moved(slot.size(), 0), nt(nb, -1);
        const uint32_t mask = (uint32_t)nb - 1;
        for (size_t j = 0; j < old.size(); ++j) {                 // khash's in-place rehash with its kick-out chain
            if (old[j] < 0 || moved[j]) continue;
            int key = old[j]; moved[j] = 1;
            for (;;) {
                uint32_t i = fnv1a(chr[(size_t)key].name) & mask, step = 0;
                while (nt[i] >= 0) i = (i + (++step)) & mask;
                nt[i] = key;
                if (i < old.size() && old[i] >= 0 && !moved[i]) { key = old[i]; moved[i] = 1; } else break;
            }
        }
        slot.swap(nt); upper = (int)(nb * 0.77 + 0.5);
    }
    int get(const std::string &name)
    {
        for (size_t i = 0; i < chr.size(); ++i) if (chr[i].name == name) return (int)i;
        if (size >= upper) grow();
        chr.push_back(Chr{ name, {} });
        place(slot, (int)chr.size() - 1);
        ++size;
        return (int)chr.size() - 1;
    }
    bool read(const char *fn)
    {
        gzFile fp = gzopen(fn, "r");
        if (!fp) return false;
        std::vector<char> line(1 << 16);
        while (gzgets(fp, line.data(), (int)line.size())) {
            char *ref = line.data();
            while (*ref && isspace((unsigned char)*ref)) ++ref;
            if (!*ref || *ref == '#') continue;
            char *e = ref; while (*e && !isspace((unsigned char)*e)) ++e;
            unsigned long long beg = 0, end = 0; int num = 0;
            if (*e) { *e = 0; num = sscanf(e + 1, "%llu %llu", &beg, &end); }
            if (num == 1) end = beg--;
            if (num < 1 || end < beg) {
                if (!strcmp(ref, "browser") || !strcmp(ref, "track")) continue;
                fprintf(stderr, "[bed_read] Parse error reading \"%s\"\n", fn);
                gzclose(fp); return false;
            }
            chr[(size_t)get(ref)].iv.emplace_back((int64_t)beg, (int64_t)end);
        }
        gzclose(fp);
        for (auto &c : chr) std::sort(c.iv.begin(), c.iv.end());
        return true;
    }
};

struct Runner {
    const Opts &o; const ConsCompute &compute; const char *fn;
    int64_t window_cols;

    // pileup_loop over one (possibly region-restricted) pass of the input: windows -> columns -> writer
    int pass(Job &job)
    {
        std::string err;
        std::vector<std::unique_ptr<AlnReader>> readers;
        readers.push_back(AlnReader::open(fn, &err));
        if (!readers[0]) { fprintf(stderr, "samtools consensus: Cannot open input file \"%s\"\n", fn); return -1; }
        const bool want_md = o.p.mode != STA_CONS_SIMPLE && o.p.use_mqual;
        if (want_md) readers[0]->set_wanted_tags({ "MD" });
        if (job.has_iter) readers[0]->set_region(job.iter_tid, job.iter_beg, job.iter_end);
        const Header &h = readers[0]->header();
        PumpConfig pc; pc.window_cols = window_cols; pc.use_endpos = false; pc.nref_limit = h.nref();
        pc.xs_n_tags = want_md ? 1 : 0; pc.xs_empty = '*';
        // the drivers' input lanes (host_chunk.h: records parsed on several threads into chunk slices; host_pump.h: one record
        // at a time, STA_INPUT_LANE=rec)
        std::unique_ptr<WindowSource> src;
        const char *lane = getenv("STA_INPUT_LANE");
        if (lane && !strcmp(lane, "rec")) src.reset(new Pump(readers, pc));
        else src.reset(new ChunkPump(readers, pc, io_default_threads()));
        WindowSource &pump = *src;
        std::vector<StagedFile> stagedv(1);
        ConsWindowOut out;
        for (;;) {
            const int tid = pump.next_tid();
            if (pump.error() || tid < 0) break;
            int64_t cursor = pump.next_pos(tid);
            for (;;) {
                if (pump.next_pos(tid) == INT64_MAX && !pump.has_carry()) break;
                // reads whose last base is the column before the window stay staged (retire(ce - 1) below): the insertion columns
                // of that position decide one flag of the window's first column (cons_window.h step_walk).  When nothing is
                // alive at `cursor` they are not needed and the next window starts at the next read.
                if (pump.has_carry() && pump.carry_max_end() <= cursor && pump.next_pos(tid) > cursor) pump.retire(cursor);
                if (pump.next_pos(tid) == INT64_MAX && !pump.has_carry()) break;
                if (!pump.has_carry()) cursor = std::max(cursor, pump.next_pos(tid));
                int64_t ce = pump.fill_staged(tid, cursor, cursor + window_cols, stagedv);
                if (pump.error()) break;
                if (pump.next_pos(tid) == INT64_MAX) {
                    const int64_t me = pump.carry_max_end();
                    if (me != INT64_MIN) ce = std::min(ce, std::max(me, cursor));
                }
                if (ce > cursor) {
                    sta_reads view = stagedv[0].view();
                    sta_window w; memset(&w, 0, sizeof w);
                    w.tid = tid; w.origin = cursor; w.col_beg = 0; w.col_end = (int32_t)(ce - cursor);
                    w.tname = h.names[(size_t)tid].c_str(); w.tlen = h.lens[(size_t)tid];
                    w.n_files = 1; w.files = &view; w.mem = STA_MEM_HOST;
                    if (compute(w, o.p, out, err) < 0) { fprintf(stderr, "samtools consensus: %s\n", err.c_str()); return -1; }
                    uint64_t c = 0;
                    const int32_t W = w.col_end - w.col_beg;
                    for (int32_t p = 0; p < W; ++p) {
                        for (int32_t n = 0; n <= out.ins[(size_t)p]; ++n, ++c) {
                            const sta_cons_col &col = out.cols[c];
                            if (col.depth <= 0) continue;
                            const int64_t pos = cursor + p + 1;
                            const int v = o.fmt == FMT_PILEUP
                                ? job.column_pileup(tid, pos, n, col, out.seq.data() + out.col_off[c], out.qual.data() + out.col_off[c])
                                : job.column_fasta(tid, pos, n, col);
                            if (v < 0) return -1;
                        }
                    }
                }
                pump.retire(ce - 1);
                cursor = std::max(cursor, ce);
            }
            if (pump.error()) break;
            pump.drop_tid_carry();
        }
        if (pump.error()) { fprintf(stderr, "samtools consensus: %s\n", pump.error_text()); return -1; }
        return 0;
    }

    // what pileup_loop_serial does after the column loop of one region (bam_consensus.c:2985-3064)
    int finish(Job &c, const Header &h)
    {
        if (o.fmt == FMT_PILEUP) {
            if (o.all_bases) {
                const int tid = c.has_iter ? c.iter_tid : c.last_tid;
                int64_t len = tid >= 0 && tid < h.nref() ? h.lens[(size_t)tid] : 0, pos = c.last_pos;
                if (c.has_iter) { len = std::min(c.iter_end, len); pos = std::max(c.iter_beg, pos); }
                if (tid >= 0) c.empty_rows(tid, pos, len);
            }
            while (!c.has_iter && o.all_bases > 1 && ++c.last_tid < h.nref()) c.empty_rows(c.last_tid, 0, (int)h.lens[(size_t)c.last_tid]);
            return 0;
        }
        for (;;) {
            if (o.all_bases) {
                const int tid = c.has_iter ? c.iter_tid : c.last_tid;
                int64_t len = tid >= 0 && tid < h.nref() ? h.lens[(size_t)tid] : 0, pos = c.last_pos;
                if (c.has_iter) { len = std::min(c.iter_end, len); pos = std::max(c.iter_beg, pos); c.last_tid = c.iter_tid; }
                if (pos < len) {
                    if (c.update_ref(c.last_tid) < 0) return -1;
                    c.fill_flat(pos, len - pos);
                }
            }
            if (c.last_tid >= 0) {
                const int tid = c.has_iter ? c.iter_tid : c.last_tid;
                const int len = (int)h.lens[(size_t)tid];
                std::string name = h.names[(size_t)c.last_tid];
                if (c.has_iter && (c.iter_beg > 0 || c.iter_end < len))
                    name += ":" + std::to_string(c.iter_beg + 1) + "-" + std::to_string(std::min<int64_t>(c.iter_end, len));
                c.dump(name);
            }
            if (!c.has_iter && o.all_bases > 1 && ++c.last_tid < h.nref()) { c.last_pos = 0; c.seq.clear(); c.qual.clear(); continue; }
            break;
        }
        return 0;
    }
};

#include "cons_qcal_tables.inc"

// one of the built-in tables (bam_consensus.c:664-670 set_qcal): 0 flat, 1 hifi, 2 hiseq, 3 r10.4_sup, 4 r10.4_dup, 5 ultima
int set_qcal(int32_t q[3][101], int id)
{
    if (id < 0 || id >= 6) return -1;
    memcpy(q, QCAL_TABLES[id], sizeof(int32_t) * 3 * 101);
    return 0;
}

// --qual-calibration (bam_consensus.c:672-738): ":name" = a built-in table, else a file of "QUAL v sub under over" lines
int load_qcal(int32_t q[3][101], const char *fn)
{
    for (int id = 1; id < 6; ++id) if (fn[0] == ':' && !strcmp(fn + 1, QCAL_NAMES[id])) return set_qcal(q, id);
    for (int i = 0; i < 101; ++i) q[0][i] = q[1][i] = q[2][i] = i;
    if (!strcmp(fn, ":flat")) return 0;
    FILE *fp = fopen(fn, "r");
    if (!fp) return -1;
    char line[1024];
    int mx = 0, last_qual = 0;
    while (fgets(line, sizeof line, fp)) {
        int v, s, u, ov;
        if (*line == '#') continue;
        if (sscanf(line, "QUAL %d %d %d %d", &v, &s, &u, &ov) != 4) { fclose(fp); return -1; }
        while (v > last_qual && last_qual < 100) { for (int k = 0; k < 3; ++k) q[k][last_qual + 1] = q[k][last_qual]; last_qual++; }
        if (v >= 0 && v < 100) { q[0][v] = s; q[1][v] = u; q[2][v] = ov; }
        if (v < mx) { fprintf(stderr, "Qual calibration file is not in ascending order\n"); fclose(fp); return -1; }
        mx = v;
    }
    for (int i = mx + 1; i < 101; ++i) for (int k = 0; k < 3; ++k) q[k][i] = q[k][mx];
    fclose(fp);
    return 0;
}

}  // namespace

int consensus_cli(int argc, char **argv, const ConsCompute &compute)
{
    Opts o;
    sta_cons_params &p = o.p; memset(&p, 0, sizeof p);
    p.mode = STA_CONS_RECALL; p.adj_qual = 1; p.use_mqual = 1; p.scale_mqual = 1.00; p.nm_adjust = 1; p.nm_halo = 50; p.sc_cost = 60;
    p.low_mqual = 1; p.high_mqual = 60; p.min_depth = 1; p.call_fract = 0.75; p.het_fract = 0.5; p.cons_cutoff = 10; p.default_qual = 10;
    p.excl_flags = 4 | 256 | 512 | 1024; p.P_het = 1e-3; p.P_indel = 2e-4; p.het_scale = 1.0; p.homopoly_redux = 0.01;
    set_qcal(p.qcal, 0);                    // (bam_consensus.c:3196: the flat table)

    static const struct option lopts[] = {
        { "use-qual", no_argument, NULL, 'q' }, { "no-use-qual", no_argument, NULL, 'q' + 1000 }, { "adj-qual", no_argument, NULL, 'q' + 100 },
        { "no-adj-qual", no_argument, NULL, 'q' + 101 }, { "use-MQ", no_argument, NULL, 'm' + 1000 }, { "no-use-MQ", no_argument, NULL, 'm' + 1001 },
        { "adj-MQ", no_argument, NULL, 'm' + 100 }, { "no-adj-MQ", no_argument, NULL, 'm' + 101 }, { "NM-halo", required_argument, NULL, 'h' + 100 },
        { "SC-cost", required_argument, NULL, 'h' + 101 }, { "scale-MQ", required_argument, NULL, 14 }, { "low-MQ", required_argument, NULL, 9 },
        { "high-MQ", required_argument, NULL, 10 }, { "min-depth", required_argument, NULL, 'd' }, { "call-fract", required_argument, NULL, 'c' },
        { "het-fract", required_argument, NULL, 'H' }, { "region", required_argument, NULL, 'r' }, { "regions-file", required_argument, NULL, 'r' + 1000 },
        { "format", required_argument, NULL, 'f' }, { "cutoff", required_argument, NULL, 'C' }, { "ambig", no_argument, NULL, 'A' },
        { "line-len", required_argument, NULL, 'l' }, { "default-qual", required_argument, NULL, 1 }, { "het-only", no_argument, NULL, 6 },
        { "show-del", required_argument, NULL, 7 }, { "show-ins", required_argument, NULL, 8 }, { "mark-ins", no_argument, NULL, 18 },
        { "output", required_argument, NULL, 'o' }, { "incl-flags", required_argument, NULL, 11 }, { "rf", required_argument, NULL, 11 },
        { "excl-flags", required_argument, NULL, 12 }, { "ff", required_argument, NULL, 12 }, { "min-MQ", required_argument, NULL, 13 },
        { "min-BQ", required_argument, NULL, 16 }, { "P-het", required_argument, NULL, 15 }, { "P-indel", required_argument, NULL, 17 },
        { "het-scale", required_argument, NULL, 19 }, { "mode", required_argument, NULL, 'm' }, { "homopoly-fix", no_argument, NULL, 'p' },
        { "homopoly-score", required_argument, NULL, 'p' + 100 }, { "homopoly-redux", required_argument, NULL, 'p' + 200 },
        { "qual-calibration", required_argument, NULL, 't' }, { "config", required_argument, NULL, 'X' }, { "ref-qual", required_argument, NULL, 20 },
        { "block-size", required_argument, NULL, 'Z' }, { "reference", required_argument, NULL, 'T' }, { "threads", required_argument, NULL, '@' },
        { NULL, 0, NULL, 0 } };
    const char *usage = "Usage: samtools consensus [options] <in.bam>\n";
    int c;
    optind = 0;          // (glibc: 0 = full re-initialisation; with 1 a second in-process call resumes at a stale pointer into the PREVIOUS argv)
    while ((c = getopt_long(argc, argv, "@:qd:c:H:r:5f:C:aAl:o:m:pt:X:T:Z:", lopts, NULL)) >= 0) {
        switch (c) {
        case 'a': o.all_bases++; break;
        case 'q': p.use_qual = 1; break;
        case 'q' + 1000: p.use_qual = 0; break;
        case 'm' + 1000: p.use_mqual = 1; break;
        case 'm' + 1001: p.use_mqual = 0; break;
        case 14: p.scale_mqual = atof(optarg); break;
        case 9: p.low_mqual = atoi(optarg); break;
        case 10: p.high_mqual = atoi(optarg); break;
        case 'd': p.min_depth = atoi(optarg); break;
        case 'c': p.call_fract = atof(optarg); break;
        case 'H': p.het_fract = atof(optarg); break;
        case 'r':
            if (o.bed_fn) { fprintf(stderr, "samtools consensus: option -r and --regions-file are incompatible\n"); return 1; }
            o.reg = optarg; break;
        case 'r' + 1000:
            if (o.reg) { fprintf(stderr, "samtools consensus: option -r and --regions-file are incompatible\n"); return 1; }
            o.bed_fn = optarg; break;
        case 'C': p.cons_cutoff = atoi(optarg); break;
        case 'A': p.ambig = 1; break;
        case 'p': p.homopoly_fix = 0.5; break;
        case 'p' + 100: p.homopoly_fix = atof(optarg); break;
        case 'p' + 200: p.homopoly_redux = atof(optarg); break;
        case 1: p.default_qual = atoi(optarg); break;
        case 6: break;
        case 7: o.show_del = (*optarg == 'y' || *optarg == 'Y'); break;
        case 8: o.show_ins = (*optarg == 'y' || *optarg == 'Y'); break;
        case 18: o.mark_ins = 1; break;
        case 13: p.min_mqual = atoi(optarg); break;
        case 16: p.min_qual = atoi(optarg); break;
        case 15: p.P_het = atof(optarg); break;
        case 17: p.P_indel = atof(optarg); break;
        case 19: p.het_scale = atof(optarg); break;
        case 'q' + 100: p.adj_qual = 1; break;
        case 'q' + 101: p.adj_qual = 0; break;
        case 'm' + 100: p.nm_adjust = 1; break;
        case 'm' + 101: p.nm_adjust = 0; break;
        case 'h' + 100: p.nm_halo = atoi(optarg); break;
        case 'h' + 101: p.sc_cost = atoi(optarg); break;
        case 'Z': case '@': break;       // block size / threads of the reference's threaded driver: windows are the engine's
        case 'm':
            if (!strcasecmp(optarg, "simple")) p.mode = STA_CONS_SIMPLE;
            else if (!strcasecmp(optarg, "bayesian_m")) p.mode = STA_CONS_MIXED;
            else if (!strcasecmp(optarg, "bayesian_p")) p.mode = STA_CONS_PRECISE;
            else if (!strcasecmp(optarg, "bayesian_r") || !strcasecmp(optarg, "bayesian")) p.mode = STA_CONS_RECALL;
            else if (!strcasecmp(optarg, "bayesian_116")) p.mode = STA_CONS_BAYES_116;
            else { fprintf(stderr, "Unknown mode %s\n", optarg); return 1; }
            break;
        case 'l': if ((o.line_len = atoi(optarg)) <= 0) o.line_len = INT_MAX; break;
        case 'f':
            if (!strcasecmp(optarg, "fasta")) o.fmt = FMT_FASTA;
            else if (!strcasecmp(optarg, "fastq")) o.fmt = FMT_FASTQ;
            else if (!strcasecmp(optarg, "pileup")) o.fmt = FMT_PILEUP;
            else { fprintf(stderr, "Unknown format %s\n", optarg); return 1; }
            break;
        case 'o': if (!(o.out = fopen(optarg, "w"))) { perror(optarg); return 1; } break;
        case 'X':
            // --config: the machine profiles of bam_consensus.c:3366-3421 -- a calibration table plus the option values tuned with it
            if (!strcasecmp(optarg, "hifi") || !strcasecmp(optarg, "r10.4_sup") || !strcasecmp(optarg, "r10.4_dup")) {
                set_qcal(p.qcal, !strcasecmp(optarg, "hifi") ? 1 : !strcasecmp(optarg, "r10.4_sup") ? 3 : 4);
                p.mode = STA_CONS_RECALL; p.homopoly_fix = 0.3; p.homopoly_redux = 0.01; p.low_mqual = 5; p.scale_mqual = 1.5; p.het_scale = 0.37;
            } else if (!strcasecmp(optarg, "hiseq")) {
                p.mode = STA_CONS_RECALL; set_qcal(p.qcal, 2); p.homopoly_redux = 0.01;
            } else if (!strcasecmp(optarg, "ultima")) {
                p.mode = STA_CONS_RECALL; set_qcal(p.qcal, 5); p.homopoly_fix = 0.3; p.homopoly_redux = 0.01; p.het_scale = 0.37; p.scale_mqual = 2; p.low_mqual = 10;
            } else { fprintf(stderr, "Unrecognised configuration name: \"%s\"\n", optarg); return 1; }
            break;
        case 11: if ((p.incl_flags = str2flag(optarg)) < 0) { fprintf(stderr, "samtools consensus: could not parse --rf %s\n", optarg); return 1; } break;
        case 12: if ((p.excl_flags = str2flag(optarg)) < 0) { fprintf(stderr, "samtools consensus: could not parse --ff %s\n", optarg); return 1; } break;
        case 't': if (load_qcal(p.qcal, optarg) < 0) { fprintf(stderr, "samtools consensus: failed to load quality calibration '%s'\n", optarg); return 1; } break;
        case 'T': o.ref_fn = optarg; break;
        case 20: o.ref_qual = atoi(optarg); break;
        default: fputs(usage, stderr); return 1;
        }
    }
    if (argc != optind + 1) { fputs(usage, argc == optind ? stdout : stderr); return argc == optind ? 0 : 1; }
    p.want_pileup = o.fmt == FMT_PILEUP;
    const char *fn = argv[optind];
    std::string err;
    std::unique_ptr<AlnReader> r0 = AlnReader::open(fn, &err);
    if (!r0) { fprintf(stderr, "samtools consensus: Cannot open input file \"%s\"\n", fn); return 1; }
    const Header h = r0->header();
    r0.reset();
    std::unique_ptr<Fasta> fa;
    if (o.ref_fn && !(fa = Fasta::load(o.ref_fn))) { fprintf(stderr, "Failed to load fai for %s\n", o.ref_fn); return 1; }
    int64_t window_cols = 1 << 20;
    if (const char *e = getenv("STA_WINDOW_COLS")) window_cols = std::min<long long>(std::max<long long>(1, atoll(e)), 1 << 24);
    Runner run{ o, compute, fn, window_cols };

    struct Iv { int tid; int64_t beg, end; };
    std::vector<Iv> ivs;
    bool by_bed = false;
    int ret = 0;
    Job job; job.o = &o; job.h = &h; job.fa = fa.get();
    if (o.bed_fn) {
        BedOrder bed;
        if (!bed.read(o.bed_fn)) { fprintf(stderr, "samtools consensus: Could not read file \"%s\"\n", o.bed_fn); return 1; }
        for (int key : bed.slot) {
            if (key < 0) continue;
            const int tid = h.tid(bed.chr[(size_t)key].name);
            if (tid < 0) { fprintf(stderr, "[W::fill_reglist_tid] Region '%s' specifies an unknown reference name\n", bed.chr[(size_t)key].name.c_str()); continue; }
            for (auto &iv : bed.chr[(size_t)key].iv) ivs.push_back(Iv{ tid, iv.first, iv.second });
        }
        by_bed = true;
        if (ivs.empty()) ret = 1;
    } else if (o.reg) {
        int t; int64_t bb, ee;
        if (!parse_region(h, o.reg, &t, &bb, &ee)) { fprintf(stderr, "samtools consensus: Failed to parse region \"%s\"\n", o.reg); ret = 1; }
        else { job.has_iter = true; job.iter_tid = t; job.iter_beg = bb; job.iter_end = ee; }
    }
    if (!ret) {
        size_t k = 0;
        do {
            if (by_bed) {
                job.row.clear(); job.seq.clear(); job.qual.clear();
                job.last_tid = -1; job.last_pos = -1; job.ref_tid = -1; job.ref = nullptr; job.has_iter = false;
                if (k >= ivs.size()) break;
                const Iv iv = ivs[k++];
                int64_t start = iv.beg, end = iv.end;
                if (start > end || start > h.lens[(size_t)iv.tid]) {
                    fprintf(stderr, "[consensus] Warning: Invalid region \"%s:%lld-%lld\"\n", h.names[(size_t)iv.tid].c_str(), (long long)start, (long long)end);
                    continue;
                }
                end = std::min(end, h.lens[(size_t)iv.tid]);
                job.has_iter = true; job.iter_tid = iv.tid; job.iter_beg = start; job.iter_end = end;
                job.last_pos = start;
            }
            if (run.pass(job) < 0 || run.finish(job, h) < 0) { ret = 1; break; }
        } while (k < ivs.size());
    }
    if (o.out != stdout) ret |= fclose(o.out) != 0; else ret |= fflush(stdout) != 0;
    if (ret) fprintf(stderr, "samtools consensus: failed\n");
    return ret;
}

}  // namespace sta
