// driver_calmd.cpp -- `samtools-amd calmd`: calmd on the engine, records out as SAM text.
// The loop of bam_fillmd (bam_md.c:457-497) restated over batches: records of one contig are staged as a window (the window is
// only a coordinate frame here), sta_calmd_plan runs BAQ (-r) and the MD / NM kernels, and every record is written the way
// sam_write1 would write it after sam_prob_realn + bam_fillmd1_core touched it:
//   calmd [-e] [-r] [-A] [-E] [-q] [-d] [-N] [-Q] [-n max_nm] [--no-PG] in.bam ref.fa  > out.sam
// The device computes what changes (NM, the MD string, '=' bases, qualities, the BQ / ZQ string and which of realn.c's tag branches a
// record took: kernels_md.hip); the host keeps the aux fields as text (host_io.h Rec::auxv) and does the bookkeeping of
// bam_md.c:156-199 on them -- a tag whose stored value is right stays where it is, a wrong one is removed and the new value appended.
// Output is SAM with the header (mode "wh"), or BAM with -b (compressed) / -u (stored BGZF blocks) through host_bamout.h; -C (sam_cap_mapq on the device: k_cap_mapq_vals) lowers the MAPQ field as bam_md.c:480-483 does.
#include "../../include/samtools_amd.h"
#include "host_io.h"
#include "host_bamout.h"
#include "host_stage.h"
#include <cctype>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <getopt.h>
#include <set>

using namespace sta;

namespace {

struct Ctx {
    sta_engine *eng = nullptr;
    sta_calmd_params cp{};
    const Header *h = nullptr;
    bool quiet = false, drop_tag = false, update = true;
    std::vector<Rec> batch;
    StagedFile staged;
    std::vector<int32_t> nm; std::vector<uint64_t> off; std::vector<char> md; std::vector<uint8_t> state, qual, seq, tag; std::vector<int16_t> cap;
    std::string line;
    FILE *out = stdout;
    std::unique_ptr<BamWriter> bam;            // -b / -u
    bool io_error = false;
};

int find_tag(const std::vector<std::string> &aux, const char *tag)
{
    for (size_t i = 0; i < aux.size(); ++i) if (aux[i].size() >= 5 && aux[i][0] == tag[0] && aux[i][1] == tag[1]) return (int)i;
    return -1;
}

void put_record(Ctx &c, const Rec &r, const uint8_t *seq4, const uint8_t *qual, const std::vector<std::string> &aux)
{
    if (c.bam) { if (!c.bam->record(*c.h, r, seq4, qual, aux)) c.io_error = true; return; }
    format_sam_record(*c.h, r, seq4, qual, aux, c.line);
    fwrite(c.line.data(), 1, c.line.size(), c.out);
}

void put_unchanged(Ctx &c, const Rec &r) { put_record(c, r, r.seq.data(), r.qual.data(), r.auxv); }

// one batch = records of one contig in non-decreasing position order
int flush(Ctx &c, int tid, const std::string *ref)
{
    if (c.batch.empty()) return 0;
    if (!ref) {
        if (c.cp.capQ > 10) return -1;            // (bam_md.c:471: "Would otherwise crash" -- fatal with -r or -C)
        for (const Rec &r : c.batch) put_unchanged(c, r); c.batch.clear(); return 0;
    }
    const bool realn = (c.cp.flag & STA_CALMD_REALN) != 0, apply = (c.cp.flag & STA_CALMD_APPLY) != 0;
    const int64_t origin = c.batch.front().pos;
    int64_t hi = origin + 1;
    c.staged.clear();
    for (Rec &r : c.batch) {
        // realn.c without -A: a ZQ:Z tag (and no BQ:Z) goes back into the qualities; its bytes ride in the BQ pool
        if (realn && !apply && r.has_zq && !r.has_bq && (int32_t)r.zq.size() >= r.l_qseq) { r.bq = r.zq; r.zq_restore = true; }
        c.staged.add(r, origin, nullptr);
        hi = std::max(hi, r.end() + 1);
    }
    c.staged.finish();
    sta_reads view = c.staged.view();
    sta_window w; memset(&w, 0, sizeof w);
    w.tid = tid; w.origin = origin; w.col_beg = 0; w.col_end = (int32_t)std::min<int64_t>(hi - origin, INT32_MAX - 1);
    w.tname = c.h->names[(size_t)tid].c_str(); w.tlen = c.h->lens[(size_t)tid];
    w.n_files = 1; w.files = &view; w.mem = STA_MEM_HOST;
    sta_plan_info pi;
    if (sta_stage_window(c.eng, &w) != STA_OK || sta_calmd_plan(c.eng, &c.cp, &pi) != STA_OK) { fprintf(stderr, "samtools calmd: %s\n", sta_last_error(c.eng)); return -1; }
    const size_t n = c.batch.size();
    c.nm.resize(n); c.off.resize(n + 1); c.md.resize((size_t)pi.out_bytes + 1); c.state.resize(n);
    c.qual.resize(c.staged.qual.size()); c.seq.resize(c.staged.seq.size() + 1); c.tag.resize(c.staged.qual.size());
    if (sta_fetch_calmd(c.eng, c.nm.data(), c.off.data(), c.md.data(), c.state.data(), c.qual.data(), c.seq.data(), c.tag.data()) != STA_OK) {
        fprintf(stderr, "samtools calmd: %s\n", sta_last_error(c.eng)); return -1;
    }
    if (c.cp.capQ > 10) {
        c.cap.resize(n);
        if (sta_fetch_calmd_mapq_cap(c.eng, c.cap.data()) != STA_OK) { fprintf(stderr, "samtools calmd: %s\n", sta_last_error(c.eng)); return -1; }
    }
    std::string t;
    for (size_t i = 0; i < n; ++i) {
        Rec &r = c.batch[i];
        const size_t boff = (size_t)c.staged.base_off8[i] << 3;
        // bam_md.c:480-483: `if (b->core.qual > q) b->core.qual = q;` -- q = -1 (too many mismatches) compares below every quality and is
        // stored into the unsigned field: 255
        if (c.cp.capQ > 10 && (int)r.mapq > (int)c.cap[i]) r.mapq = (uint8_t)c.cap[i];
        std::vector<std::string> &aux = r.auxv;
        const uint8_t st = c.state[i];
        if (realn && !(r.flag & 4) && r.l_qseq > 0 && r.qual[0] != 0xff) {
            // sam_prob_realn's tag branches (HTSlib realn.c; call site bam_md.c:474-479)
            const int bq = find_tag(aux, "BQ"), zq = find_tag(aux, "ZQ");
            if (bq >= 0 && zq >= 0) aux.erase(aux.begin() + zq);                             // both: the ZQ tag is removed
            if (st & STA_CALMD_BQ_TO_ZQ) { const int k = find_tag(aux, "BQ"); if (k >= 0) aux[(size_t)k][0] = 'Z'; }
            if (st & STA_CALMD_ZQ_TO_BQ) { const int k = find_tag(aux, "ZQ"); if (k >= 0) aux[(size_t)k][0] = 'B'; }
            if (st & STA_CALMD_NEW_TAG) {
                t = apply ? "ZQ:Z:" : "BQ:Z:";
                t.append((const char *)c.tag.data() + boff, (size_t)r.l_qseq);
                aux.push_back(t);
            }
        }
        if (r.l_qseq == 0) {
            if (!c.quiet)
                fprintf(stderr, "[bam_fillmd1] no sequence in alignment record for '%s' at %s:%lld, skipped\n", r.qname.c_str(), c.h->names[(size_t)tid].c_str(), (long long)r.pos + 1);
        } else if ((st & STA_CALMD_HAS_MD) && c.update) {
            // bam_md.c:156-193
            const int nm = c.nm[i];
            const int old_nm = find_tag(aux, "NM");
            char num[48];
            snprintf(num, sizeof num, "NM:i:%d", nm);
            if (old_nm < 0) aux.push_back(num);
            else {
                const std::string &o = aux[(size_t)old_nm];
                const int old_i = o[3] == 'i' ? (int)strtoll(o.c_str() + 5, nullptr, 10) : 0;           // bam_aux2i: 0 for a non-integer tag
                if (old_i != nm) {
                    if (!c.quiet) fprintf(stderr, "[bam_fillmd1] different NM for read '%s': %d -> %d\n", r.qname.c_str(), old_i, nm);
                    aux.erase(aux.begin() + old_nm);
                    aux.push_back(num);
                }
            }
            const char *ms = c.md.data() + c.off[i];
            const size_t ml = (size_t)(c.off[i + 1] - c.off[i]);
            const int old_md = find_tag(aux, "MD");
            t = "MD:Z:"; t.append(ms, ml);
            if (old_md < 0) aux.push_back(t);
            else {
                const std::string &o = aux[(size_t)old_md];
                bool is_diff = o.size() - 5 != ml;
                for (size_t k = 0; !is_diff && k < ml; ++k) is_diff = toupper((unsigned char)o[5 + k]) != toupper((unsigned char)ms[k]);
                if (is_diff) {
                    if (!c.quiet) fprintf(stderr, "[bam_fillmd1] different MD for read '%s': '%s' -> '%.*s'\n", r.qname.c_str(), o.c_str() + 5, (int)ml, ms);
                    aux.erase(aux.begin() + old_md);
                    aux.push_back(t);
                }
            }
        }
        if (c.drop_tag && r.l_qseq > 0) {
            // bam_md.c:195-199 (bam_aux_drop_other): nothing but the RG tag stays
            const int rg = find_tag(aux, "RG");
            if (rg >= 0) { std::string keep = aux[(size_t)rg]; aux.assign(1, keep); } else aux.clear();
        }
        put_record(c, r, c.seq.data() + (boff >> 1), c.qual.data() + boff, aux);
    }
    c.batch.clear();
    return 0;
}

// the output header: the input's text, plus -- unless --no-PG -- what sam_hdr_add_pg(h, "samtools", VN, CL) adds: one @PG line per end
// of a PP chain (or one line when there is none), ID made unique
std::string output_header(const Header &h, bool no_pg, int argc, char **argv)
{
    std::string text = h.text;
    if (!text.empty() && text.back() != '\n') text += '\n';
    if (no_pg) return text;
    std::vector<std::string> ids; std::set<std::string> is_pp;
    size_t p = 0;
    while (p < text.size()) {
        size_t e = text.find('\n', p); if (e == std::string::npos) e = text.size();
        if (e - p > 4 && text.compare(p, 4, "@PG\t") == 0) {
            size_t f = p + 4;
            while (f < e) {
                size_t g = text.find('\t', f); if (g == std::string::npos || g > e) g = e;
                if (g - f > 3 && text.compare(f, 3, "ID:") == 0) ids.push_back(text.substr(f + 3, g - f - 3));
                if (g - f > 3 && text.compare(f, 3, "PP:") == 0) is_pp.insert(text.substr(f + 3, g - f - 3));
                f = g + 1;
            }
        }
        p = e + 1;
    }
    std::string cl = "samtools-amd";
    for (int i = 0; i < argc; ++i) { cl += ' '; cl += argv[i]; }
    std::set<std::string> used(ids.begin(), ids.end());
    auto fresh = [&]() { std::string id = "samtools"; for (int k = 1; used.count(id); ++k) id = "samtools." + std::to_string(k); used.insert(id); return id; };
    std::vector<std::string> ends;
    for (const std::string &id : ids) if (!is_pp.count(id)) ends.push_back(id);
    if (ends.empty()) text += "@PG\tID:" + fresh() + "\tPN:samtools\tVN:" + sta_version() + "\tCL:" + cl + "\n";
    else for (const std::string &pp : ends) text += "@PG\tID:" + fresh() + "\tPN:samtools\tPP:" + pp + "\tVN:" + sta_version() + "\tCL:" + cl + "\n";
    return text;
}

}  // namespace

extern "C" int sta_main_calmd(int argc, char **argv)
{
    Ctx c;
    int o;
    bool no_pg = false;
    int bam_level = -1;                        // -1: SAM text
    static const struct option lopts[] = { { "no-PG", no_argument, NULL, 1 }, { NULL, 0, NULL, 0 } };
    optind = 0;          // (glibc: 0 = full re-initialisation; with 1 a second in-process call resumes at a stale pointer into the PREVIOUS argv)
    while ((o = getopt_long(argc, argv, "EqQreuNhbSC:n:Ad", lopts, NULL)) >= 0) {
        switch (o) {
        case 'e': c.cp.flag |= STA_CALMD_USE_EQUAL; break;
        case 'r': c.cp.flag |= STA_CALMD_REALN; break;
        case 'A': c.cp.flag |= STA_CALMD_APPLY; break;
        case 'E': c.cp.flag |= STA_CALMD_EXTENDED; break;
        case 'q': c.cp.flag |= STA_CALMD_BIN_QUAL; break;
        case 'n': c.cp.max_nm = atoi(optarg); break;
        case 'C': c.cp.capQ = atoi(optarg); break;
        case 'd': c.drop_tag = true; break;
        case 'N': c.update = false; break;
        case 'Q': c.quiet = true; break;
        case 'h': case 'S': break;
        case 1: no_pg = true; break;
        case 'b': if (bam_level < 0) bam_level = 6; break;
        case 'u': bam_level = 0; break;
        default: fprintf(stderr, "[calmd] option -%c is not part of the engine's rows\n", o); return 1;
        }
    }
    if (argc - optind != 2) { fprintf(stderr, "usage: samtools-amd calmd [-erAEqdNQbu] [-n max_nm] [-C capQ] [--no-PG] in.bam ref.fa\n"); return 1; }
    if (sta_device_count() < 1) { fprintf(stderr, "samtools calmd: no usable HIP device (the MI355X engine has no CPU fallback)\n"); return 2; }
    std::string err;
    auto rd = AlnReader::open(argv[optind], &err);
    if (!rd) { fprintf(stderr, "samtools calmd: %s\n", err.c_str()); return 1; }
    rd->set_keep_aux(true);
    c.h = &rd->header();
    auto fa = Fasta::load(argv[optind + 1]);
    if (!fa) { fprintf(stderr, "samtools calmd: Failed to open reference file '%s'\n", argv[optind + 1]); return 1; }
    if (sta_engine_create(&c.eng, 0, nullptr) != STA_OK) { fprintf(stderr, "samtools calmd: no usable HIP device\n"); return 2; }
    {
        const std::string text = output_header(*c.h, no_pg, argc, argv);
        if (bam_level >= 0) { c.bam.reset(new BamWriter(c.out, bam_level)); if (!c.bam->header(*c.h, text)) c.io_error = true; }
        else fwrite(text.data(), 1, text.size(), c.out);
    }
    const bool realn = (c.cp.flag & STA_CALMD_REALN) != 0;
    size_t max_batch = 1 << 18;
    if (const char *e = getenv("STA_CALMD_BATCH")) max_batch = (size_t)std::max<long long>(1, atoll(e));
    int status = 0, cur_tid = -2;
    const std::string *ref = nullptr;
    unsigned skipped = 0;
    Rec r;
    int st;
    while ((st = rd->next(r)) > 0) {
        const bool new_run = r.tid != cur_tid || (!c.batch.empty() && (r.pos < c.batch.back().pos || r.pos - c.batch.front().pos > (1 << 30)));
        if (new_run || c.batch.size() >= max_batch) {
            if (flush(c, cur_tid, ref) < 0) { status = 1; break; }
        }
        if (r.tid != cur_tid) {
            cur_tid = r.tid; ref = nullptr;
            if (r.tid >= 0) {
                ref = fa->fetch(c.h->names[(size_t)r.tid]);
                sta_clear_references(c.eng);
                if (!ref) {
                    fprintf(stderr, "[bam_fillmd] fail to find sequence '%s' in the reference.\n", c.h->names[(size_t)r.tid].c_str());
                    if (realn) { status = 1; break; }
                } else if (sta_set_reference(c.eng, r.tid, ref->data(), (int64_t)ref->size(), STA_MEM_HOST) != STA_OK) { status = 1; break; }
            }
        }
        if (r.tid < 0) { put_unchanged(c, r); continue; }
        if (ref && r.l_qseq == 0) ++skipped;
        c.batch.push_back(r);
    }
    if (!status && flush(c, cur_tid, ref) < 0) status = 1;
    if (st < 0) { fprintf(stderr, "[bam_fillmd] Error reading input.\n"); status = 1; }
    if (skipped) fprintf(stderr, "[calmd] Warning: %u records skipped due to no query sequence\n", skipped);
    sta_engine_destroy(c.eng);
    if (c.bam && !c.bam->close()) c.io_error = true;
    if (fflush(c.out) != 0 || c.io_error) { fprintf(stderr, "[bam_fillmd] error when closing output file\n"); status = 1; }
    return status;
}
