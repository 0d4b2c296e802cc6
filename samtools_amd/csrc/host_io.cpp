// host_io.cpp -- SAM/BAM/FASTA/BED decoding for the drivers (see host_io.h).
#include "host_io.h"
#include <climits>
#include "host_bgzf.h"
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <zlib.h>
#include <cmath>
#include <cstring>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>
#include <cstdlib>
#include <cstdio>
#include <cctype>
#include <algorithm>
#include <fstream>

namespace sta {

namespace {
// seq_nt16_table (hts.c): IUPAC character -> 4-bit code, everything else 15
struct Nt16 {
    uint8_t t[256];
    Nt16()
    {
        memset(t, 15, sizeof t);
        const char *codes = "=ACMGRSVTWYHKDBN";
        for (int i = 0; i < 16; ++i) { t[(unsigned char)codes[i]] = (uint8_t)i; t[(unsigned char)tolower(codes[i])] = (uint8_t)i; }
        t['0'] = 1; t['1'] = 2; t['2'] = 4; t['3'] = 8;
    }
};
const Nt16 g_nt16;

inline bool is_refop(int op) { return op == 0 || op == 2 || op == 3 || op == 7 || op == 8; }
}  // namespace

struct AlnReader::Impl {
    std::vector<std::string> want;   // aux tags to format (--output-extra)
    bool keep_aux = false;           // every aux field as SAM text in Rec::auxv (record writers)

    std::unique_ptr<ByteSource> src;
    std::string path; int threads = 0;       // for seek_voffset(): the source is reopened at a block offset
    uint64_t open_coffset = 0, pulled = 0;   // where the source was (re)opened in the file, and the inflated bytes taken from it since
    bool is_bam = false;
    std::vector<uint8_t> buf; size_t bp = 0, bl = 0; bool eof = false;
    std::string line; bool have_line = false;
    std::vector<uint8_t> blk;
    int32_t n_ref = INT32_MAX;       // BAM: references in the header (records naming a refID beyond it are malformed)

    // ---- parse-ahead: a background thread decodes records into batches, next() hands them out in order ----
    struct Batch { std::vector<Rec> r; size_t n = 0; };
    static constexpr size_t BATCH = 2048, DEPTH = 4;
    std::thread th;
    std::mutex m;
    std::condition_variable cv_ready, cv_spare;
    std::deque<Batch> ready;
    std::vector<Batch> spare;
    bool started = false, stop = false, done = false;
    int final_status = 0;            // what next() returns once the batches are drained: 0 = EOF, <0 = error
    Batch cur; size_t cur_i = 0;

    bool fill()
    {
        if (eof) return false;
        if (buf.empty()) buf.resize(1 << 18);
        size_t n = src->read(buf.data(), buf.size());
        if (n == 0) { eof = true; bp = bl = 0; return false; }
        bp = 0; bl = n; pulled += n;
        return true;
    }
    size_t read(void *dst, size_t n)
    {
        uint8_t *d = (uint8_t *)dst; size_t got = 0;
        while (got < n) {
            if (bp >= bl) {
                if (n - got >= ((size_t)1 << 18) && !eof) {            // a bulk read: straight into the caller's buffer
                    const size_t k = src->read(d + got, n - got);
                    if (k == 0) { eof = true; bp = bl = 0; break; }
                    got += k; pulled += k;
                    continue;
                }
                if (!fill()) break;
            }
            size_t k = std::min(bl - bp, n - got);
            memcpy(d + got, buf.data() + bp, k);
            bp += k; got += k;
        }
        return got;
    }
    bool getline(std::string &s)
    {
        s.clear();
        bool any = false;
        for (;;) {
            if (bp >= bl && !fill()) break;
            any = true;
            uint8_t *b = buf.data() + bp;
            uint8_t *nl = (uint8_t *)memchr(b, '\n', bl - bp);
            if (nl) {
                s.append((char *)b, (size_t)(nl - b));
                bp += (size_t)(nl - b) + 1;
                if (!s.empty() && s.back() == '\r') s.pop_back();
                return true;
            }
            s.append((char *)b, bl - bp);
            bp = bl;
        }
        return any;
    }
    void shutdown()
    {
        if (!started) return;
        { std::lock_guard<std::mutex> g(m); stop = true; }
        cv_spare.notify_all(); cv_ready.notify_all();
        if (th.joinable()) th.join();
    }
};

AlnReader::~AlnReader()
{
    if (p_) { p_->shutdown(); delete p_; }
}

static void header_from_text(Header &h, bool add_refs)
{
    size_t p = 0;
    while (p < h.text.size()) {
        size_t e = h.text.find('\n', p);
        if (e == std::string::npos) e = h.text.size();
        if (e - p >= 3 && h.text.compare(p, 3, "@SQ") == 0) {
            std::string sn; int64_t ln = 0;
            size_t q = p + 3;
            while (q < e) {
                if (h.text[q] == '\t') { ++q; continue; }
                size_t f = q;
                while (q < e && h.text[q] != '\t') ++q;
                if (q - f >= 3 && h.text[f + 2] == ':') {
                    if (h.text[f] == 'S' && h.text[f + 1] == 'N') sn = h.text.substr(f + 3, q - f - 3);
                    else if (h.text[f] == 'L' && h.text[f + 1] == 'N') ln = strtoll(h.text.c_str() + f + 3, nullptr, 10);
                }
            }
            if (!sn.empty()) {
                if (add_refs) { h.index[sn] = (int)h.names.size(); h.names.push_back(sn); h.lens.push_back(ln); }
                else { int t = h.tid(sn); if (t >= 0 && ln > h.lens[(size_t)t]) h.lens[(size_t)t] = ln; }   // long references
            }
        }
        p = e + 1;
    }
}

std::unique_ptr<AlnReader> AlnReader::open(const std::string &path, std::string *err, int threads)
{
    std::unique_ptr<AlnReader> r(new AlnReader());
    r->p_ = new Impl();
    Impl &im = *r->p_;
    im.src = ByteSource::open(path, threads, err);
    if (!im.src) return nullptr;
    im.path = path; im.threads = threads;
    im.fill();
    if (im.bl >= 4 && memcmp(im.buf.data(), "BAM\1", 4) == 0) {
        im.is_bam = true; im.bp = 4;
        int32_t l_text = 0, n_ref = 0;
        if (im.read(&l_text, 4) != 4 || l_text < 0) { if (err) *err = "truncated BAM header"; return nullptr; }
        r->hdr_.text.resize((size_t)l_text);
        if (im.read(&r->hdr_.text[0], (size_t)l_text) != (size_t)l_text) { if (err) *err = "truncated BAM header"; return nullptr; }
        while (!r->hdr_.text.empty() && r->hdr_.text.back() == '\0') r->hdr_.text.pop_back();
        if (im.read(&n_ref, 4) != 4) { if (err) *err = "truncated BAM header"; return nullptr; }
        if (n_ref < 0) { if (err) *err = "invalid BAM header (negative n_ref)"; return nullptr; }
        im.n_ref = n_ref;
        for (int i = 0; i < n_ref; ++i) {
            int32_t l_name = 0, l_ref = 0;
            if (im.read(&l_name, 4) != 4) return nullptr;
            if (l_name <= 0 || l_name > (1 << 20)) { if (err) *err = "invalid BAM header (reference name length)"; return nullptr; }
            std::string nm((size_t)l_name, '\0');
            if (im.read(&nm[0], (size_t)l_name) != (size_t)l_name) return nullptr;
            while (!nm.empty() && nm.back() == '\0') nm.pop_back();
            if (im.read(&l_ref, 4) != 4) return nullptr;
            r->hdr_.index[nm] = (int)r->hdr_.names.size();
            r->hdr_.names.push_back(nm); r->hdr_.lens.push_back(l_ref);
        }
        header_from_text(r->hdr_, false);
    } else {
        while (im.getline(im.line)) {
            if (im.line.empty()) continue;
            if (im.line[0] != '@') { im.have_line = true; break; }
            r->hdr_.text += im.line; r->hdr_.text += '\n';
        }
        header_from_text(r->hdr_, true);
    }
    {
        // @HD ... SO:coordinate (the first header line)
        const std::string &t = r->hdr_.text;
        const size_t e = t.find('\n');
        const std::string first = t.substr(0, e == std::string::npos ? t.size() : e);
        r->sorted_hint_ = first.compare(0, 3, "@HD") == 0 && first.find("\tSO:coordinate") != std::string::npos;
    }
    return r;
}

bool AlnReader::seek_voffset(uint64_t voffset)
{
    Impl &im = *p_;
    if (!im.is_bam || im.started || voffset == 0 || voffset == UINT64_MAX) return false;
    std::string err;
    std::unique_ptr<ByteSource> at = ByteSource::open_bgzf_at(im.path, im.threads, voffset >> 16, &err);
    if (!at) return false;
    im.src = std::move(at);
    im.bp = im.bl = 0; im.eof = false;
    im.open_coffset = voffset >> 16; im.pulled = 0;
    sorted_hint_ = true;                       // (an index exists for coordinate-sorted files only)
    size_t skip = (size_t)(voffset & 0xffff);
    uint8_t tmp[4096];
    while (skip) { const size_t k = im.read(tmp, skip < sizeof tmp ? skip : sizeof tmp); if (!k) return false; skip -= k; }
    return true;
}

std::unique_ptr<BaiIndex> BaiIndex::load_file(const std::string &p, const std::string &bam_path)
{
    FILE *fp = fopen(p.c_str(), "rb");
    if (!fp) return nullptr;
    std::unique_ptr<BaiIndex> ix(new BaiIndex());
    bool ok = false;
    auto rd = [&](void *d, size_t n) { return fread(d, 1, n, fp) == n; };
    char magic[4]; int32_t n_ref = 0;
    if (rd(magic, 4) && memcmp(magic, "BAI\1", 4) == 0 && rd(&n_ref, 4) && n_ref >= 0 && n_ref < (1 << 26)) {
        ok = true;
        ix->lin_.resize((size_t)n_ref);
        for (int32_t t = 0; t < n_ref && ok; ++t) {
            int32_t n_bin = 0;
            if (!rd(&n_bin, 4) || n_bin < 0) { ok = false; break; }
            for (int32_t b = 0; b < n_bin && ok; ++b) {
                uint32_t bin; int32_t n_chunk = 0;
                if (!rd(&bin, 4) || !rd(&n_chunk, 4) || n_chunk < 0 || fseeko(fp, (off_t)n_chunk * 16, SEEK_CUR) != 0) ok = false;
            }
            int32_t n_intv = 0;
            if (!ok || !rd(&n_intv, 4) || n_intv < 0) { ok = false; break; }
            ix->lin_[(size_t)t].resize((size_t)n_intv);
            if (n_intv && !rd(ix->lin_[(size_t)t].data(), (size_t)n_intv * 8)) ok = false;
        }
    }
    fclose(fp);
    if (!ok) return nullptr;
    struct stat si, sd;
    ix->stale_ = stat(p.c_str(), &si) == 0 && stat(bam_path.c_str(), &sd) == 0
                 && (si.st_mtim.tv_sec < sd.st_mtim.tv_sec || (si.st_mtim.tv_sec == sd.st_mtim.tv_sec && si.st_mtim.tv_nsec < sd.st_mtim.tv_nsec));
    return ix;
}

std::unique_ptr<BaiIndex> BaiIndex::load_for(const std::string &bam_path)
{
    std::vector<std::string> cand = { bam_path + ".bai" };
    if (bam_path.size() > 4 && bam_path.compare(bam_path.size() - 4, 4, ".bam") == 0) cand.push_back(bam_path.substr(0, bam_path.size() - 4) + ".bai");
    for (const std::string &p : cand) if (auto ix = load_file(p, bam_path)) return ix;
    return nullptr;
}

uint64_t BaiIndex::start_offset(int tid, int64_t pos) const
{
    if (tid < 0 || (size_t)tid >= lin_.size()) return 0;
    if (pos < 0) pos = 0;
    // the first non-empty window at or after pos of this reference, else the first of a later one
    for (size_t t = (size_t)tid; t < lin_.size(); ++t) {
        const std::vector<uint64_t> &io = lin_[t];
        for (size_t k = t == (size_t)tid ? (size_t)(pos >> 14) : 0; k < io.size(); ++k) if (io[k]) return io[k];
    }
    return UINT64_MAX;
}

static void finish_rec(Rec &r)
{
    r.rlen = 0;
    for (uint32_t c : r.cigar) if (is_refop((int)(c & 0xf))) r.rlen += (c >> 4);
}

static void tag_from_sam(const char *val, size_t n, char type, std::string &out);

static void aux_text_from_sam(const char *f, size_t n, std::string &out);

static int parse_sam(const Header &h, std::string &line, Rec &r, const std::vector<std::string> *want, bool keep_aux)
{
    char *f[11]; size_t fl[11]; int nf = 0;
    char *p = &line[0], *e = p + line.size();
    while (nf < 11) {
        char *t = (char *)memchr(p, '\t', (size_t)(e - p));
        f[nf] = p; fl[nf] = t ? (size_t)(t - p) : (size_t)(e - p); ++nf;
        if (!t) { p = e; break; }
        p = t + 1;
    }
    if (nf < 11) return -2;
    char *aux = p;
    for (int i = 0; i < 11; ++i) f[i][fl[i]] = 0;
    r.qname.assign(f[0], fl[0]);
    r.flag = (uint16_t)strtol(f[1], nullptr, 0);
    r.tid = (fl[2] == 1 && f[2][0] == '*') ? -1 : h.tid(f[2]);
    r.pos = strtoll(f[3], nullptr, 10) - 1;
    r.mapq = (uint8_t)strtol(f[4], nullptr, 10);
    r.cigar.clear();
    if (!(fl[5] == 1 && f[5][0] == '*')) {
        char *c = f[5];
        while (*c) {
            char *q; unsigned long len = strtoul(c, &q, 10);
            const char *ops = "MIDNSHP=XB"; const char *o = *q ? strchr(ops, *q) : nullptr;
            if (!o) return -2;
            r.cigar.push_back((uint32_t)(len << 4 | (unsigned)(o - ops)));
            c = q + 1;
        }
    }
    if (fl[6] == 1 && f[6][0] == '=') r.mtid = r.tid;
    else if (fl[6] == 1 && f[6][0] == '*') r.mtid = -1;
    else r.mtid = h.tid(f[6]);
    r.mpos = strtoll(f[7], nullptr, 10) - 1;
    r.isize = strtoll(f[8], nullptr, 10);
    size_t l = (fl[9] == 1 && f[9][0] == '*') ? 0 : fl[9];
    r.l_qseq = (int32_t)l;
    r.seq.assign((l + 1) / 2, 0);
    for (size_t i = 0; i < l; ++i) r.seq[i >> 1] |= (uint8_t)(g_nt16.t[(unsigned char)f[9][i]] << ((~i & 1) << 2));
    r.qual.resize(l);
    if (fl[10] == 1 && f[10][0] == '*') std::fill(r.qual.begin(), r.qual.end(), 0xff);
    else { if (fl[10] != l) return -2; for (size_t i = 0; i < l; ++i) r.qual[i] = (uint8_t)(f[10][i] - 33); }
    r.has_bq = r.has_zq = false; r.bq.clear(); r.rg.clear(); r.mm.clear(); r.ml.clear(); r.has_ml = false;
    if (want) { r.tagtext.assign(want->size(), std::string()); r.tag_has.assign(want->size(), 0); }
    r.auxv.clear(); r.zq.clear(); r.aux_bam.clear();
    while (aux < e) {
        char *t = (char *)memchr(aux, '\t', (size_t)(e - aux));
        size_t n = t ? (size_t)(t - aux) : (size_t)(e - aux);
        if (keep_aux && n >= 5 && aux[2] == ':' && aux[4] == ':') { r.auxv.emplace_back(); aux_text_from_sam(aux, n, r.auxv.back()); if (r.auxv.back().empty()) r.auxv.pop_back(); }
        if (want && n >= 5 && aux[2] == ':' && aux[4] == ':')
            for (size_t w = 0; w < want->size(); ++w)
                if (!r.tag_has[w] && (*want)[w][0] == aux[0] && (*want)[w][1] == aux[1]) { r.tag_has[w] = 1; tag_from_sam(aux + 5, n - 5, aux[3], r.tagtext[w]); }
        if (n >= 5 && aux[2] == ':' && aux[4] == ':' && aux[3] == 'Z') {
            if (aux[0] == 'R' && aux[1] == 'G') r.rg.assign(aux + 5, n - 5);
            else if (aux[0] == 'B' && aux[1] == 'Q') { r.has_bq = true; r.bq.assign(aux + 5, aux + n); }
            else if (aux[0] == 'Z' && aux[1] == 'Q') { if (keep_aux && !r.has_zq) r.zq.assign(aux + 5, aux + n); r.has_zq = true; }
            else if (aux[0] == 'M' && (aux[1] == 'M' || aux[1] == 'm')) r.mm.assign(aux + 5, n - 5);
        }
        if (n >= 7 && aux[2] == ':' && aux[3] == 'B' && aux[4] == ':' && aux[0] == 'M' && (aux[1] == 'L' || aux[1] == 'l') && (aux[5] == 'C' || aux[5] == 'c')) {
            // ML:B:C,v,v,...
            r.has_ml = true;
            const char *q = aux + 6, *qe = aux + n;
            while (q < qe) { if (*q == ',') { ++q; continue; } char *nx; long v = strtol(q, &nx, 10); if (nx == q) break; r.ml.push_back((uint8_t)v); q = nx; }
        }
        if (!t) break;
        aux = t + 1;
    }
    finish_rec(r);
    return 1;
}

// HTSlib kputd (kstring.c; called for 'f' / 'd' aux values at bam_plcmd.c:838-840).  It is NOT printf("%g"): inside
// [0.0001, 999999] the value is scaled to a 10^-10 fixed-point integer (truncating), half a unit of the sixth significant digit is
// added (round half UP on the truncated expansion, where printf rounds the exact binary value half to even), six significant
// digits are kept and trailing zeros / a trailing point dropped; zero prints "0" / "-0"; everything else is left to "%g".
// E.g. 123456.5 -> "123457" (%g: "123456"), 12345.25 -> "12345.3" (%g: "12345.2").  Restated from the published algorithm.
static void format_kputd(double d, std::string &out)
{
    out.clear();
    if (d == 0) { out = std::signbit(d) ? "-0" : "0"; return; }
    if (d < 0) { out = "-"; d = -d; }
    if (!(d >= 0.0001 && d <= 999999)) { char b[64]; snprintf(b, sizeof b, "%g", d); out += b; return; }
    uint64_t v = (uint64_t)(d * 10000000000LL);
    // position of the sixth significant digit by the magnitude of d, as the source's ladder of comparisons does
    static const double lim[] = { 0.001, 0.01, 0.1, 1, 10, 100, 1000, 10000, 100000 };
    uint64_t half = 5;
    for (double l : lim) { if (d < l) break; half *= 10; }
    v += half;
    char dig[24]; int nd = 0;
    do { dig[nd++] = (char)('0' + v % 10); v /= 10; } while (v >= 1);        // least significant first
    std::string t;
    if (nd <= 10) {                                  // below 1: "0." + leading zeros + the first six digits
        t = "0.";
        t.append((size_t)(10 - nd), '0');
        for (int k = 0; k < 6 && k < nd; ++k) t += dig[nd - 1 - k];
    } else {                                         // nd - 10 integer digits, then the fraction, seven characters in all
        const int ni = nd - 10;
        for (int k = 0; k < ni; ++k) t += dig[nd - 1 - k];
        t += '.';
        for (int k = ni; (int)t.size() < 7; ++k) t += dig[nd - 1 - k];
        if (t.size() > 7) t.resize(7);
        if (t[6] == '.') t.resize(6);
    }
    // trailing zeros behind a decimal point go, and the point with them
    if (t.find('.') != std::string::npos) {
        size_t e = t.size();
        while (e > 1 && t[e - 1] == '0') --e;
        if (t[e - 1] == '.') --e;
        t.resize(e);
    }
    out += t;
}

// the formatter on its own (include/samtools_amd.h): what the drivers print for an 'f' / 'd' aux value
extern "C" int sta_format_aux_float(double v, char *buf, int cap)
{
    std::string t;
    format_kputd(v, t);
    if (!buf || cap <= (int)t.size()) return -1;
    memcpy(buf, t.c_str(), t.size() + 1);
    return (int)t.size();
}

// text of one aux value the way mpileup prints it (bam_plcmd.c:811-850): Z/H as is, integers in decimal, floats through kputd,
// A as the character, anything else (B arrays) as '*'
static void tag_from_sam(const char *val, size_t n, char type, std::string &out)
{
    if (type == 'Z' || type == 'H' || type == 'A') out.assign(val, n);
    else if (type == 'i') { char b[32]; snprintf(b, sizeof b, "%lld", strtoll(std::string(val, n).c_str(), nullptr, 10)); out = b; }
    else if (type == 'f') format_kputd((double)strtof(std::string(val, n).c_str(), nullptr), out);
    else out = "*";
}

// ---- an aux field as sam_format1 writes it ("TG:T:value", HTSlib sam.c; SAM spec 1.5): integers of every width as `i`, floats
// through kputd, B arrays comma separated behind their subtype.  Kept per record only for the commands that write records (calmd).
static void aux_put_ll(std::string &o, long long v) { char b[32]; snprintf(b, sizeof b, "%lld", v); o += b; }
static void aux_put_fl(std::string &o, double v) { std::string t; format_kputd(v, t); o += t; }

// from SAM text: what the field reads after a round trip through the binary record (sam_parse1 + sam_format1)
static void aux_text_from_sam(const char *f, size_t n, std::string &out)
{
    const char type = f[3];
    const std::string v(f + 5, n - 5);
    out.assign(f, 5);
    if (type == 'i') aux_put_ll(out, strtoll(v.c_str(), nullptr, 10));
    else if (type == 'f') aux_put_fl(out, (double)strtof(v.c_str(), nullptr));
    else if (type == 'd') aux_put_fl(out, strtod(v.c_str(), nullptr));
    else if (type == 'A') out += v.empty() ? ' ' : v[0];
    else if (type == 'Z' || type == 'H') out += v;
    else if (type == 'B' && !v.empty() && strchr("cCsSiIf", v[0])) {
        const char sub = v[0];
        out += sub;
        const char *p = v.c_str() + 1;
        while (*p) {
            if (*p == ',') { ++p; continue; }
            char *q;
            out += ',';
            if (sub == 'f') { float x = strtof(p, &q); aux_put_fl(out, (double)x); }
            else {
                long long x = strtoll(p, &q, 10);
                switch (sub) {     // stored at the subtype's width
                case 'c': x = (int8_t)x; break; case 'C': x = (uint8_t)x; break; case 's': x = (int16_t)x; break;
                case 'S': x = (uint16_t)x; break; case 'i': x = (int32_t)x; break; default: x = (uint32_t)x; break;
                }
                aux_put_ll(out, x);
            }
            if (q == p) { out.pop_back(); break; }
            p = q;
        }
    } else out.clear();
}

// from the BAM encoding: tag = the two name bytes, p = the value (behind the type byte t)
static void aux_text_from_bam(const uint8_t *tag, const uint8_t *p, int t, std::string &out)
{
    out.assign((const char *)tag, 2); out += ':';
    auto num = [&](long long v) { out += "i:"; aux_put_ll(out, v); };
    switch (t) {
    case 'A': out += "A:"; out += (char)p[0]; break;
    case 'c': num((int8_t)p[0]); break;
    case 'C': num(p[0]); break;
    case 's': { int16_t v; memcpy(&v, p, 2); num(v); break; }
    case 'S': { uint16_t v; memcpy(&v, p, 2); num(v); break; }
    case 'i': { int32_t v; memcpy(&v, p, 4); num(v); break; }
    case 'I': { uint32_t v; memcpy(&v, p, 4); num(v); break; }
    case 'f': { float v; memcpy(&v, p, 4); out += "f:"; aux_put_fl(out, (double)v); break; }
    case 'd': { double v; memcpy(&v, p, 8); out += "d:"; aux_put_fl(out, v); break; }
    case 'Z': case 'H': out += (char)t; out += ':'; out += (const char *)p; break;
    case 'B': {
        const int sub = p[0]; uint32_t cnt; memcpy(&cnt, p + 1, 4);
        const uint8_t *q = p + 5;
        out += "B:"; out += (char)sub;
        for (uint32_t k = 0; k < cnt; ++k) {
            out += ',';
            switch (sub) {
            case 'c': aux_put_ll(out, (int8_t)q[0]); q += 1; break;
            case 'C': aux_put_ll(out, q[0]); q += 1; break;
            case 's': { int16_t v; memcpy(&v, q, 2); aux_put_ll(out, v); q += 2; break; }
            case 'S': { uint16_t v; memcpy(&v, q, 2); aux_put_ll(out, v); q += 2; break; }
            case 'i': { int32_t v; memcpy(&v, q, 4); aux_put_ll(out, v); q += 4; break; }
            case 'I': { uint32_t v; memcpy(&v, q, 4); aux_put_ll(out, v); q += 4; break; }
            default: { float v; memcpy(&v, q, 4); aux_put_fl(out, (double)v); q += 4; break; }
            }
        }
        break; }
    }
}

void format_sam_record(const Header &h, const Rec &r, const uint8_t *seq4, const uint8_t *qual, const std::vector<std::string> &aux, std::string &s)
{
    static const char nt[] = "=ACMGRSVTWYHKDBN";
    char num[96];
    s.clear();
    s += r.qname;
    snprintf(num, sizeof num, "\t%d\t", (int)r.flag); s += num;
    s += (r.tid >= 0 && r.tid < h.nref()) ? h.names[(size_t)r.tid].c_str() : "*";
    snprintf(num, sizeof num, "\t%lld\t%d\t", (long long)r.pos + 1, (int)r.mapq); s += num;
    if (r.cigar.empty()) s += '*';
    else for (uint32_t cg : r.cigar) { snprintf(num, sizeof num, "%u%c", cg >> 4, "MIDNSHP=XB"[cg & 0xf]); s += num; }
    s += '\t';
    if (r.mtid < 0) s += '*';
    else if (r.mtid == r.tid) s += '=';
    else s += r.mtid < h.nref() ? h.names[(size_t)r.mtid].c_str() : "*";
    snprintf(num, sizeof num, "\t%lld\t%lld\t", (long long)r.mpos + 1, (long long)r.isize); s += num;
    if (r.l_qseq == 0) s += "*\t*";
    else {
        for (int k = 0; k < r.l_qseq; ++k) s += nt[(seq4[(size_t)k >> 1] >> ((~k & 1) << 2)) & 0xf];
        s += '\t';
        if (qual[0] == 0xff) s += '*'; else for (int k = 0; k < r.l_qseq; ++k) s += (char)(qual[(size_t)k] + 33);
    }
    for (const std::string &a : aux) { s += '\t'; s += a; }
    s += '\n';
}

static int aux_size(int t) { switch (t) { case 'A': case 'c': case 'C': return 1; case 's': case 'S': return 2; case 'i': case 'I': case 'f': return 4; case 'd': return 8; } return 0; }

static int parse_bam(AlnReader::Impl &im, Rec &r);

int AlnReader::next_raw(Rec &r)
{
    Impl &im = *p_;
    if (im.is_bam) return parse_bam(im, r);
    if (im.have_line) im.have_line = false;
    else { do { if (!im.getline(im.line)) return 0; } while (im.line.empty()); }
    return parse_sam(hdr_, im.line, r, im.want.empty() ? nullptr : &im.want, im.keep_aux);
}

static int parse_bam_mem(const uint8_t *b, int32_t bs, const std::vector<std::string> &wanted, Rec &r, int32_t n_ref, bool keep_aux);

static int parse_bam(AlnReader::Impl &im, Rec &r)
{
    int32_t bs = 0;
    size_t n = im.read(&bs, 4);
    if (n == 0) return 0;
    if (n != 4 || bs < 32) return -2;
    if (im.blk.size() < (size_t)bs) im.blk.resize((size_t)bs * 2);
    if (im.read(im.blk.data(), (size_t)bs) != (size_t)bs) return -2;
    return parse_bam_mem(im.blk.data(), bs, im.want, r, im.n_ref, im.keep_aux);
}

// one BAM alignment record (SAM spec 4.2) of bs bytes, block_size prefix already consumed
static int parse_bam_mem(const uint8_t *b, int32_t bs, const std::vector<std::string> &wanted, Rec &r, int32_t n_ref, bool keep_aux)
{
    int32_t refID, pos, l_seq, nref, npos, tlen; uint16_t n_cig, flag;
    memcpy(&refID, b, 4); memcpy(&pos, b + 4, 4);
    uint8_t l_rn = b[8]; r.mapq = b[9];
    memcpy(&n_cig, b + 12, 2); memcpy(&flag, b + 14, 2); memcpy(&l_seq, b + 16, 4);
    memcpy(&nref, b + 20, 4); memcpy(&npos, b + 24, 4); memcpy(&tlen, b + 28, 4);
    size_t need = 32 + (size_t)l_rn + 4 * (size_t)n_cig + ((size_t)l_seq + 1) / 2 + (size_t)l_seq;
    if ((size_t)bs < need || l_seq < 0) return -2;
    // sam.c bam_read1: ids beyond the header are an error, not data
    if (refID < -1 || refID >= n_ref || nref < -1 || nref >= n_ref) return -2;
    size_t o = 32;
    r.qname.assign((const char *)b + o, l_rn ? l_rn - 1u : 0u); o += l_rn;
    r.cigar.resize(n_cig); if (n_cig) memcpy(r.cigar.data(), b + o, 4 * (size_t)n_cig); o += 4 * (size_t)n_cig;
    r.seq.assign(b + o, b + o + ((size_t)l_seq + 1) / 2); o += ((size_t)l_seq + 1) / 2;
    r.qual.assign(b + o, b + o + (size_t)l_seq); o += (size_t)l_seq;
    r.tid = refID; r.pos = pos; r.flag = flag; r.mtid = nref; r.mpos = npos; r.isize = tlen; r.l_qseq = l_seq;
    r.has_bq = r.has_zq = false; r.bq.clear(); r.rg.clear(); r.mm.clear(); r.ml.clear(); r.has_ml = false;
    const uint8_t *p = b + o, *e = b + bs;
    const bool want = !wanted.empty();
    if (want) { r.tagtext.assign(wanted.size(), std::string()); r.tag_has.assign(wanted.size(), 0); }
    const uint8_t *cg = nullptr; uint32_t cg_n = 0;          // CG:B,I: the real CIGAR of a read with more than 65535 operations
    r.auxv.clear(); r.zq.clear(); r.aux_bam.clear(); r.cigar_from_tag = false;
    int cg_field = -1;
    while (p + 3 <= e) {
        int t = p[2]; const uint8_t *tag = p; p += 3;
        // size of the value, checked against the record end before anything is read (a truncated field is a malformed record)
        size_t vlen;
        if (t == 'Z' || t == 'H') {
            const uint8_t *q = (const uint8_t *)memchr(p, 0, (size_t)(e - p));
            if (!q) return -2;
            vlen = (size_t)(q - p) + 1;
        } else if (t == 'B') {
            if (p + 5 > e) return -2;
            int sz = aux_size(p[0]); uint32_t cnt; memcpy(&cnt, p + 1, 4);
            if (!sz || (uint64_t)sz * cnt > (uint64_t)(e - p - 5)) return -2;
            vlen = 5 + (size_t)sz * cnt;
            if (tag[0] == 'C' && tag[1] == 'G' && (p[0] == 'I' || p[0] == 'i') && !cg) { cg = p + 5; cg_n = cnt; cg_field = (int)r.auxv.size(); }      // bam_aux_get: the first CG
            if (tag[0] == 'M' && (tag[1] == 'L' || tag[1] == 'l') && sz == 1) { r.has_ml = true; r.ml.assign(p + 5, p + 5 + cnt); }
        } else {
            int sz = aux_size(t);
            if (!sz || p + sz > e) return -2;
            vlen = (size_t)sz;
        }
        if (want)
            for (size_t w = 0; w < wanted.size(); ++w)
                if (!r.tag_has[w] && wanted[w][0] == (char)tag[0] && wanted[w][1] == (char)tag[1]) {
                    r.tag_has[w] = 1;
                    std::string &out = r.tagtext[w];
                    char nb[64];
                    if (t == 'Z' || t == 'H') out.assign((const char *)p, vlen - 1);
                    else if (t == 'A') out.assign(1, (char)p[0]);
                    else if (t == 'c') { snprintf(nb, sizeof nb, "%d", (int)(int8_t)p[0]); out = nb; }
                    else if (t == 'C') { snprintf(nb, sizeof nb, "%d", (int)p[0]); out = nb; }
                    else if (t == 's') { int16_t v; memcpy(&v, p, 2); snprintf(nb, sizeof nb, "%d", (int)v); out = nb; }
                    else if (t == 'S') { uint16_t v; memcpy(&v, p, 2); snprintf(nb, sizeof nb, "%d", (int)v); out = nb; }
                    else if (t == 'i') { int32_t v; memcpy(&v, p, 4); snprintf(nb, sizeof nb, "%d", v); out = nb; }
                    else if (t == 'I') { uint32_t v; memcpy(&v, p, 4); snprintf(nb, sizeof nb, "%u", v); out = nb; }
                    else if (t == 'f') { float v; memcpy(&v, p, 4); format_kputd((double)v, out); }
                    else if (t == 'd') { double v; memcpy(&v, p, 8); format_kputd(v, out); }
                    else out = "*";
                }
        if (keep_aux) {
            r.auxv.emplace_back(); aux_text_from_bam(tag, p, t, r.auxv.back());
            r.aux_bam.emplace_back(r.auxv.back(), std::string((const char *)tag, 3 + vlen));
        }
        if (t == 'Z') {
            if (tag[0] == 'R' && tag[1] == 'G') r.rg.assign((const char *)p, vlen - 1);
            else if (tag[0] == 'B' && tag[1] == 'Q') { r.has_bq = true; r.bq.assign(p, p + vlen - 1); }
            else if (tag[0] == 'Z' && tag[1] == 'Q') { if (keep_aux && !r.has_zq) r.zq.assign(p, p + vlen - 1); r.has_zq = true; }
            else if (tag[0] == 'M' && (tag[1] == 'M' || tag[1] == 'm')) r.mm.assign((const char *)p, vlen - 1);
        }
        p += vlen;
    }
    // SAM spec 4.2.2 / sam.c bam_tag2cigar: a CIGAR of more than 65535 operations is stored in CG:B,I and the record carries
    // the placeholder <l_seq>S<ref span>N
    // (bam_tag2cigar's conditions: a mapped record whose first op is <l_seq>S, CG of subtype I or i, at least as long as the
    // placeholder and below 2^29 ops; the second placeholder op is not looked at)
    if (cg && n_cig >= 1 && refID >= 0 && pos >= 0 && (r.cigar[0] & 0xf) == 4 && (int64_t)(r.cigar[0] >> 4) == (int64_t)l_seq
        && cg_n >= (uint32_t)n_cig && cg_n < (1u << 29)) {
        r.cigar.resize(cg_n);
        r.cigar_from_tag = true;
        if (cg_n) memcpy(r.cigar.data(), cg, 4 * (size_t)cg_n);
        if (keep_aux && cg_field >= 0 && (size_t)cg_field < r.auxv.size()) r.auxv.erase(r.auxv.begin() + cg_field);     // bam_tag2cigar removes the tag
    }
    finish_rec(r);
    return 1;
}

// ---- raw record groups for the chunked reader (host_chunk.cpp): whole BAM records (with their block_size prefix) or whole
// SAM lines ('\n' terminated), about `target` bytes per call; appended to `out`.  Not to be mixed with next().
int AlnReader::raw_group(pvector<uint8_t> &out, size_t target, int64_t *n_records)
{
    Impl &im = *p_;
    int64_t nrec = 0;
    if (im.is_bam) {
        // `target` bytes in one go, then the block_size chain inside them; the record that straddles the end is completed
        // (the group = the whole records up to the first boundary at or beyond `target`).  The buffer is NOT cleared first: a caller's
        // vector keeps its size from the last group (about `target`), so resize() does not zero a megabyte per call -- this runs on
        // the one thread at a time that may cut the stream, and that memset was as expensive as the copy itself
        out.resize(target);
        size_t got = im.read(out.data(), target), o = 0;
        out.resize(got);
        while (o < got) {
            if (got - o < 4) {
                const size_t need = 4 - (got - o);
                out.resize(got + need);
                if (im.read(&out[got], need) != need) return -2;
                got += need;
            }
            int32_t bs; memcpy(&bs, &out[o], 4);
            if (bs < 32) return -2;
            const size_t end = o + 4 + (size_t)bs;
            if (end > got) {
                out.resize(end);
                if (im.read(&out[got], end - got) != end - got) return -2;
                got = end;
            }
            ++nrec; o = end;
        }
    } else {
        out.clear();
        while (out.size() < target) {
            if (im.have_line) im.have_line = false;
            else if (!im.getline(im.line)) break;
            if (im.line.empty()) continue;
            out.insert(out.end(), im.line.begin(), im.line.end());
            out.push_back('\n');
            ++nrec;
        }
    }
    if (im.src && im.src->failed()) return -1;
    if (n_records) *n_records = nrec;
    return nrec ? 1 : 0;
}

bool AlnReader::is_bam() const { return p_->is_bam; }

// Where the next unread record starts, for a reader that maps the file itself (host_chunk.cpp): the compressed offset the byte source
// was opened at and the inflated bytes consumed since.  false: not a BAM file on disk, or records have already been handed out.
bool AlnReader::record_stream_position(std::string *path, uint64_t *coffset, uint64_t *consumed) const
{
    const Impl &im = *p_;
    if (!im.is_bam || im.started || im.path.empty() || im.path == "-") return false;
    *path = im.path; *coffset = im.open_coffset; *consumed = im.pulled - (im.bl - im.bp);
    return true;
}

// the caller reads the file through its own mapping from here on: the stream source and its threads are let go
void AlnReader::release_source()
{
    Impl &im = *p_;
    im.src.reset(); im.eof = true; im.bp = im.bl = 0;
}

// parses one record of a group returned by raw_group(); *used = bytes consumed.  1 = record, <0 = malformed
int AlnReader::parse_raw(const uint8_t *p, size_t avail, size_t *used, Rec &r, std::string &scratch) const
{
    const Impl &im = *p_;
    if (im.is_bam) {
        if (avail < 4) return -2;
        int32_t bs; memcpy(&bs, p, 4);
        if (bs < 32 || (size_t)bs + 4 > avail) return -2;
        *used = (size_t)bs + 4;
        return parse_bam_mem(p + 4, bs, im.want, r, im.n_ref, im.keep_aux);
    }
    const uint8_t *nl = (const uint8_t *)memchr(p, '\n', avail);
    if (!nl) return -2;
    scratch.assign((const char *)p, (size_t)(nl - p));
    *used = (size_t)(nl - p) + 1;
    return parse_sam(hdr_, scratch, r, im.want.empty() ? nullptr : &im.want, im.keep_aux);
}

void AlnReader::set_wanted_tags(const std::vector<std::string> &tags) { p_->want = tags; }
void AlnReader::set_keep_aux(bool on) { p_->keep_aux = on; }

// parser thread: decode + region filter, one batch at a time
void AlnReader::parse_ahead()
{
    Impl &im = *p_;
    for (;;) {
        Impl::Batch b;
        {
            std::unique_lock<std::mutex> lk(im.m);
            im.cv_spare.wait(lk, [&] { return im.stop || !im.spare.empty() || im.ready.size() < Impl::DEPTH; });
            if (im.stop) return;
            if (!im.spare.empty()) { b = std::move(im.spare.back()); im.spare.pop_back(); }
        }
        if (b.r.size() < Impl::BATCH) b.r.resize(Impl::BATCH);
        b.n = 0;
        int status = 1;
        while (b.n < Impl::BATCH) {
            Rec &r = b.r[b.n];
            status = next_raw(r);
            if (status <= 0) break;
            if (past_region(r)) { status = 0; break; }    // sorted input: nothing further can overlap the region
            if (has_reg_ && (r.tid != rtid_ || r.pos >= rend_ || r.endpos() <= rbeg_)) continue;
            r.accepted = false;
            ++b.n;
        }
        if (status < 0 || (im.src && im.src->failed())) status = status < 0 ? status : -1;
        std::lock_guard<std::mutex> g(im.m);
        if (b.n) im.ready.push_back(std::move(b));
        if (status <= 0) { im.final_status = status; im.done = true; im.cv_ready.notify_all(); return; }
        im.cv_ready.notify_one();
    }
}

int AlnReader::next(Rec &r)
{
    Impl &im = *p_;
    if (!im.started) { im.started = true; im.th = std::thread([this] { parse_ahead(); }); }
    for (;;) {
        if (im.cur_i < im.cur.n) {
            std::swap(r, im.cur.r[im.cur_i++]);       // the consumer's old buffers go back into the batch for reuse
            if (on_record) on_record(r);
            return 1;
        }
        std::unique_lock<std::mutex> lk(im.m);
        if (!im.cur.r.empty()) { im.cur.n = 0; im.spare.push_back(std::move(im.cur)); im.cur = Impl::Batch(); im.cv_spare.notify_one(); }
        im.cur_i = 0;
        im.cv_ready.wait(lk, [&] { return !im.ready.empty() || im.done; });
        if (im.ready.empty()) return im.final_status;
        im.cur = std::move(im.ready.front()); im.ready.pop_front();
        im.cv_spare.notify_one();
    }
}

// HTSlib hts_parse_decimal (what hts_parse_reg reads coordinates with): digits with thousands commas, an optional fraction and
// exponent, and the suffixes k / M / G -- "1M", "1.5k", "2,500,000", "1e6".  *endp = first character not consumed.
static long long parse_decimal(const char *s, const char **endp)
{
    long long n = 0; int decimals = 0, e = 0; bool digits = false;
    const char *p = s;
    while (isspace((unsigned char)*p)) ++p;
    const bool neg = *p == '-' && isdigit((unsigned char)p[1]);
    if (*p == '+' || neg) ++p;
    for (; isdigit((unsigned char)*p) || (*p == ',' && digits); ++p) if (*p != ',') { if (n < LLONG_MAX / 10 - 1) n = n * 10 + (*p - '0'); digits = true; }
    if (*p == '.') { ++p; for (; isdigit((unsigned char)*p); ++p) { if (n < LLONG_MAX / 10 - 1) { n = n * 10 + (*p - '0'); ++decimals; } digits = true; } }
    if (!digits) { *endp = s; return 0; }
    if ((*p == 'e' || *p == 'E') && (isdigit((unsigned char)p[1]) || ((p[1] == '+' || p[1] == '-') && isdigit((unsigned char)p[2])))) { char *q; e = (int)strtol(p + 1, &q, 10); p = q; }
    switch (*p) { case 'k': case 'K': e += 3; ++p; break; case 'm': case 'M': e += 6; ++p; break; case 'g': case 'G': e += 9; ++p; break; }
    e -= decimals;
    while (e > 0) { if (n < LLONG_MAX / 10 - 1) n *= 10; --e; }       /* (absurd coordinates saturate instead of overflowing) */
    while (e < 0) { n /= 10; ++e; }
    *endp = p;
    return neg ? -n : n;
}

bool parse_region(const Header &h, const std::string &reg, int *tid, int64_t *beg, int64_t *end)
{
    *beg = 0; *end = INT64_MAX;
    int t = h.tid(reg);
    if (t >= 0) { *tid = t; return true; }
    size_t colon = reg.rfind(':');
    if (colon == std::string::npos) return false;
    t = h.tid(reg.substr(0, colon));
    if (t < 0) return false;
    const std::string num = reg.substr(colon + 1);
    const char *q;
    long long b = parse_decimal(num.c_str(), &q);
    if (q == num.c_str()) { if (*q == '-') b = 1; else return false; }
    if (b < 0 && !*q) { *tid = t; *beg = 0; *end = -b; return true; }            // hts_parse_region: chr:-100 is chr:1-100
    long long e = INT64_MAX;
    if (*q == '-') { if (q[1]) { const char *q2; e = parse_decimal(q + 1, &q2); if (q2 == q + 1 || *q2) return false; } }
    else if (*q) return false;
    *tid = t; *beg = b > 0 ? b - 1 : 0; *end = e;
    return *beg < *end;
}

// ---- index-driven loading: the .fai line of a contig is NAME, LENGTH, OFFSET of its first base, LINEBASES, LINEWIDTH (faidx format) ----
struct Fasta::Lazy {
    struct Ent { int64_t len = 0, offset = 0; int32_t linebases = 0, linewidth = 0; int state = 0; };   // state: 0 not read, 1 being read, 2 there
    std::string path;
    int fd = -1;
    int64_t file_size = 0;
    std::vector<Ent> ent;
    std::mutex m; std::condition_variable cv;
    std::thread pre;                       // the one read-ahead in flight
    bool broken = false;                   // the file did not fit its index: everything was parsed instead (seqs_ replaced under m)
    bool spawning = false;                 // a thread is replacing `pre`
    ~Lazy() { if (pre.joinable()) pre.join(); if (fd >= 0) close(fd); }
    // the bases of entry e into out; false if the file does not look the way the index says
    bool read_contig(const Ent &e, std::string &out) const
    {
        out.clear();
        if (e.len == 0) return true;
        if (e.linebases <= 0 || e.linewidth < e.linebases || e.offset <= 0) return false;
        const int64_t full = e.len / e.linebases, rest = e.len % e.linebases;
        const int64_t bytes = full * e.linewidth + rest;
        if (e.offset + bytes - (rest == 0 ? e.linewidth - e.linebases : 0) > file_size) return false;
        char before = 0;
        if (pread(fd, &before, 1, (off_t)(e.offset - 1)) != 1 || before != '\n') return false;      // the bases start behind the name line
        out.resize((size_t)e.len);
        std::vector<char> buf((size_t)std::min<int64_t>(bytes, (int64_t)e.linewidth * 65536));
        int64_t done = 0, got_bases = 0;
        while (done < bytes) {
            // whole lines per read, so that a line never straddles two reads
            const int64_t want = std::min<int64_t>((int64_t)buf.size(), bytes - done);
            int64_t have = 0;
            while (have < want) { const ssize_t k = pread(fd, buf.data() + have, (size_t)(want - have), (off_t)(e.offset + done + have)); if (k <= 0) break; have += k; }
            if (have <= 0) break;
            for (int64_t o = 0; o < have; o += e.linewidth) {
                const int64_t n = std::min<int64_t>(std::min<int64_t>(e.linebases, have - o), e.len - got_bases);
                memcpy(&out[(size_t)got_bases], buf.data() + o, (size_t)n);
                got_bases += n;
            }
            done += have;
        }
        if (got_bases != e.len) return false;
        // what faidx would have kept: printable characters only; a line break or a name line inside means the index is stale
        unsigned bad = 0;
        for (size_t i = 0; i < out.size(); ++i) bad |= (unsigned)((unsigned char)(out[i] - 33) >= 94u) | (unsigned)(out[i] == '>');
        return bad == 0;
    }
};

Fasta::~Fasta() { delete lz_; }

const std::string *Fasta::fetch(const std::string &name) const
{
    auto it = idx_.find(name);
    if (it == idx_.end()) return nullptr;
    const size_t i = it->second;
    if (!lazy_) return &seqs_[i];
    Lazy &z = *lz_;
    std::unique_lock<std::mutex> lk(z.m);
    auto load_now = [&](size_t k) {          // called with the lock held and ent[k].state == 1; returns with the lock held
        std::string tmp;
        lk.unlock();
        const bool ok = z.read_contig(z.ent[k], tmp);
        lk.lock();
        if (ok && !z.broken) seqs_[k].swap(tmp);
        else if (!z.broken) {
            // the file is not what the index describes: parse all of it once, the way a run without an index does
            z.broken = true;
            lk.unlock();
            std::unique_ptr<Fasta> whole = load_whole(z.path);
            lk.lock();
            // (contigs already handed out keep their buffers: a device thread may still be reading one through the pointer it was given)
            for (size_t j = 0; j < names_.size(); ++j) {
                if (j < z.ent.size() && z.ent[j].state == 2) continue;
                const std::string *s2 = whole ? whole->fetch(names_[j]) : nullptr;
                if (s2) seqs_[j] = *s2; else seqs_[j].clear();
            }
            for (auto &e : z.ent) e.state = 2;
        }
        z.ent[k].state = 2;
        z.cv.notify_all();
    };
    while (z.ent[i].state == 1) z.cv.wait(lk);
    if (z.ent[i].state == 0) { z.ent[i].state = 1; load_now(i); }
    // the next contig of the file on a thread of its own (one at a time)
    if (i + 1 < z.ent.size() && z.ent[i + 1].state == 0 && !z.broken && !z.spawning) {
        z.spawning = true;                                   // one thread at a time retires the old read-ahead and starts the next
        std::thread old = std::move(z.pre);
        if (old.joinable()) { lk.unlock(); old.join(); lk.lock(); }
        if (z.ent[i + 1].state == 0 && !z.broken) {
            z.ent[i + 1].state = 1;
            const size_t k = i + 1;
            z.pre = std::thread([this, k] {
                Lazy &zz = *lz_;
                std::string tmp;
                const bool ok = zz.read_contig(zz.ent[k], tmp);
                std::lock_guard<std::mutex> g(zz.m);
                if (ok && !zz.broken) { seqs_[k].swap(tmp); zz.ent[k].state = 2; }
                else if (zz.ent[k].state == 1) zz.ent[k].state = 0;        // let the thread that needs it deal with the mismatch
                zz.cv.notify_all();
            });
        }
        z.spawning = false;
    }
    return &seqs_[i];
}

std::unique_ptr<Fasta> Fasta::load(const std::string &path)
{
    // index first: plain file with a readable <path>.fai
    if (!getenv("STA_FASTA_WHOLE")) {
        FILE *fi = fopen((path + ".fai").c_str(), "r");
        int fd = fi ? open(path.c_str(), O_RDONLY) : -1;
        unsigned char magic[2] = { 0, 0 };
        if (fi && fd >= 0 && pread(fd, magic, 2, 0) == 2 && !(magic[0] == 0x1f && magic[1] == 0x8b)) {
            std::unique_ptr<Fasta> fa(new Fasta());
            fa->lz_ = new Lazy();
            Lazy &z = *fa->lz_;
            z.path = path; z.fd = fd;
            { struct stat st; if (fstat(fd, &st) == 0) z.file_size = (int64_t)st.st_size; }
            char line[8192];
            bool ok = true;
            while (fgets(line, sizeof line, fi)) {
                char *tab = strchr(line, '\t');
                if (!tab) { if (line[0] == '\n' || line[0] == 0) continue; ok = false; break; }
                Lazy::Ent e; long long a, b; int c, d;
                if (sscanf(tab + 1, "%lld\t%lld\t%d\t%d", &a, &b, &c, &d) != 4 || a < 0 || b < 0) { ok = false; break; }
                e.len = a; e.offset = b; e.linebases = c; e.linewidth = d;
                std::string name(line, (size_t)(tab - line));
                if (fa->idx_.count(name)) continue;                 // faidx keeps the first of equal names
                fa->idx_[name] = fa->names_.size();
                fa->names_.push_back(name);
                z.ent.push_back(e);
            }
            fclose(fi); fi = nullptr;
            // does the file look the way the index says?  It starts with the first name, and a line ends right in front of every contig's
            // first base and no contig reaches beyond the file (one byte read per contig); anything else: parse the file instead
            if (ok && !fa->names_.empty()) {
                std::vector<char> head(fa->names_[0].size() + 1);
                if (pread(fd, head.data(), head.size(), 0) != (ssize_t)head.size() || head[0] != '>' || memcmp(head.data() + 1, fa->names_[0].data(), fa->names_[0].size()) != 0) ok = false;
                for (size_t k = 0; ok && k < z.ent.size(); ++k) {
                    const Lazy::Ent &e = z.ent[k];
                    char before = 0;
                    if (e.offset <= 0 || e.offset > z.file_size || pread(fd, &before, 1, (off_t)(e.offset - 1)) != 1 || before != '\n') ok = false;
                    else if (e.len > 0 && (e.linebases <= 0 || e.linewidth < e.linebases || e.offset + (e.len / e.linebases) * e.linewidth + e.len % e.linebases - (e.linewidth - e.linebases) > z.file_size)) ok = false;
                }
            }
            if (ok && !fa->names_.empty()) { fa->seqs_.resize(fa->names_.size()); fa->lazy_ = true; return fa; }
            // (the Lazy object owns fd and closes it)
            fd = -1;
        }
        if (fi) fclose(fi);
        if (fd >= 0) close(fd);
    }
    return load_whole(path);
}

std::unique_ptr<Fasta> Fasta::load_whole(const std::string &path)
{
    gzFile fp = gzopen(path.c_str(), "rb");
    if (!fp) return nullptr;
    gzbuffer(fp, 1 << 20);
    std::unique_ptr<Fasta> fa(new Fasta());
    // Line at a time inside 4 MiB reads: a sequence line is appended whole (faidx keeps the isgraph() characters; the scan for
    // anything else runs over the line once and almost never finds one).  A genome-sized FASTA (gigabytes) used to take longer
    // to parse byte by byte than the whole device pipeline took to pile it.
    std::vector<char> buf(4 << 20);
    std::string name; bool in_name = false, bol = true;
    uint64_t file_bytes = 0, consumed = 0;
    { struct stat st; if (gzdirect(fp) && stat(path.c_str(), &st) == 0) file_bytes = (uint64_t)st.st_size; }
    int n;
    for (; (n = gzread(fp, buf.data(), (unsigned)buf.size())) > 0; consumed += (uint64_t)n) {
        const char *p = buf.data(), *e = p + n;
        while (p < e) {
            const char *nl = (const char *)memchr(p, '\n', (size_t)(e - p));
            const char *le = nl ? nl : e;                 // this piece of the line (a line may straddle reads)
            if (in_name) name.append(p, le);
            else if (bol && p < le && *p == '>') { in_name = true; name.assign(p + 1, le); }
            else if (p < le && !fa->seqs_.empty()) {
                std::string &sq = fa->seqs_.back();
                // isgraph() in the C locale = 33 .. 126; one pass that the compiler can vectorise
                unsigned bad = 0;
                for (const char *q = p; q < le; ++q) bad |= (unsigned)((unsigned char)(*q - 33) >= 94u);
                if (!bad) sq.append(p, le);
                else for (const char *q = p; q < le; ++q) if ((unsigned char)(*q - 33) < 94u) sq += *q;
            }
            if (p < le) bol = false;
            if (nl) {
                if (in_name) {
                    in_name = false;
                    size_t k = 0; while (k < name.size() && !isspace((unsigned char)name[k])) ++k;
                    name.resize(k);
                    fa->idx_[name] = fa->seqs_.size();
                    fa->names_.push_back(name);
                    fa->seqs_.emplace_back();
                    // a contig cannot be longer than what is left of the file: one reservation instead of repeated doubling and
                    // copying of a string that reaches hundreds of megabytes (untouched pages cost nothing; plain files only)
                    if (file_bytes > consumed + (uint64_t)(nl - buf.data())) fa->seqs_.back().reserve((size_t)std::min<uint64_t>(file_bytes - consumed - (uint64_t)(nl - buf.data()), (uint64_t)1 << 32));
                }
                bol = true;
                p = nl + 1;
            } else p = e;
        }
    }
    gzclose(fp);
    return fa;
}

std::unique_ptr<Bed> Bed::load(const std::string &path)
{
    gzFile fp = gzopen(path.c_str(), "rb");
    if (!fp) return nullptr;
    std::unique_ptr<Bed> b(new Bed());
    std::unordered_map<std::string, std::vector<std::pair<int64_t, int64_t>>> raw;
    std::vector<char> line(1 << 16);
    while (gzgets(fp, line.data(), (int)line.size())) {
        char *ref = line.data();
        size_t l = strlen(ref);
        while (l && (ref[l - 1] == '\n' || ref[l - 1] == '\r')) ref[--l] = 0;
        while (*ref && isspace((unsigned char)*ref)) ++ref;
        if (!*ref || *ref == '#') continue;
        char *re = ref; while (*re && !isspace((unsigned char)*re)) ++re;
        unsigned long long beg = 0, end = 0; int num = 0;
        if (*re) { *re = 0; num = sscanf(re + 1, "%llu %llu", &beg, &end); }
        if (num == 1) end = beg--;
        if (num < 1 || end < beg) {
            if (!strcmp(ref, "browser") || !strcmp(ref, "track")) continue;
            fprintf(stderr, "[bed_read] Parse error reading \"%s\"\n", path.c_str());
            gzclose(fp);
            return nullptr;
        }
        raw[ref].emplace_back((int64_t)beg, (int64_t)end);
    }
    gzclose(fp);
    for (auto &kv : raw) {
        auto &v = kv.second;
        std::sort(v.begin(), v.end());
        Ivals iv;
        for (auto &pr : v) {
            if (pr.second <= pr.first) continue;     // empty interval never overlaps anything
            if (!iv.beg.empty() && pr.first <= iv.end.back()) { if (pr.second > iv.end.back()) iv.end.back() = pr.second; }
            else { iv.beg.push_back(pr.first); iv.end.push_back(pr.second); }
        }
        b->m_[kv.first] = std::move(iv);
    }
    return b;
}

bool Bed::overlap(const std::string &chr, int64_t beg, int64_t end) const
{
    const Ivals *iv = get(chr);
    if (!iv) return false;
    size_t i = (size_t)(std::upper_bound(iv->end.begin(), iv->end.end(), beg) - iv->end.begin());
    return i < iv->beg.size() && iv->beg[i] < end;
}

int str2flag(const char *s)
{
    char *end;
    long v = strtol(s, &end, 0);
    if (end != s && *end == 0) return v < 0 ? -1 : (int)v;
    static const struct { const char *n; int f; } names[] = {
        { "PAIRED", 1 }, { "PROPER_PAIR", 2 }, { "UNMAP", 4 }, { "MUNMAP", 8 }, { "REVERSE", 16 }, { "MREVERSE", 32 },
        { "READ1", 64 }, { "READ2", 128 }, { "SECONDARY", 256 }, { "QCFAIL", 512 }, { "DUP", 1024 }, { "SUPPLEMENTARY", 2048 } };
    int flag = 0;
    const char *p = s;
    while (*p) {
        const char *e = p; while (*e && *e != ',') ++e;
        bool hit = false;
        for (auto &nm : names)
            if (strlen(nm.n) == (size_t)(e - p) && strncasecmp(p, nm.n, (size_t)(e - p)) == 0) { flag |= nm.f; hit = true; break; }
        if (!hit) return -1;
        p = *e ? e + 1 : e;
    }
    return flag;
}

bool read_file_list(const std::string &path, std::vector<std::string> *out)
{
    std::ifstream in(path);
    if (!in) return false;
    std::string l;
    while (std::getline(in, l)) {
        while (!l.empty() && isspace((unsigned char)l.back())) l.pop_back();
        if (!l.empty()) out->push_back(l);
    }
    return !out->empty();
}

}  // namespace sta
