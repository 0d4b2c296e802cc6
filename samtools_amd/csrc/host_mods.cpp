// host_mods.cpp -- base modifications for `mpileup --output-mods` on the host side.
//
// The reference parses a read's MM / ML tags when the read enters the pileup (bam_plcmd.c:356-362: hts_base_mod_state_alloc +
// bam_parse_basemod through the iterator's constructor hook) and asks bam_mods_at_qpos for every base it prints (:86-109).
// Both live in HTSlib (sam_mods.c, absent here).  The engine needs the answer per (read, query position) on the device, so the
// host evaluates the tags once per read while staging and hands over, for every modified base, the exact text pileup_seq
// appends: "[" + one "<strand><code><probability>" per modification in MM order + "]".
//   MM:Z:  ([ACGTUN][-+]([a-z]+|[0-9]+)[.?]?(,[0-9]+)*;)*     ML:B:C one value per (position, code), in MM order.
// A delta skips that many bases of the entry's canonical kind (any base for N) before the next modified one, counted along the
// ORIGINAL read: for a reverse-strand record from the end of SEQ, on complemented bases (SAM tags specification 1.7).
#include "host_stage.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace sta {

namespace {
int code16(int ch)      // seq_nt16_table for the canonical-base letters
{
    switch (ch) { case 'A': return 1; case 'C': return 2; case 'G': return 4; case 'T': case 'U': return 8; default: return 15; }
}
int comp16(int c) { return ((c & 1) << 3) | ((c & 2) << 1) | ((c & 4) >> 1) | ((c & 8) >> 3); }
struct Hit { uint32_t qpos; uint32_t order; int code, strand, qual; };
}  // namespace

void format_base_mods(const Rec &r, pvector<uint32_t> &qpos, pvector<uint32_t> &toff, pvector<char> &text)
{
    if (r.mm.empty() || r.l_qseq <= 0) return;
    const int L = r.l_qseq;
    const bool rev = (r.flag & 16) != 0;
    std::vector<Hit> hits;
    size_t ml_i = 0;
    const char *p = r.mm.c_str();
    while (*p) {
        int base = *p++;
        if (base >= 'a' && base <= 'z') base -= 32;
        if (*p != '+' && *p != '-') return;              // malformed: no modifications (HTSlib reports a parse error)
        const int strand = *p++ == '-';
        const int want = base == 'N' ? 15 : code16(base);
        int codes[64], n_codes = 0;
        if (*p >= '0' && *p <= '9') { char *q; codes[n_codes++] = -(int)strtol(p, &q, 10); p = q; }
        else while (*p >= 'a' && *p <= 'z' && n_codes < 64) codes[n_codes++] = *p++;
        if (!n_codes) return;
        if (*p == '?' || *p == '.') ++p;
        int cand = rev ? L : -1;
        while (*p == ',') {
            char *q; long d = strtol(p + 1, &q, 10); p = q;
            int need = (int)d + 1, at = -1;
            while (need > 0) {
                cand += rev ? -1 : 1;
                if (cand < 0 || cand >= L) { cand = rev ? -1 : L; break; }
                int c = (r.seq[(size_t)cand >> 1] >> ((~cand & 1) << 2)) & 0xf;
                if (rev) c = comp16(c);
                if (want == 15 || c == want) --need;
            }
            if (need == 0) at = cand;
            for (int c = 0; c < n_codes; ++c) {
                const int qual = r.has_ml && ml_i < r.ml.size() ? (int)r.ml[ml_i] : -1;
                ++ml_i;
                if (at >= 0) hits.push_back(Hit{ (uint32_t)at, (uint32_t)hits.size(), codes[c], strand, qual });
            }
        }
        if (*p == ';') ++p; else if (*p) return;
    }
    std::stable_sort(hits.begin(), hits.end(), [](const Hit &a, const Hit &b) { return a.qpos < b.qpos; });
    for (size_t i = 0; i < hits.size();) {
        size_t j = i;
        qpos.push_back(hits[i].qpos);
        toff.push_back((uint32_t)text.size());
        text.push_back('[');
        for (; j < hits.size() && hits[j].qpos == hits[i].qpos; ++j) {
            if (j - i >= 256) continue;                  // bam_mods_at_qpos is asked for at most 256 modifications per base
            char buf[48]; int n;
            if (hits[j].code < 0) n = snprintf(buf, sizeof buf, "%c(%d)", "+-"[hits[j].strand], -hits[j].code);
            else n = snprintf(buf, sizeof buf, "%c%c", "+-"[hits[j].strand], hits[j].code);
            if (hits[j].qual >= 0) n += snprintf(buf + n, sizeof buf - (size_t)n, "%d", hits[j].qual);
            text.insert(text.end(), buf, buf + n);
        }
        text.push_back(']');
        i = j;
    }
}

}  // namespace sta
