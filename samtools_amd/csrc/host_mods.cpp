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
#include <string>
#include <vector>

namespace sta {

namespace {
int code16(int ch)      // seq_nt16_table for the canonical-base letters
{
    switch (ch) { case 'A': return 1; case 'C': return 2; case 'G': return 4; case 'T': case 'U': return 8; default: return 15; }
}
int comp16(int c) { return ((c & 1) << 3) | ((c & 2) << 1) | ((c & 4) >> 1) | ((c & 8) >> 3); }
}  // namespace

// the MM / ML evaluation itself, on plain fields: seq = 4-bit packed bases, mm = the MM:Z string, ml = the ML:B:C values (n_ml of them,
// has_ml = the tag is there).  Fills `hits` sorted by query position (MM order within a position).  false: malformed MM (HTSlib reports a
// parse error; the callers show no modifications).
bool parse_base_mods(const uint8_t *seq, int L, bool rev, const char *mm, const uint8_t *ml, size_t n_ml, bool has_ml, std::vector<ModHit> &hits)
{
    hits.clear();
    if (!mm || !*mm || L <= 0) return true;
    size_t ml_i = 0;
    const char *p = mm;
    while (*p) {
        int base = *p++;
        if (base >= 'a' && base <= 'z') base -= 32;
        if (*p != '+' && *p != '-') { hits.clear(); return false; }
        const int strand = *p++ == '-';
        const int want = base == 'N' ? 15 : code16(base);
        int codes[64], n_codes = 0;
        if (*p >= '0' && *p <= '9') { char *q; codes[n_codes++] = -(int)strtol(p, &q, 10); p = q; }
        else while (*p >= 'a' && *p <= 'z' && n_codes < 64) codes[n_codes++] = *p++;
        if (!n_codes) { hits.clear(); return false; }
        if (*p == '?' || *p == '.') ++p;
        int cand = rev ? L : -1;
        while (*p == ',') {
            char *q; long d = strtol(p + 1, &q, 10); p = q;
            int need = (int)d + 1, at = -1;
            while (need > 0) {
                cand += rev ? -1 : 1;
                if (cand < 0 || cand >= L) { cand = rev ? -1 : L; break; }
                int c = (seq[(size_t)cand >> 1] >> ((~cand & 1) << 2)) & 0xf;
                if (rev) c = comp16(c);
                if (want == 15 || c == want) --need;
            }
            if (need == 0) at = cand;
            for (int c = 0; c < n_codes; ++c) {
                const int qual = has_ml && ml_i < n_ml ? (int)ml[ml_i] : -1;
                ++ml_i;
                if (at >= 0) hits.push_back(ModHit{ (uint32_t)at, (uint32_t)hits.size(), codes[c], strand, qual, base });
            }
        }
        if (*p == ';') ++p; else if (*p) { hits.clear(); return false; }
    }
    std::stable_sort(hits.begin(), hits.end(), [](const ModHit &a, const ModHit &b) { return a.qpos < b.qpos; });
    return true;
}

// "[" + one "<strand><code><probability>" per modification + "]": what pileup_seq (bam_plcmd.c:86-109) and bam_plp_insertion_mod append
// behind a modified base; at most 256 modifications per base are shown (the size of the array bam_mods_at_qpos is handed there)
size_t append_mod_text(const ModHit *h, size_t n, std::string &out)
{
    const size_t before = out.size();
    out.push_back('[');
    for (size_t j = 0; j < n && j < 256; ++j) {
        char buf[48]; int k;
        if (h[j].code < 0) k = snprintf(buf, sizeof buf, "%c(%d)", "+-"[h[j].strand], -h[j].code);
        else k = snprintf(buf, sizeof buf, "%c%c", "+-"[h[j].strand], h[j].code);
        if (h[j].qual >= 0) k += snprintf(buf + k, sizeof buf - (size_t)k, "%d", h[j].qual);
        out.append(buf, (size_t)k);
    }
    out.push_back(']');
    return out.size() - before;
}

void format_base_mods(const Rec &r, pvector<uint32_t> &qpos, pvector<uint32_t> &toff, pvector<char> &text)
{
    if (r.mm.empty() || r.l_qseq <= 0) return;
    std::vector<ModHit> hits;
    if (!parse_base_mods(r.seq.data(), r.l_qseq, (r.flag & 16) != 0, r.mm.c_str(), r.ml.data(), r.ml.size(), r.has_ml, hits)) return;
    std::string t;
    for (size_t i = 0; i < hits.size();) {
        size_t j = i;
        while (j < hits.size() && hits[j].qpos == hits[i].qpos) ++j;
        qpos.push_back(hits[i].qpos);
        toff.push_back((uint32_t)text.size());
        t.clear();
        append_mod_text(&hits[i], j - i, t);
        text.insert(text.end(), t.begin(), t.end());
        i = j;
    }
}

}  // namespace sta
