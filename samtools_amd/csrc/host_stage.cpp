// host_stage.cpp -- see host_stage.h
#include "host_stage.h"
#include "host_chunk.h"
#include <cstring>
#include <climits>

namespace sta {

void StagedFile::clear()
{
    pos.clear(); l_qseq.clear(); mtid.clear(); isize.clear(); flag.clear(); mapq.clear(); aux.clear();
    cig_off.clear(); base_off8.clear(); name_off.clear(); cigar.clear(); mpos.clear();
    seq.clear(); qual.clear(); bq.clear(); names.clear();
    xcol_off.clear(); xcol_text.clear(); n_xcols = 0;
    mod_off.clear(); mod_qpos.clear(); mod_toff.clear(); mod_text.clear(); with_mods = false;
    any_bq = false;
}

void StagedFile::add(const Rec &r, int64_t origin, const std::set<std::string> *rg_excl, const XcolSpec *xs)
{
    if (xs && xs->n_cols() > 0) {
        n_xcols = xs->n_cols();
        if (xs->rnext) {
            xcol_off.push_back((uint32_t)xcol_text.size());
            if (r.mtid >= 0 && xs->hdr && r.mtid < xs->hdr->nref()) { const std::string &nm = xs->hdr->names[(size_t)r.mtid]; xcol_text.insert(xcol_text.end(), nm.begin(), nm.end()); }
            else xcol_text.push_back('*');
        }
        for (int t = 0; t < xs->n_tags; ++t) {
            xcol_off.push_back((uint32_t)xcol_text.size());
            if ((size_t)t < r.tag_has.size() && r.tag_has[(size_t)t]) xcol_text.insert(xcol_text.end(), r.tagtext[(size_t)t].begin(), r.tagtext[(size_t)t].end());
            else xcol_text.push_back(xs->empty);
        }
    }
    if (xs && xs->mods) {
        with_mods = true;
        mod_off.push_back((uint32_t)mod_qpos.size());
        format_base_mods(r, mod_qpos, mod_toff, mod_text);
    }
    pos.push_back((int32_t)(r.pos - origin));
    flag.push_back(r.flag);
    mapq.push_back(r.mapq);
    uint8_t a = 0;
    bool bq_ok = r.has_bq && (int32_t)r.bq.size() >= r.l_qseq;
    if (bq_ok) a |= STA_AUX_HAS_BQ;
    if (r.has_zq) a |= STA_AUX_HAS_ZQ;
    if (rg_excl && !r.rg.empty() && rg_excl->count(r.rg)) a |= STA_AUX_SKIP;
    if (r.accepted) a |= STA_AUX_ACCEPTED;
    aux.push_back(a);
    l_qseq.push_back(r.l_qseq);
    cig_off.push_back((uint32_t)cigar.size());
    cigar.insert(cigar.end(), r.cigar.begin(), r.cigar.end());
    // bases: padded to a multiple of 8 so that seq offset = qual offset / 2 stays whole
    size_t b0 = qual.size();
    base_off8.push_back((uint32_t)(b0 >> 3));
    size_t padded = ((size_t)r.l_qseq + 7) & ~(size_t)7;
    qual.resize(b0 + padded, 0);
    if (r.l_qseq) memcpy(&qual[b0], r.qual.data(), (size_t)r.l_qseq);
    seq.resize((b0 + padded) / 2, 0);
    if (r.l_qseq) memcpy(&seq[b0 / 2], r.seq.data(), ((size_t)r.l_qseq + 1) / 2);
    bq.resize(b0 + padded, 64);               // '@' = "no adjustment"
    if (bq_ok) { memcpy(&bq[b0], r.bq.data(), (size_t)r.l_qseq); any_bq = true; }
    mtid.push_back(r.mtid);
    mpos.push_back(r.mpos);
    int64_t is = r.isize;
    if (is > INT32_MAX) is = INT32_MAX;
    if (is < -INT32_MAX) is = -INT32_MAX;
    isize.push_back((int32_t)is);
    name_off.push_back((uint32_t)names.size());
    names.insert(names.end(), r.qname.begin(), r.qname.end());
    names.push_back('\0');
}

void StagedFile::add_range(const Chunk &c, int64_t i0, int64_t i1, int64_t origin, const XcolSpec *xs)
{
    if (i1 <= i0) return;
    if (xs && xs->n_tags > 0) {
        n_xcols = xs->n_tags;
        for (int64_t i = i0; i < i1; ++i)
            for (int t = 0; t < xs->n_tags; ++t) {
                xcol_off.push_back((uint32_t)xcol_text.size());
                const size_t e = (size_t)i * (size_t)c.n_tags + (size_t)t;
                if (t < c.n_tags && c.tag_has[e]) xcol_text.insert(xcol_text.end(), c.tag_text.data() + c.tag_off[e], c.tag_text.data() + c.tag_off[e + 1]);
                else xcol_text.push_back(xs->empty);
            }
    }
    const size_t a = (size_t)i0, b = (size_t)i1, m = b - a, n0 = pos.size();
    pos.resize(n0 + m); isize.resize(n0 + m);
    for (size_t k = 0; k < m; ++k) pos[n0 + k] = (int32_t)(c.pos[a + k] - origin);
    for (size_t k = 0; k < m; ++k) {
        int64_t is = c.isize[a + k];
        isize[n0 + k] = (int32_t)(is > INT32_MAX ? INT32_MAX : is < -INT32_MAX ? -INT32_MAX : is);
    }
    flag.insert(flag.end(), c.flag.begin() + i0, c.flag.begin() + i1);
    mapq.insert(mapq.end(), c.mapq.begin() + i0, c.mapq.begin() + i1);
    aux.insert(aux.end(), c.aux.begin() + i0, c.aux.begin() + i1);
    l_qseq.insert(l_qseq.end(), c.l_qseq.begin() + i0, c.l_qseq.begin() + i1);
    mtid.insert(mtid.end(), c.mtid.begin() + i0, c.mtid.begin() + i1);
    mpos.insert(mpos.end(), c.mpos.begin() + i0, c.mpos.begin() + i1);
    // pools: the slice [first offset of i0, first offset of i1) of every pool, then offsets shifted by (new base - old base)
    const uint32_t cg0 = c.cig_off[a], cg1 = c.cig_off[b], b0 = c.base_off8[a], b1 = c.base_off8[b], nm0 = c.name_off[a], nm1 = c.name_off[b];
    const uint32_t cg_base = (uint32_t)cigar.size(), q_base8 = (uint32_t)(qual.size() >> 3), nm_base = (uint32_t)names.size();
    cigar.insert(cigar.end(), c.cigar.begin() + cg0, c.cigar.begin() + cg1);
    qual.insert(qual.end(), c.qual.begin() + ((size_t)b0 << 3), c.qual.begin() + ((size_t)b1 << 3));
    seq.insert(seq.end(), c.seq.begin() + ((size_t)b0 << 2), c.seq.begin() + ((size_t)b1 << 2));
    if (c.has_bq_pool) {
        bq.insert(bq.end(), c.bq.begin() + ((size_t)b0 << 3), c.bq.begin() + ((size_t)b1 << 3));
        if (!any_bq) for (size_t k = a; k < b; ++k) if (c.aux[k] & STA_AUX_HAS_BQ) { any_bq = true; break; }
    } else bq.resize(qual.size(), 64);
    names.insert(names.end(), c.names.begin() + nm0, c.names.begin() + nm1);
    cig_off.resize(n0 + m); base_off8.resize(n0 + m); name_off.resize(n0 + m);
    for (size_t k = 0; k < m; ++k) {
        cig_off[n0 + k] = c.cig_off[a + k] - cg0 + cg_base;
        base_off8[n0 + k] = c.base_off8[a + k] - b0 + q_base8;
        name_off[n0 + k] = c.name_off[a + k] - nm0 + nm_base;
    }
}

void StagedFile::finish()
{
    cig_off.push_back((uint32_t)cigar.size());
    name_off.push_back((uint32_t)names.size());
    if (n_xcols) xcol_off.push_back((uint32_t)xcol_text.size());
    if (with_mods) { mod_off.push_back((uint32_t)mod_qpos.size()); mod_toff.push_back((uint32_t)mod_text.size()); }
}

sta_reads StagedFile::view() const
{
    sta_reads v;
    memset(&v, 0, sizeof v);
    v.n_reads = n();
    v.pos = pos.data(); v.flag = flag.data(); v.mapq = mapq.data(); v.aux = aux.data(); v.l_qseq = l_qseq.data();
    v.cig_off = cig_off.data(); v.base_off8 = base_off8.data(); v.mtid = mtid.data(); v.mpos = mpos.data();
    v.isize = isize.data(); v.name_off = name_off.data(); v.cigar = cigar.data(); v.seq = seq.data(); v.qual = qual.data();
    v.bq = any_bq ? bq.data() : nullptr; v.names = names.data();
    v.n_cigar_total = cigar.size(); v.n_bases_total = qual.size(); v.n_name_bytes = names.size();
    if (n_xcols) { v.n_xcols = n_xcols; v.xcol_off = xcol_off.data(); v.xcol_text = xcol_text.data(); v.n_xcol_bytes = xcol_text.size(); }
    if (with_mods && mod_off.size() == pos.size() + 1) {
        v.mod_off = mod_off.data(); v.mod_qpos = mod_qpos.data(); v.mod_toff = mod_toff.data(); v.mod_text = mod_text.data();
        v.n_mod_entries = mod_qpos.size(); v.n_mod_bytes = mod_text.size();
    }
    return v;
}

}  // namespace sta
