// host_stage.cpp -- see host_stage.h
#include "host_stage.h"
#include "host_chunk.h"
#include <algorithm>
#include <atomic>
#include <cstring>
#include <thread>
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
    raw_first = -1; raw_verify = 0; raw_pieces.clear(); raw_rec_off.clear(); raw_keep.clear();
    clip.clear(); mate.clear(); tpl = 0;
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
    bool bq_ok = (r.has_bq || r.zq_restore) && (int32_t)r.bq.size() >= r.l_qseq;
    if (bq_ok) a |= r.zq_restore ? STA_AUX_ZQ_RESTORE : STA_AUX_HAS_BQ;
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
    // BQ pool: materialised from the first record that carries BQ:Z on ('@' = "no adjustment" for everything before it)
    if (bq_ok && !any_bq) { bq.assign(b0, 64); any_bq = true; }
    if (any_bq) { bq.resize(b0 + padded, 64); if (bq_ok) memcpy(&bq[b0], r.bq.data(), (size_t)r.l_qseq); }
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
    bool slice_bq = false;
    if (c.has_bq_pool) for (size_t k = a; k < b; ++k) if (c.aux[k] & STA_AUX_HAS_BQ) { slice_bq = true; break; }
    if (slice_bq && !any_bq) { bq.assign(qual.size() - (((size_t)b1 - b0) << 3), 64); any_bq = true; }
    if (any_bq) {
        if (c.has_bq_pool) bq.insert(bq.end(), c.bq.begin() + ((size_t)b0 << 3), c.bq.begin() + ((size_t)b1 << 3));
        else bq.resize(qual.size(), 64);
    }
    names.insert(names.end(), c.names.begin() + nm0, c.names.begin() + nm1);
    cig_off.resize(n0 + m); base_off8.resize(n0 + m); name_off.resize(n0 + m);
    for (size_t k = 0; k < m; ++k) {
        cig_off[n0 + k] = c.cig_off[a + k] - cg0 + cg_base;
        base_off8[n0 + k] = c.base_off8[a + k] - b0 + q_base8;
        name_off[n0 + k] = c.name_off[a + k] - nm0 + nm_base;
    }
}

void StagedFile::add_ranges(const Slice *g, size_t n_g, int64_t origin, const XcolSpec *xs, int threads, size_t min_bytes_for_threads, PoolSizes *hw, int raw_mode)
{
    if (!n_g) return;
    const int nt = xs ? xs->n_tags : 0;
    if (nt > 0 || any_bq) raw_mode = 0;
    for (size_t s = 0; s < n_g && raw_mode; ++s) if (!g[s].c->raw || !g[s].c->raw_ok || g[s].c->rec_off.size() != (size_t)g[s].c->n()) raw_mode = 0;
    // destination offsets of every slice
    struct Dst { size_t rec, cig, b8, nm, xoff, xtext; };
    std::vector<Dst> d(n_g + 1);
    d[0] = Dst{ pos.size(), cigar.size(), qual.size() >> 3, names.size(), xcol_off.size(), xcol_text.size() };
    bool slices_bq = false;
    size_t bytes = 0;
    for (size_t s = 0; s < n_g; ++s) {
        const Chunk &c = *g[s].c; const size_t a = (size_t)g[s].i0, b = (size_t)g[s].i1;
        Dst n = d[s];
        n.rec += b - a; n.cig += c.cig_off[b] - c.cig_off[a]; n.b8 += c.base_off8[b] - c.base_off8[a]; n.nm += c.name_off[b] - c.name_off[a];
        if (nt > 0) {
            n.xoff += (b - a) * (size_t)nt;
            for (size_t i = a; i < b; ++i)
                for (int t = 0; t < nt; ++t) {
                    const size_t e = i * (size_t)c.n_tags + (size_t)t;
                    n.xtext += (t < c.n_tags && c.tag_has[e]) ? c.tag_off[e + 1] - c.tag_off[e] : 1;
                }
        }
        if (c.has_bq_pool && !slices_bq) for (size_t k = a; k < b; ++k) if (c.aux[k] & STA_AUX_HAS_BQ) { slices_bq = true; break; }
        d[s + 1] = n;
        bytes += (b - a) * 48 + ((size_t)(c.base_off8[b] - c.base_off8[a]) << 3) * 3 / 2;
    }
    const Dst &z = d[n_g];
    if (slices_bq) raw_mode = 0;
    if (slices_bq && !any_bq) { bq.assign(qual.size(), 64); any_bq = true; }
    if (raw_mode) {
        // the engine gets the raw records of the slices: one piece per slice, every new read's record offset inside the concatenation
        uint64_t base = 0, total = 0;
        for (size_t s = 0; s < n_g; ++s) total += g[s].c->raw->size();
        if (total > 0xfffffff0ull) raw_mode = 0;
        else {
            raw_first = (int64_t)d[0].rec; raw_verify = raw_mode == 2;
            raw_rec_off.resize(z.rec - d[0].rec);
            for (size_t s = 0; s < n_g; ++s) {
                const Chunk &c = *g[s].c; const size_t a = (size_t)g[s].i0, b = (size_t)g[s].i1;
                if (b <= a) continue;
                const uint32_t p0 = c.rec_off[a], p1 = b < (size_t)c.n() ? c.rec_off[b] : (uint32_t)c.raw->size();
                raw_pieces.push_back(sta_raw_piece{ c.raw->data() + p0, (uint64_t)(p1 - p0) });
                raw_keep.push_back(c.raw);
                uint32_t *dst = &raw_rec_off[d[s].rec - d[0].rec];
                for (size_t k = 0; k < b - a; ++k) dst[k] = (uint32_t)(base + (c.rec_off[a + k] - p0));
                base += p1 - p0;
            }
        }
    }
    const bool copy_pools = raw_mode != 1;
    {
        PoolSizes local; PoolSizes &h = hw ? *hw : local;
        h.rec = std::max(h.rec, z.rec); h.cig = std::max(h.cig, z.cig); h.b8 = std::max(h.b8, z.b8); h.nm = std::max(h.nm, z.nm);
        h.xoff = std::max(h.xoff, z.xoff); h.xtext = std::max(h.xtext, z.xtext);
        auto room = [](auto &v, size_t need, size_t mark) { if (v.capacity() < need) v.reserve(std::max(need, mark) + std::max(need, mark) / 4 + 64); };
        room(pos, z.rec, h.rec); room(isize, z.rec, h.rec); room(flag, z.rec, h.rec); room(mapq, z.rec, h.rec); room(aux, z.rec, h.rec);
        room(l_qseq, z.rec, h.rec); room(mtid, z.rec, h.rec); room(mpos, z.rec, h.rec);
        room(cig_off, z.rec + 1, h.rec + 1); room(base_off8, z.rec, h.rec); room(name_off, z.rec + 1, h.rec + 1);
        room(cigar, z.cig, h.cig); room(qual, z.b8 << 3, h.b8 << 3); room(seq, z.b8 << 2, h.b8 << 2); room(names, z.nm, h.nm);
        if (any_bq) room(bq, z.b8 << 3, h.b8 << 3);
        if (nt > 0) { room(xcol_off, z.xoff + 1, h.xoff + 1); room(xcol_text, z.xtext, h.xtext); }
    }
    pos.resize(z.rec); isize.resize(z.rec); flag.resize(z.rec); mapq.resize(z.rec); aux.resize(z.rec); l_qseq.resize(z.rec);
    mtid.resize(z.rec); mpos.resize(z.rec); cig_off.resize(z.rec); base_off8.resize(z.rec); name_off.resize(z.rec);
    cigar.resize(z.cig); qual.resize(z.b8 << 3); seq.resize(z.b8 << 2); names.resize(z.nm);
    if (any_bq) bq.resize(z.b8 << 3);
    if (nt > 0) { n_xcols = nt; xcol_off.resize(z.xoff); xcol_text.resize(z.xtext); }
    const bool with_bq = any_bq; const char empty = xs ? xs->empty : '*';
    auto copy_slice = [&](size_t s) {
        const Chunk &c = *g[s].c; const size_t a = (size_t)g[s].i0, b = (size_t)g[s].i1, m = b - a;
        const Dst &o = d[s];
        for (size_t k = 0; k < m; ++k) pos[o.rec + k] = (int32_t)(c.pos[a + k] - origin);
        for (size_t k = 0; k < m; ++k) {
            const int64_t is = c.isize[a + k];
            isize[o.rec + k] = (int32_t)(is > INT32_MAX ? INT32_MAX : is < -INT32_MAX ? -INT32_MAX : is);
        }
        memcpy(&flag[o.rec], &c.flag[a], m * sizeof(uint16_t));
        memcpy(&mapq[o.rec], &c.mapq[a], m);
        memcpy(&aux[o.rec], &c.aux[a], m);
        memcpy(&l_qseq[o.rec], &c.l_qseq[a], m * sizeof(int32_t));
        memcpy(&mtid[o.rec], &c.mtid[a], m * sizeof(int32_t));
        memcpy(&mpos[o.rec], &c.mpos[a], m * sizeof(int64_t));
        const uint32_t cg0 = c.cig_off[a], cg1 = c.cig_off[b], b0 = c.base_off8[a], b1 = c.base_off8[b], nm0 = c.name_off[a], nm1 = c.name_off[b];
        if (copy_pools && cg1 > cg0) memcpy(&cigar[o.cig], &c.cigar[cg0], (size_t)(cg1 - cg0) * sizeof(uint32_t));
        if (copy_pools && b1 > b0) {
            memcpy(&qual[o.b8 << 3], &c.qual[(size_t)b0 << 3], (size_t)(b1 - b0) << 3);
            memcpy(&seq[o.b8 << 2], &c.seq[(size_t)b0 << 2], (size_t)(b1 - b0) << 2);
            if (with_bq) {
                if (c.has_bq_pool) memcpy(&bq[o.b8 << 3], &c.bq[(size_t)b0 << 3], (size_t)(b1 - b0) << 3);
                else memset(&bq[o.b8 << 3], 64, (size_t)(b1 - b0) << 3);
            }
        }
        if (copy_pools && nm1 > nm0) memcpy(&names[o.nm], &c.names[nm0], nm1 - nm0);
        for (size_t k = 0; k < m; ++k) {
            cig_off[o.rec + k] = c.cig_off[a + k] - cg0 + (uint32_t)o.cig;
            base_off8[o.rec + k] = c.base_off8[a + k] - b0 + (uint32_t)o.b8;
            name_off[o.rec + k] = c.name_off[a + k] - nm0 + (uint32_t)o.nm;
        }
        if (nt > 0) {
            size_t xo = o.xoff, xt = o.xtext;
            for (size_t i = a; i < b; ++i)
                for (int t = 0; t < nt; ++t) {
                    xcol_off[xo++] = (uint32_t)xt;
                    const size_t e = i * (size_t)c.n_tags + (size_t)t;
                    if (t < c.n_tags && c.tag_has[e]) { const size_t l = c.tag_off[e + 1] - c.tag_off[e]; memcpy(&xcol_text[xt], c.tag_text.data() + c.tag_off[e], l); xt += l; }
                    else xcol_text[xt++] = empty;
                }
        }
    };
    int nth = threads < 1 ? 1 : threads;
    if ((size_t)nth > n_g) nth = (int)n_g;
    if (bytes < min_bytes_for_threads) nth = 1;                  // small windows: not worth waking threads
    if (nth <= 1) { for (size_t s = 0; s < n_g; ++s) copy_slice(s); return; }
    std::atomic<size_t> next{ 0 };
    auto work = [&] { for (size_t s; (s = next.fetch_add(1)) < n_g;) copy_slice(s); };
    std::vector<std::thread> th;
    for (int t = 1; t < nth; ++t) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
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
    if (tpl == 1 && clip.size() == pos.size()) v.olap_clip = clip.data();
    if (tpl == 2 && mate.size() == pos.size()) v.olap_mate = mate.data();
    v.raw_first = v.n_reads;
    if (raw_first >= 0 && !raw_pieces.empty() && !any_bq) {
        v.raw_first = raw_first; v.n_raw_pieces = (int32_t)raw_pieces.size(); v.raw_pieces = raw_pieces.data(); v.raw_rec_off = raw_rec_off.data();
        v.raw_verify = raw_verify;
    }
    return v;
}

}  // namespace sta
