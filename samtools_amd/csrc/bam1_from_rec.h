// bam1_from_rec.h -- builds an HTSlib-layout bam1_t from a decoded record (host_io.h Rec): what sam_read1 would have
// produced.  Used by the small clients of the bam_plp_* surface (plpdump, bedcov, coverage).
#pragma once
#include "../../include/samtools_amd_plp.h"
#include "host_io.h"
#include <cstdlib>
#include <cstring>

namespace sta {

inline void rec_to_bam1(const Rec &r, bam1_t *b)
{
    size_t lqn = r.qname.size() + 1, pad = (4 - (lqn & 3)) & 3;
    size_t need = lqn + pad + r.cigar.size() * 4 + ((size_t)r.l_qseq + 1) / 2 + (size_t)r.l_qseq;
    if (b->m_data < need) { b->data = (uint8_t *)realloc(b->data, need); b->m_data = (uint32_t)need; }
    b->l_data = (int)need;
    b->core.pos = r.pos; b->core.tid = r.tid; b->core.bin = 0; b->core.qual = r.mapq; b->core.l_extranul = (uint8_t)pad;
    b->core.flag = r.flag; b->core.l_qname = (uint16_t)(lqn + pad); b->core.n_cigar = (uint32_t)r.cigar.size();
    b->core.l_qseq = r.l_qseq; b->core.mtid = r.mtid; b->core.mpos = r.mpos; b->core.isize = r.isize;
    uint8_t *p = b->data;
    memcpy(p, r.qname.c_str(), lqn); p += lqn;
    memset(p, 0, pad); p += pad;
    if (!r.cigar.empty()) memcpy(p, r.cigar.data(), r.cigar.size() * 4);
    p += r.cigar.size() * 4;
    if (r.l_qseq) memcpy(p, r.seq.data(), ((size_t)r.l_qseq + 1) / 2);
    p += ((size_t)r.l_qseq + 1) / 2;
    if (r.l_qseq) memcpy(p, r.qual.data(), (size_t)r.l_qseq);
}

}  // namespace sta
