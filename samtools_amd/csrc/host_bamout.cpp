// host_bamout.cpp -- see host_bamout.h
#include "host_bamout.h"
#include <climits>
#include <cstdlib>
#include <cstring>
#include <zlib.h>

namespace sta {

namespace {

void put_u16(std::vector<uint8_t> &o, uint32_t v) { o.push_back((uint8_t)v); o.push_back((uint8_t)(v >> 8)); }
void put_u32(std::vector<uint8_t> &o, uint32_t v) { for (int k = 0; k < 4; ++k) o.push_back((uint8_t)(v >> (8 * k))); }
void put_bytes(std::vector<uint8_t> &o, const void *p, size_t n) { const uint8_t *b = (const uint8_t *)p; o.insert(o.end(), b, b + n); }

// SAM spec 5.3 (reg2bin): the smallest bin containing [beg, end)
int reg2bin(int64_t beg, int64_t end)
{
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

// one "TG:T:value" field -> its BAM encoding (sam_parse1's choices: the smallest integer type that holds the value)
bool aux_to_bam(const std::string &f, std::vector<uint8_t> &o)
{
    if (f.size() < 5 || f[2] != ':' || f[4] != ':') return false;
    const char type = f[3];
    const char *v = f.c_str() + 5;
    o.push_back((uint8_t)f[0]); o.push_back((uint8_t)f[1]);
    switch (type) {
    case 'A': o.push_back('A'); o.push_back((uint8_t)(*v ? *v : ' ')); return true;
    case 'i': {
        const long long x = strtoll(v, nullptr, 10);
        if (x < 0) {
            if (x >= INT8_MIN) { o.push_back('c'); o.push_back((uint8_t)(int8_t)x); }
            else if (x >= INT16_MIN) { o.push_back('s'); put_u16(o, (uint32_t)(int16_t)x); }
            else { o.push_back('i'); put_u32(o, (uint32_t)(int32_t)x); }
        } else {
            if (x <= UINT8_MAX) { o.push_back('C'); o.push_back((uint8_t)x); }
            else if (x <= UINT16_MAX) { o.push_back('S'); put_u16(o, (uint32_t)x); }
            else { o.push_back('I'); put_u32(o, (uint32_t)x); }
        }
        return true; }
    case 'f': { o.push_back('f'); const float x = strtof(v, nullptr); put_bytes(o, &x, 4); return true; }
    case 'd': { o.push_back('d'); const double x = strtod(v, nullptr); put_bytes(o, &x, 8); return true; }
    case 'Z': case 'H': o.push_back((uint8_t)type); put_bytes(o, v, strlen(v) + 1); return true;
    case 'B': {
        const char sub = *v;
        if (!sub || !strchr("cCsSiIf", sub)) return false;
        o.push_back('B'); o.push_back((uint8_t)sub);
        const size_t cnt_at = o.size(); put_u32(o, 0);
        uint32_t cnt = 0;
        const char *p = v + 1;
        while (*p) {
            if (*p == ',') { ++p; continue; }
            char *q;
            if (sub == 'f') { const float x = strtof(p, &q); if (q == p) break; put_bytes(o, &x, 4); }
            else {
                const long long x = strtoll(p, &q, 10);
                if (q == p) break;
                if (sub == 'c' || sub == 'C') o.push_back((uint8_t)x);
                else if (sub == 's' || sub == 'S') put_u16(o, (uint32_t)x);
                else put_u32(o, (uint32_t)x);
            }
            p = q; ++cnt;
        }
        for (int k = 0; k < 4; ++k) o[cnt_at + (size_t)k] = (uint8_t)(cnt >> (8 * k));
        return true; }
    }
    return false;
}

}  // namespace

bool BamWriter::put(const void *p, size_t n)
{
    const uint8_t *b = (const uint8_t *)p;
    while (n) {
        const size_t room = 0xff00 - buf_.size(), take = n < room ? n : room;
        buf_.insert(buf_.end(), b, b + take);
        b += take; n -= take;
        if (buf_.size() == 0xff00 && !flush_block()) return false;
    }
    return true;
}

// one BGZF block (SAM spec 4.1): gzip member with the BC extra subfield = total block size - 1
bool BamWriter::flush_block()
{
    if (buf_.empty()) return true;
    comp_.resize(compressBound((uLong)buf_.size()) + 64);
    z_stream zs; memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, level_ == 0 ? 0 : Z_DEFAULT_COMPRESSION, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
    zs.next_in = buf_.data(); zs.avail_in = (uInt)buf_.size();
    zs.next_out = comp_.data(); zs.avail_out = (uInt)comp_.size();
    const int rc = deflate(&zs, Z_FINISH);
    const size_t clen = zs.total_out;
    deflateEnd(&zs);
    if (rc != Z_STREAM_END || clen + 26 > 0x10000) return false;
    uint8_t head[18] = { 0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0, 0 };
    const uint32_t bsize = (uint32_t)(clen + 25);
    head[16] = (uint8_t)bsize; head[17] = (uint8_t)(bsize >> 8);
    uint8_t tail[8];
    const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), buf_.data(), (uInt)buf_.size()), isz = (uint32_t)buf_.size();
    for (int k = 0; k < 4; ++k) { tail[k] = (uint8_t)(crc >> (8 * k)); tail[4 + k] = (uint8_t)(isz >> (8 * k)); }
    const bool ok = fwrite(head, 1, 18, fp_) == 18 && fwrite(comp_.data(), 1, clen, fp_) == clen && fwrite(tail, 1, 8, fp_) == 8;
    buf_.clear();
    return ok;
}

bool BamWriter::header(const Header &h, const std::string &text)
{
    std::vector<uint8_t> o;
    put_bytes(o, "BAM\1", 4);
    put_u32(o, (uint32_t)text.size()); put_bytes(o, text.data(), text.size());
    put_u32(o, (uint32_t)h.nref());
    for (int i = 0; i < h.nref(); ++i) {
        const std::string &n = h.names[(size_t)i];
        put_u32(o, (uint32_t)n.size() + 1); put_bytes(o, n.c_str(), n.size() + 1);
        put_u32(o, (uint32_t)(h.lens[(size_t)i] > INT32_MAX ? INT32_MAX : h.lens[(size_t)i]));
    }
    return put(o.data(), o.size()) && flush_block();
}

bool BamWriter::record(const Header &h, const Rec &r, const uint8_t *seq4, const uint8_t *qual, const std::vector<std::string> &aux)
{
    (void)h;
    if (r.qname.size() > 254 || r.l_qseq < 0) return false;         // l_read_name is one byte and counts the NUL (SAM spec 4.2)
    std::vector<uint8_t> &o = rec_;
    o.clear();
    const size_t n_cig = r.cigar.size();
    const bool long_cigar = n_cig > 0xffff;                       // SAM spec 4.2.2: the real CIGAR moves to CG:B,I
    int64_t rlen = (r.flag & 4) ? 0 : r.rlen;
    const int64_t end = r.pos + (rlen > 0 ? rlen : 1);
    put_u32(o, 0);                                                  // block_size, patched below
    put_u32(o, (uint32_t)r.tid); put_u32(o, (uint32_t)(int32_t)r.pos);
    o.push_back((uint8_t)(r.qname.size() + 1)); o.push_back(r.mapq);
    put_u16(o, (uint32_t)(uint16_t)reg2bin(r.pos, end));           // (an unplaced record, pos -1: hts_reg2bin(-1, 0) = 4680, by the arithmetic shifts)
    put_u16(o, long_cigar ? 2u : (uint32_t)n_cig);
    put_u16(o, r.flag);
    put_u32(o, (uint32_t)r.l_qseq);
    put_u32(o, (uint32_t)r.mtid); put_u32(o, (uint32_t)(int32_t)r.mpos); put_u32(o, (uint32_t)(int32_t)r.isize);
    put_bytes(o, r.qname.c_str(), r.qname.size() + 1);
    if (long_cigar) { put_u32(o, ((uint32_t)r.l_qseq << 4) | 4u); put_u32(o, ((uint32_t)r.rlen << 4) | 3u); }
    else for (uint32_t c : r.cigar) put_u32(o, c);
    const size_t sb = ((size_t)r.l_qseq + 1) / 2;
    const size_t s0 = o.size();
    put_bytes(o, seq4, sb);
    if (r.l_qseq & 1) o[s0 + sb - 1] &= 0xf0;                       // the unused low nibble of an odd-length sequence is zero
    put_bytes(o, qual, (size_t)r.l_qseq);
    for (const std::string &a : aux) {
        // a field that came from a BAM record and still reads the same goes out with the bytes it came in with
        const std::string *raw = nullptr;
        for (const auto &pr : r.aux_bam) if (pr.first == a) { raw = &pr.second; break; }
        if (raw) o.insert(o.end(), raw->begin(), raw->end());
        else if (!aux_to_bam(a, o)) return false;
    }
    if (long_cigar) {
        o.push_back('C'); o.push_back('G'); o.push_back('B'); o.push_back('I');
        put_u32(o, (uint32_t)n_cig);
        for (uint32_t c : r.cigar) put_u32(o, c);
    }
    const uint32_t bs = (uint32_t)(o.size() - 4);
    for (int k = 0; k < 4; ++k) o[(size_t)k] = (uint8_t)(bs >> (8 * k));
    return flush_try(o.size()) && put(o.data(), o.size());
}

bool BamWriter::close()
{
    if (!flush_block()) return false;
    static const uint8_t eof[28] = { 0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    return fwrite(eof, 1, 28, fp_) == 28 && fflush(fp_) == 0;
}

}  // namespace sta
