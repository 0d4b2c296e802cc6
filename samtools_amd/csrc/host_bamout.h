// host_bamout.h -- BAM output for the commands that write records (calmd -b / -u).
// Stands where HTSlib's bam_write1 + bgzf.c stand behind sam_write1 (bam_md.c:486-489); the format is SAM spec sections 4.1 (BGZF)
// and 4.2 (BAM).  Records never straddle a BGZF block unless they are larger than one (bgzf_flush_try), the header ends its own block.
#pragma once
#include "host_io.h"
#include <cstdio>
#include <memory>

namespace sta {

class BamWriter {
public:
    // level: 0 = stored deflate blocks (calmd -u), otherwise zlib's default level (-b)
    BamWriter(FILE *fp, int level) : fp_(fp), level_(level) {}
    bool header(const Header &h, const std::string &text);
    // seq4 / qual: the record's bases (4-bit packed from an even offset, one quality byte each); aux: its fields as SAM text
    bool record(const Header &h, const Rec &r, const uint8_t *seq4, const uint8_t *qual, const std::vector<std::string> &aux);
    bool close();                        // flushes the open block and appends the end-of-file marker block
private:
    bool put(const void *p, size_t n);   // bgzf_write
    bool flush_block();
    bool flush_try(size_t need) { return buf_.size() + need > 0xff00 ? flush_block() : true; }
    FILE *fp_; int level_;
    std::vector<uint8_t> buf_, rec_, comp_;
};

}  // namespace sta
