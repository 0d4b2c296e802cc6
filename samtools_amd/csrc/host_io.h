// host_io.h -- host-side input decoding for the samtools-amd drivers (C++).
// Stands where HTSlib's sam_open/sam_hdr_read/sam_read1/sam_itr_next, faidx and samtools'
// bedidx.c stand for the reference drivers (bam_plcmd.c:500-569, bam2depth.c:926-976).
// Records are decoded straight into the fields the staging layer needs.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "host_pinned.h"
#include <utility>
#include <unordered_map>
#include <memory>
#include <functional>

namespace sta {

struct Header {
    std::vector<std::string> names;
    std::vector<int64_t> lens;
    std::string text;
    std::unordered_map<std::string, int> index;
    int tid(const std::string &n) const { auto it = index.find(n); return it == index.end() ? -1 : it->second; }
    int nref() const { return (int)names.size(); }
};

struct Rec {
    int32_t tid = -1, mtid = -1;
    int64_t pos = 0, mpos = 0, isize = 0;
    uint16_t flag = 0;
    uint8_t mapq = 0;
    int32_t l_qseq = 0;
    std::string qname;
    std::vector<uint32_t> cigar;
    std::vector<uint8_t> seq;    // 4-bit packed
    std::vector<uint8_t> qual;
    std::vector<uint8_t> bq;     // BQ:Z bytes (empty if absent)
    std::vector<uint8_t> zq;     // ZQ:Z bytes, kept only with set_keep_aux (calmd -r without -A turns them back into BQ:Z)
    bool zq_restore = false;     // staging: `bq` holds the ZQ:Z string to be added back to the qualities (STA_AUX_ZQ_RESTORE)
    std::vector<std::string> auxv;   // with set_keep_aux: every aux field as sam_format1 prints it ("NM:i:3"), in record order
    // with set_keep_aux, BAM input only: (text, raw BAM bytes) of every aux field as it was read.  The BAM writer copies the raw
    // bytes of a field whose text it is handed unchanged (sam_write1 on the same bam1_t keeps untouched aux bytes: floats at full
    // precision, integer widths as they were); only fields a command rewrote are encoded from text.  Empty for SAM input.
    std::vector<std::pair<std::string, std::string>> aux_bam;
    bool has_bq = false, has_zq = false;
    std::string rg;              // RG:Z value ("" if absent)
    std::string mm;              // MM:Z / Mm:Z base-modification list ("" if absent) and its ML / Ml probabilities (--output-mods)
    std::vector<uint8_t> ml; bool has_ml = false;
    // --output-extra aux tags: text of every wanted tag as mpileup prints it (bam_plcmd.c:811-850); tag_has[i] = 0 if absent
    std::vector<std::string> tagtext;
    std::vector<char> tag_has;
    int64_t rlen = 0;            // reference span (bam_cigar2rlen)
    bool cigar_from_tag = false; // BAM: the CIGAR came out of a CG:B,I tag (the record's own CIGAR field is the <l_seq>S<span>N placeholder)
    bool accepted = false;       // carried from an earlier window whose -d replay kept it (STA_AUX_ACCEPTED when re-staged)
    // template state kept by the input lanes in file order (host_names.h): the lane's running number of the record; depth -s: the column
    // below which it is not counted; mpileup: the record whose overlap-hash entry it found (-1: none) and, for both partners of such a
    // pair, the other one's end (a record stays staged while its partner can still touch a column)
    int64_t id = -1, clip = 0, mate_id = -1, mate_end = INT64_MIN;
    uint64_t name_h = 0;         // qname_hash64 of qname (chunk lane: computed by the decode threads)
    int64_t end() const { return pos + rlen; }
    int64_t endpos() const { int64_t l = (flag & 4) ? 0 : rlen; return pos + (l > 0 ? l : 1); }   // bam_endpos
};

class AlnReader {
public:
    // threads: BGZF inflate workers (<= 0: $STA_IO_THREADS, else 4..8 depending on the machine); records are parsed one batch ahead on a further thread
    static std::unique_ptr<AlnReader> open(const std::string &path, std::string *err, int threads = 0);
    ~AlnReader();
    const Header &header() const { return hdr_; }
    void set_region(int tid, int64_t beg, int64_t end) { has_reg_ = true; rtid_ = tid; rbeg_ = beg; rend_ = end; }
    // two-character aux tags whose values next() should format into Rec::tagtext (in this order)
    void set_wanted_tags(const std::vector<std::string> &tags);
    // keep every aux field of a record as SAM text in Rec::auxv (for the commands that write records: calmd)
    void set_keep_aux(bool on);
    // called for every record next() returns (read-level statistics of `coverage`: coverage.c:182-196 counts in its callback)
    std::function<void(const Rec &)> on_record;
    // 1 = record, 0 = EOF, <0 = error
    int next(Rec &r);
    // chunked access (host_chunk.h): raw_group() cuts the byte stream into groups of whole records on one thread (1 = group,
    // 0 = end of data, <0 = error), parse_raw() decodes one record of a group and may run on any number of threads at once.
    int raw_group(pvector<uint8_t> &out, size_t target, int64_t *n_records);      // (a page-locked vector: a BAM group may be uploaded as it is, host_chunk.h)
    int parse_raw(const uint8_t *p, size_t avail, size_t *used, Rec &r, std::string &scratch) const;
    bool is_bam() const;
    bool record_stream_position(std::string *path, uint64_t *coffset, uint64_t *consumed) const;
    void release_source();
    // Start reading at a BAI virtual offset (coffset << 16 | offset inside the inflated block) instead of behind the header:
    // what sam_itr_querys does with the index for a region (bam_plcmd.c:550, bam2depth.c:961-975).  Only before the first
    // record has been asked for, BGZF BAM only; false (and nothing changes) otherwise.
    bool seek_voffset(uint64_t voffset);
    // a record beyond the region in a position-sorted file: nothing further can match (next() / the chunk lane stop there)
    // the first record beyond the region ends the reading -- only where the input is known to be in coordinate order: the header says so
    // (@HD SO:coordinate) or the reader was positioned through an index (which exists for sorted files only); anything else is
    // filtered to its last record, as HTSlib's unindexed region filter would
    bool past_region(const Rec &r) const { return has_reg_ && sorted_hint_ && r.tid >= 0 && (r.tid > rtid_ || (r.tid == rtid_ && r.pos >= rend_)); }
    bool has_region() const { return has_reg_; }
    bool in_region(const Rec &r) const { return !has_reg_ || !(r.tid != rtid_ || r.pos >= rend_ || r.endpos() <= rbeg_); }
    struct Impl;
private:
    AlnReader() = default;
    Impl *p_ = nullptr;
    Header hdr_;
    bool has_reg_ = false; int rtid_ = 0; int64_t rbeg_ = 0, rend_ = 0;
    bool sorted_hint_ = false;
    int next_raw(Rec &r);
    void parse_ahead();
};

// hts_parse_reg-like ("chr", "chr:beg", "chr:beg-end", thousands commas); 0-based half open
bool parse_region(const Header &h, const std::string &reg, int *tid, int64_t *beg, int64_t *end);

// The linear index of a BAI file (SAM spec 5.2): per reference, the smallest virtual offset of any alignment overlapping each
// 16 kbp window.  Stands where hts_idx_load + the iterator's start offset stand for region runs; bins are skipped (a sorted
// scan from the linear offset reads at most one window of extra records).
class BaiIndex {
public:
    // <bam>.bai, else the .bam suffix replaced by .bai; nullptr when there is none or it is malformed
    static std::unique_ptr<BaiIndex> load_for(const std::string &bam_path);
    // an index file named by the user (-X / --customized-index: bam_plcmd.c:1243-1262, bam2depth.c:873-911); nullptr if unreadable or not a BAI
    static std::unique_ptr<BaiIndex> load_file(const std::string &index_path, const std::string &bam_path);
    // virtual offset to start reading from so that every alignment overlapping [pos, ...) of tid -- and everything after -- is
    // seen; 0 = unknown (read from the start); UINT64_MAX = nothing at or beyond (tid, pos)
    uint64_t start_offset(int tid, int64_t pos) const;
    // the index file was last written before the data file: HTSlib warns about such a pair and uses it all the same, and so do the drivers
    // (copied or checked-out data often carries an index whose time stamp is not the later one)
    bool older_than_data() const { return stale_; }
private:
    std::vector<std::vector<uint64_t>> lin_;
    bool stale_ = false;
};

// Reference FASTA (faidx stand-in: fai_load + faidx_fetch_seq64).  With a `.fai` beside an uncompressed file the index is all
// load() reads; a contig's bases are read (offset, line geometry of its index line) when they are first asked for, and the contig
// behind it in the file is fetched on a thread of its own meanwhile -- a genome-sized reference used to cost seconds of parsing
// before the first window.  Without an index, or for a compressed file, or when the file does not fit its index, the whole file is
// parsed as before.  fetch() may be called from several threads; the pointers it returns stay valid for the object's lifetime.
class Fasta {
public:
    static std::unique_ptr<Fasta> load(const std::string &path);
    ~Fasta();
    const std::string *fetch(const std::string &name) const;
    // contig names in file (index) order
    const std::vector<std::string> &names() const { return names_; }
    bool lazy() const { return lazy_; }
private:
    Fasta() = default;
    struct Lazy;
    static std::unique_ptr<Fasta> load_whole(const std::string &path);
    std::vector<std::string> names_;
    mutable std::vector<std::string> seqs_;
    std::unordered_map<std::string, size_t> idx_;
    bool lazy_ = false;
    Lazy *lz_ = nullptr;
};

// BED / position list (bedidx.c:258-364), kept as merged disjoint sorted intervals per contig:
// the overlap predicate of bed_overlap() (bedidx.c:159-197) is preserved by merging.
class Bed {
public:
    static std::unique_ptr<Bed> load(const std::string &path);
    struct Ivals { std::vector<int64_t> beg, end; };
    const Ivals *get(const std::string &chr) const { auto it = m_.find(chr); return it == m_.end() ? nullptr : &it->second; }
    bool overlap(const std::string &chr, int64_t beg, int64_t end) const;
private:
    std::unordered_map<std::string, Ivals> m_;
};

// One record as a SAM text line the way sam_format1 writes it (HTSlib sam.c; SAM spec 1.4): seq4 / qual are the record's bases
// (4-bit packed from an even offset, one quality byte each -- the record's own or a changed copy), aux its fields as text
void format_sam_record(const Header &h, const Rec &r, const uint8_t *seq4, const uint8_t *qual, const std::vector<std::string> &aux, std::string &out);

int str2flag(const char *s);   // bam_str2flag
bool read_file_list(const std::string &path, std::vector<std::string> *out);   // bam_plcmd.c:944-999

}  // namespace sta
