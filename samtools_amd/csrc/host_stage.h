// host_stage.h -- builds the structure-of-arrays staging buffers (sta_reads) from decoded records.
// This is the "C host code stages pre-decoded BAM records" half of the boundary (BASELINE.json
// north_star); the arrays are what sta_stage_window() copies to HBM.
#pragma once
#include <string>
#include <vector>
#include "host_io.h"
#include "host_pinned.h"
#include "../../include/samtools_amd.h"
#include <set>
#include <memory>

namespace sta {

// what the host formats per read for --output-extra (RNEXT first, then the aux tags)
struct XcolSpec {
    bool rnext = false;
    const Header *hdr = nullptr;     // contig names for RNEXT
    int n_tags = 0;
    char empty = '*';                // --output-empty: printed for a read without the tag
    bool mods = false;               // --output-mods: stage the bracketed modification text of every modified base
    int n_cols() const { return (rnext ? 1 : 0) + n_tags; }
};

struct Chunk;       // host_chunk.h
// appends the modified bases of one record (host_mods.cpp): query position, offset of its "[+m128]" text, the text
void format_base_mods(const Rec &r, pvector<uint32_t> &qpos, pvector<uint32_t> &toff, pvector<char> &text);
// the evaluation of a read's MM / ML tags on plain fields (host_mods.cpp), shared with the drop-in surface (plp_api.cpp: bam_parse_basemod)
struct ModHit { uint32_t qpos; uint32_t order; int code, strand, qual, canonical; };      // code < 0: ChEBI number (negated); canonical: the MM entry's base letter
bool parse_base_mods(const uint8_t *seq, int l_qseq, bool rev, const char *mm, const uint8_t *ml, size_t n_ml, bool has_ml, std::vector<ModHit> &hits);
size_t append_mod_text(const ModHit *h, size_t n, std::string &out);

struct StagedFile {
    // page-locked (host_pinned.h): sta_stage_window copies straight out of these
    pvector<int32_t> pos, l_qseq, mtid, isize;
    pvector<uint16_t> flag;
    pvector<uint8_t> mapq, aux;
    pvector<uint32_t> cig_off, base_off8, name_off, cigar;
    pvector<int64_t> mpos;
    pvector<uint8_t> seq, qual, bq;
    pvector<char> names;
    pvector<uint32_t> xcol_off; pvector<char> xcol_text; int n_xcols = 0;
    // --output-mods (host_mods.cpp): entries of read i = mod_off[i] .. mod_off[i+1], sorted by query position
    pvector<uint32_t> mod_off, mod_qpos, mod_toff; pvector<char> mod_text; bool with_mods = false;
    bool any_bq = false;
    // template state kept by the input lane (host_names.h; sta_reads.olap_clip / olap_mate): tpl = 1: clip[] filled (depth -s), 2: mate[] filled
    // (mpileup overlaps), 0: neither -- the engine then replays the name hash from the staged names
    pvector<int64_t> clip; pvector<int32_t> mate; int tpl = 0;
    void clear();
    // origin: absolute coordinate of relative 0; rg_excl: -G read groups to drop (may be null)
    void add(const Rec &r, int64_t origin, const std::set<std::string> *rg_excl, const XcolSpec *xs = nullptr);
    // bulk form of add() for records [i0, i1) of a decoded chunk (no read-group list, no RNEXT / modification columns): pool
    // slices are copied whole and the offsets rebased; the chunk's aux-tag text becomes the tag columns
    void add_range(const Chunk &c, int64_t i0, int64_t i1, int64_t origin, const XcolSpec *xs = nullptr);     // xs: tag columns only
    // the same for a window's whole list of slices: sizes first, every pool grown once, then the slices copied by `threads`
    // threads (each slice knows its destination offsets after the prefix sums)
    struct Slice { const Chunk *c; int64_t i0, i1; };
    // largest pool sizes any window of this input has needed so far: a staging object that has to grow (every pool growth is a
    // page-locking call) grows once, to a quarter beyond these, instead of creeping up window by window in each ring slot
    struct PoolSizes { size_t rec = 0, cig = 0, b8 = 0, nm = 0, xoff = 0, xtext = 0; };
    // raw_mode (host_chunk.h): 1 = the slices' pool bytes are NOT copied: the chunks' raw alignment records are handed to the engine
    // instead (sta_reads.raw_*), which cuts the pools out of them on the device; 2 = both, and the engine compares.  Falls back to
    // copying when a slice has no usable raw bytes (SAM input, a CIGAR from a CG tag) or BQ:Z values are around.
    void add_ranges(const Slice *g, size_t n_g, int64_t origin, const XcolSpec *xs, int threads, size_t min_bytes_for_threads = (size_t)4 << 20,
                    PoolSizes *high_water = nullptr, int raw_mode = 0);
    // device staging of this window's new reads (set by add_ranges)
    int64_t raw_first = -1; int raw_verify = 0;
    std::vector<sta_raw_piece> raw_pieces;
    pvector<uint32_t> raw_rec_off;
    std::vector<std::shared_ptr<pvector<uint8_t>>> raw_keep;      // the chunks' buffers stay alive while the window may still be uploaded
    void finish();                 // closes the offset arrays
    sta_reads view() const;        // pointers into this object (valid until the next add/clear)
    int64_t n() const { return (int64_t)pos.size(); }
};

}  // namespace sta
