// sta_dev.h -- internal device-side view of a staged window and kernel launchers.
// Not part of the C-ABI (include/samtools_amd.h is).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/samtools_amd.h"

// per-read packed info word produced by k_prep_reads
// internal bit of sta_mplp_params.flag (never part of the C ABI): the pipeline runs for calmd -- no "read starts beyond the FASTA
// contig" skip (that is mplp_func's, bam_plcmd.c:440-445; bam_md.c:461-476 hands every placed record to sam_prob_realn)
#define STA_MPLP_INT_CALMD (1 << 30)
#define RI_PUSHED   0x1u    // passed mplp_func filters -> reaches bam_plp_push
#define RI_KEEP     0x2u    // in the pileup: pushed, reference span > 0, not dropped by -d cap
#define RI_SIMPLE   0x4u    // CIGAR is a single M/=/X op
#define RI_REV      0x8u
#define RI_OLAP_EL  0x10u   // eligible for mate-overlap hashing (overlap_push conditions)
#define RI_BAQ      0x20u   // BAQ must be computed for this read
#define RI_UNMAP_SPAN 0x80u // depth: R.end is the CIGAR's reach, the covered span (bam_endpos) is one column
#define RI_BAQ_SLOW 0x40u   // ... by the general-band kernel (band width != 7 or very long read); listed in `chain`
#define RI_BAQ_BW_SHIFT 16  // bits 16..20: band width handled by a band-in-registers BAQ kernel (7 or 8), 0 = general kernel
#define RI_MAPQ_SHIFT 8     // bits 8..15: mapping quality after -C
#define RI_BAQ_S    0x200000u  // BAQ by the class-S kernel (baq_band7s.h: one M operation, unclipped window, band width 7, one read length per group of 64)

// one input file's reads as the kernels see them (all device pointers)
struct StaReadsDev {
    int64_t n;
    const int32_t *pos;
    const uint16_t *flag;
    const uint8_t *mapq;
    const uint8_t *aux;
    const int32_t *l_qseq;
    const uint32_t *cig_off;
    const uint32_t *base_off8;
    const int32_t *mtid;
    const int64_t *mpos;
    const int32_t *isize;
    const uint32_t *name_off;
    const uint32_t *cigar;
    const uint8_t *seq;
    const uint8_t *qual_in;
    const uint8_t *bq;
    const char *names;
    uint64_t n_bases_total;
    int32_t n_xcols; const uint32_t *xcol_off; const char *xcol_text;     // host-formatted text columns (RNEXT, aux tags)
    const uint32_t *mod_off, *mod_qpos, *mod_toff; const char *mod_text;  // --output-mods: per-read modification text (NULL: none)
    // what every read found in the reference's name hash, kept by the caller in file order (sta_reads.olap_clip / olap_mate; NULL: the
    // device replays the hash from the staged names, kernels_overlap.hip)
    const int64_t *clip_in;     // depth -s: absolute column below which the read is not counted (0: none)
    const int32_t *mate;        // mpileup: read whose overlap-hash entry this one found (-1: none)
    // engine workspace
    uint8_t *qual;        // working qualities (== qual_in when nothing rewrites them)
    int32_t *end;         // pos + reference span
    int32_t *maxend;      // inclusive prefix max of `end` over RI_KEEP reads
    uint32_t *info;       // RI_*
    int32_t *clip;        // depth -s: columns below this are not counted (0 = none)
    // mate-overlap visibility fix-up (see placeholder_qual in dev_util.h): for a read whose deletion / ref-skip run straddles
    // the start of its mate, the query index the run's placeholders look at (-1 = none), that base's quality BEFORE the
    // pair was resolved, and the mate's read index.  NULL when overlaps are off.
    int32_t *fix_y, *fix_mate; uint8_t *fix_q;
    int32_t *chain;       // n+4 ints: BAQ slow-read list ([0] = count, then read indices) until the overlap pass reuses it as hash chains
    // class-S BAQ (round 5): the candidates are gathered per read length into dense groups of 64.  STA_SLIST_BINS words each of: the
    // histogram of candidate lengths (k_prep_reads), the first list position of a length (host: every length starts a new group of 64),
    // the gather kernel's cursors; then the list itself (read indices, -1 = padding).  NULL: no class-S kernel in this plan.
    int32_t *s_ws;
};
#define STA_SLIST_BINS 260                 // read lengths 0..256 (class S: 16..256)
#define STA_SLIST_HEAD (3 * STA_SLIST_BINS)

// window constants shared by the column kernels
struct StaWinDev {
    int32_t col_beg, col_end;
    int64_t origin;
    int32_t tid;
    int64_t tlen;
    int32_t nfiles;
    const StaReadsDev *files;       // device array
    const char *ref;                // contig sequence indexed by absolute position (or nullptr)
    int64_t ref_len;
    const char *tname; int32_t tname_len;
    int32_t has_bed; int64_t n_bed; const int64_t *bed_beg, *bed_end;
    int32_t has_reg; int64_t reg_beg, reg_end;
    int32_t baq_plain;              // 0: extended BAQ (what mpileup always asks for, realn.c flag bit 2); 1: per-base BAQ (calmd -r without -E)
};

struct StaCounters {          // device-side reduction targets, zeroed per plan
    unsigned long long n_lines, n_data_cols, n_kept, piled_bases, n_dropped, max_wave_bytes, n_anom, maxcnt_flag, max_lq, max_bw, n_baq, max_lq_fast, n_baq_fast, n_baq_bw8, n_baq_general, n_baq_s, max_lq_s, n_baq_bw7l;
    unsigned long long out_bytes, overflow;      // single-pass kernels: total text bytes of the window; set when the output buffer was too small
    unsigned long long n_olap_el;                // mpileup: reads eligible for the mate-overlap pass (RI_OLAP_EL), the gate of that pass
};

// one BGZF block for k_bgzf_inflate (kernels_inflate.hip): deflate bytes comp[comp_off, comp_off + clen) -> out[out_off, out_off + isize).
// The compressed buffer must be readable 1 KiB beyond the last block's data (the decoder's input window runs ahead).
struct StaBgzfBlock { uint64_t comp_off; uint32_t clen, isize; uint64_t out_off; };
void sta_launch_bgzf_inflate(hipStream_t s, const uint8_t *comp, const StaBgzfBlock *blocks, int n_blocks, uint8_t *out, uint32_t *status);

// ---- launchers (defined in the .hip files) ----
// in-kernel prefix maximum of the preparation kernels (kernels_common.hip ChunkScan): STA_CHUNK_WORDS zero-initialised 8-byte words
// on the device, and the host's view of the ticket counter / launch number kept in them
#define STA_CHUNK_WORDS (1 + 2048 + 7)
struct StaChunkState { unsigned long long *words = nullptr; unsigned long long tickets = 0; uint32_t epoch = 0; };
// R.end / R.info / R.maxend of every file; with `wfirst` also the tile kernels' column -> read index (returns false when no launch
// could carry it: every file empty)
bool sta_launch_prep_reads(hipStream_t s, const StaWinDev &w, const StaReadsDev *files_host, int nfiles, const sta_mplp_params &p, StaCounters *ctr,
                           StaChunkState &st, uint32_t *wfirst);
// returns true when it also cleared `zero` (the depth kernel's look-back status): sta_launch_depth_fused then skips its memset
bool sta_launch_prep_reads_depth(hipStream_t s, const StaWinDev &w, const StaReadsDev *files_host, int nfiles,
                                 const sta_depth_params &p, StaCounters *ctr, StaChunkState &st, void *zero = nullptr, size_t zero_bytes = 0);
void sta_launch_cap_mapq(hipStream_t s, const StaReadsDev &r, const StaWinDev &w, int thres, int min_mq, StaCounters *ctr);
void sta_launch_cap_mapq_vals(hipStream_t s, const StaReadsDev &R, const StaWinDev &w, int thres, int16_t *cap);
void sta_launch_qual_prep(hipStream_t s, const StaReadsDev &r, int illumina13);
void sta_launch_maxend_scan(hipStream_t s, const StaReadsDev &r, void *tmp, size_t tmp_bytes);
size_t sta_scan_tmp_bytes(int64_t n);
void sta_launch_scan_max_i32(hipStream_t s, const int32_t *in, int32_t *out, int64_t n, void *tmp);
// exclusive scan of u32 lengths into u64 offsets (offs has n+1 entries)
void sta_launch_len_scan(hipStream_t s, const uint32_t *len, uint64_t *offs, int64_t n, void *tmp, size_t tmp_bytes);
// The measuring pass.  With wfirst / status / offs (and an option set the tile kernels cover) it is k_mplp_len_rm + k_tile_scan and returns
// true: colinfo, TILE-RELATIVE row offsets + tile bases (sta_mplp_tile_base), n_lines, n_data_cols and max_wave_bytes are all produced, the
// caller skips the scan and column-statistics launches.  Otherwise the generic walker k_mplp_len (line lengths only), false.
bool sta_launch_mplp_len(hipStream_t s, const StaWinDev &w, const sta_mplp_params &p, uint32_t *line_len, uint2 *colinfo /*[nfiles][ncols] (count, seq bytes)*/,
                         StaCounters *ctr, const uint32_t *wfirst /* sta_launch_wave_first's table */, void *status /* sta_mplp_len_status_bytes() */, uint64_t *offs,
                         int detect_maxcnt /* > 0: also run the -d detector (kernels_maxcnt.hip) for this cap */,
                         uint32_t *gen_xlen = nullptr /* generic walker: [nfiles][sta_mplp_generic_extras()][ncols] bytes per extra column, kept for the emit */);
int sta_mplp_generic_extras(const sta_mplp_params &p);
// wfirst[nfiles][ncols / 64 + 2]: first read starting at or beyond every 64-column group (where the tile kernels start looking)
void sta_launch_wave_first(hipStream_t s, const StaWinDev &w, uint32_t *wfirst, void *status);
size_t sta_mplp_len_status_bytes(int64_t ncols);
const uint64_t *sta_mplp_tile_base(const void *status, int64_t ncols);
int64_t sta_mplp_deep_strips(int64_t ncols);      // strips of the read-major emit kernel; strip_rng holds 2 x int64 per (file, strip)
void sta_launch_mplp_emit(hipStream_t s, const StaWinDev &w, const sta_mplp_params &p, const uint64_t *offs, const uint2 *colinfo,
                          char *out, uint32_t lds_cap /* generic walker's slice */, int64_t *strip_rng /* workspace of the read-major kernel */, uint32_t tile_cap, int deep_mode,
                          const uint32_t *wfirst, const uint64_t *tbase /* sta_mplp_tile_base() */, bool tile /* the measuring pass returned true */,
                          const uint32_t *gen_xlen = nullptr /* what sta_launch_mplp_len kept (NULL: the generic emit measures for itself) */);
bool sta_mplp_has_fast_path(const sta_mplp_params &p);
bool sta_mplp_has_xfast_path(const sta_mplp_params &p);      // extra columns on the read-major kernels (k_mplp_len_rm<true> + k_mplp_emit_deep<true>)
int sta_mplp_xfast_extras(const sta_mplp_params &p);         // their number (0: not that path): gen_xlen holds [nfiles][this][ncols] words
bool sta_mplp_tile_ok(const sta_mplp_params &p);
void sta_launch_wave_bytes_max(hipStream_t s, const uint64_t *offs, const uint32_t *line_len, int64_t ncols, StaCounters *ctr);

// binary per-column entries for the bam_plp_* surface (kernels_plpapi.hip)
void sta_launch_plp_count(hipStream_t s, const StaWinDev &w, uint32_t *line_len);
void sta_launch_plp_fill(hipStream_t s, const StaWinDev &w, const uint64_t *offs, void *entries);

// coverage / bedcov column reductions (kernels_cov.hip)
void sta_launch_glf_cols(hipStream_t s, const StaWinDev &w, int min_baseQ, int capQ, const char *ref, int64_t ref_len,
                         const double *fk, const double *beta, const double *lhet, void *out, uint8_t *redo, double mean_depth);
size_t sta_glf_redo_bytes(const StaWinDev &w);
void sta_launch_calmd_tag(hipStream_t s, const StaReadsDev &r, int apply, uint8_t *tag_pool, uint8_t *state, const uint8_t *bq_pool);
void sta_launch_md_len(hipStream_t s, const StaReadsDev &r, const StaWinDev &w, int32_t *nm, uint32_t *md_len, uint8_t *state);
void sta_launch_md_emit(hipStream_t s, const StaReadsDev &r, const StaWinDev &w, int use_equal, int bin_qual, int max_nm,
                        const int32_t *nm, const uint64_t *md_off, char *md_text, uint8_t *seq_work);
void sta_launch_cov_cols(hipStream_t s, const StaWinDev &w, int mode, int min_baseQ, int min_depth, int skip_dn,
                         unsigned long long *totals, unsigned long long *per_file,
                         uint32_t *hist, int hist_bins, int hist_depth, int64_t hist_beg, int64_t hist_bin_width);

// `stats` coverage distribution from sorted marks (kernels_statcov.hip)
size_t sta_statcov_tmp_bytes(int64_t n);
void sta_launch_statcov(hipStream_t s, const int64_t *pos, const int32_t *delta, int64_t n, long long carry_in,
                        int cov_min, int cov_max, int cov_step, int ncov, unsigned long long *cov, void *tmp);

// overlap (mate) resolution
size_t sta_overlap_table_slots(int64_t n_reads);
size_t sta_overlap_table_bytes(size_t slots);
// mpileup: the pass's setup, gated on StaCounters.n_olap_el on the device (kernels_overlap.hip k_olap_setup); `dev_file` = the file's
// entry of StaWinDev.files, `table` must hold sta_overlap_table_bytes(slots)
void sta_launch_overlap_setup(hipStream_t s, const StaReadsDev &r, StaReadsDev *dev_file, bool copy_qual, void *table, size_t slots, const StaCounters *ctr);
void sta_launch_overlap(hipStream_t s, const StaReadsDev &r, int64_t origin, int32_t tid, void *table, size_t slots,
                        int32_t *chain_next, StaCounters *ctr);
// -d cap
void sta_launch_maxcnt_detect(hipStream_t s, const StaReadsDev &r, int maxcnt, StaCounters *ctr);
void sta_launch_maxcnt(hipStream_t s, const StaReadsDev &r, int maxcnt, int32_t col_lo, int32_t ncols_span,
                       int32_t *scratch, StaCounters *ctr);
// BAQ
size_t sta_baq_scratch_bytes(int64_t n_reads, int max_lq, int max_bw);
void sta_launch_baq(hipStream_t s, const StaReadsDev &r, const StaWinDev &w, int redo, void *scratch, size_t scratch_bytes,
                    int lq_max, int bw_max, int64_t n_slow);
// band-in-registers kernels (band width 7: reads taken directly; 8: through the list in `chain`)
#define STA_BAQ7_LQ_MAX 2048
size_t sta_baq_band_scratch_bytes(int64_t n_reads, int lq_cap, int *groups_per_launch, int slab_gib_cap /* 0 = $STA_BAQ_SLAB_GIB or 48 */);
void sta_launch_baq_band(hipStream_t s, const StaReadsDev &r, const StaWinDev &w, void *scratch, int lq_cap, int bw,
                         int64_t g0, int64_t ng, int use_list, int pass /*0 forward, 1 backward*/);

void sta_launch_baq_list(hipStream_t s, const StaReadsDev &r, const StaWinDev &w, void *scratch, int lq_cap, int bw, int64_t ng,
                         const int32_t *range = nullptr /* device: [lo, hi) of the list groups to look at */);   // both passes, list reads of band width bw
// the list into class order (band width 7 | 8 | general); tmp = chain[0] + 4 words, the last four receive the group ranges
// [lo7, hi7, lo8, hi8] for sta_launch_baq_list
void sta_launch_baq_list_partition(hipStream_t s, const StaReadsDev &r, int32_t *tmp);
// class S (baq_band7s.h): one fused persistent kernel over all groups of 64 reads; scratch = 256-byte header + two slots per resident wave
size_t sta_baq7s_scratch_bytes(int lq_cap, int64_t ngroups, int *waves_out);
void sta_launch_baq7s(hipStream_t s, const StaReadsDev &r, const StaWinDev &w, void *scratch, int lq_cap, int waves, int64_t ngroups);
// the class-S candidates of a file (RI_BAQ_S) into r.s_ws's list at the positions the host laid out per read length
void sta_launch_baq7s_gather(hipStream_t s, const StaReadsDev &r);

// device staging (kernels_stage.hip): pools of reads [raw_first, raw_first + n_raw) out of raw BAM alignment records
void sta_launch_bam_pools(hipStream_t s, const uint8_t *raw, const uint32_t *rec_off, uint64_t raw_bytes, int64_t raw_first, int64_t n_raw, const StaReadsDev &d,
                          uint32_t *cigar, uint8_t *seq, uint8_t *qual, char *names, unsigned long long *bad);
void sta_launch_stage_compare(hipStream_t s, const void *a, const void *b, uint64_t n, unsigned long long *bad);

// depth
size_t sta_depth_fused_status_bytes(int64_t ncols);
void sta_launch_depth_fused(hipStream_t s, const StaWinDev &w, const sta_depth_params &p, void *status, int32_t *counts, char *out,
                            uint64_t capacity, StaCounters *ctr, uint32_t lbuf, bool status_zeroed = false,
                            uint32_t *lens = nullptr /* ncols words: with it, windows of many tiles take the split form (count | scan | emit) */);
void sta_launch_depth_pair(hipStream_t s, const StaReadsDev &r, int64_t origin, int32_t tid, void *table, size_t slots,
                           int32_t *chain_next, StaCounters *ctr);

// consensus (kernels_cons.hip; the window record and the step functions are in cons_window.h)
namespace cons { struct Win; struct Par; struct Tables; }
void sta_launch_cons_read_a(hipStream_t s, const cons::Win &w, const cons::Par &o, const cons::Tables *t, bool walk_all);
void sta_launch_cons_prepare(hipStream_t s, const cons::Win &w, const cons::Par &o, const cons::Tables *t, int32_t *gran2read /* n_bases / 8 + 1 words */, int64_t n_bases);
void sta_launch_cons_collen(hipStream_t s, const uint32_t *ins, uint32_t *len, int64_t W);
void sta_launch_cons_read_b(hipStream_t s, const cons::Win &w);
void sta_launch_cons_colpos(hipStream_t s, const cons::Win &w);
void sta_launch_cons_walk(hipStream_t s, const cons::Win &w, const cons::Par &o, int64_t n_list, bool so_words);
void sta_launch_cons_col(hipStream_t s, const cons::Win &w, const cons::Par &o, const cons::Tables *t, int64_t n_cols);
void sta_launch_cons_text(hipStream_t s, const cons::Win &w, const cons::Par &o, int64_t n_cols);
