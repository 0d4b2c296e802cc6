/*
 * samtools_amd.h -- C-ABI of the MI355X-native mpileup/depth engine.
 *
 * This is the drop-in boundary for the samtools pileup hot path
 * (BASELINE.json north_star).  Plain C, plain pointers
 * and sizes, no torch / C++ types.  The functions below replace, in bulk
 * ("window at a time") form, what the reference does one column at a time:
 *
 *   reference interface (file:line)                      replaced by
 *   ---------------------------------------------------  ---------------------------
 *   mplp_func read callback        bam_plcmd.c:400-461    sta_stage_window + prep/BAQ kernels
 *   bam_mplp_init/_auto/...        bam_plcmd.c:581-607    sta_mpileup_plan / sta_mpileup_emit
 *   bam_mplp_set_maxcnt            bam_plcmd.c:597        sta_mplp_params.max_depth
 *   bam_mplp_init_overlaps         bam_plcmd.c:586        STA_MPLP_SMART_OVERLAPS
 *   sam_prob_realn (BAQ)           bam_plcmd.c:451        STA_MPLP_REALN / STA_MPLP_REDO_BAQ
 *   sam_cap_mapq (-C)              bam_plcmd.c:453-457    sta_mplp_params.capQ_thres
 *   --output-extra tags / RNEXT    bam_plcmd.c:779-852    sta_reads.xcol_* + sta_mplp_params.n_tags
 *   --output-mods (MM / ML)        bam_plcmd.c:86-109,356 sta_reads.mod_* + STA_MPLP_OUTPUT_MODS
 *   mpileup() column loop+format   bam_plcmd.c:607-868    sta_mpileup_emit / sta_mpileup_run (text on device)
 *   pileup_seq                     bam_plcmd.c:54-169     sta_mpileup_emit
 *   print_empty_pileup             bam_plcmd.c:372-398    sta_mpileup_emit with params.all
 *   mplp_get_ref                   bam_plcmd.c:289-352    sta_set_reference
 *   add_depth / incr_hist[_qual]   bam2depth.c:165-477    sta_depth_plan / sta_depth_emit
 *   fastdepth_core merge + -s hash bam2depth.c:486-699    sta_depth_plan (per-file read sets)
 *   bed_overlap                    bedidx.c:159-197       sta_window.bed_* (merged intervals)
 *   coverage / bedcov column loops coverage.c:621-672, bedcov.c:316-333   sta_cov_plan
 *   bcf_call_init / bcf_call_glfgen bam2bcf.c:38-48,65-123 sta_glf_plan (+ sta_glf_consensus, bam_tview.c:194-212)
 *   bam_fillmd1_core + BAQ tag     bam_md.c:64-224,474-479 sta_calmd_plan / sta_fetch_calmd
 *   stats coverage round buffer    stats.c:311-391,1452-1508 sta_statcov_begin / _add / _fetch
 *   sam_open / sam_read1 (host)    bam_plcmd.c:500-569    sta_io_scan exercises the drivers' reader (BGZF worker pool)
 *   bam_mpileup / main_depth (CLI) bam_plcmd.c:1075, bam2depth.c:732   sta_main_mpileup / sta_main_depth
 *   pileup_loop + get_next_base    consensus_pileup.c:69-608            sta_consensus_run (columns incl. insertion columns)
 *   nm_init / calculate_consensus_* / consensus_base   bam_consensus.c:1012-2183   sta_consensus_run
 *   main_consensus (CLI)           bam_consensus.c:3149                 sta_main_consensus
 *
 * The per-column callback surface (bam_plp_* / bam_mplp_* / bam_plbuf_*) is
 * declared in samtools_amd_plp.h.
 *
 * Threading: one engine is used from one host thread (like bam_mplp_t).
 * Errors: functions return 0 on success, negative on failure; the message is
 * available from sta_last_error().  There is NO CPU fallback: every compute
 * entry point fails with STA_ERR_NO_DEVICE if no HIP device is usable.
 */
#ifndef SAMTOOLS_AMD_H
#define SAMTOOLS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STA_OK 0
#define STA_ERR_NO_DEVICE (-2)
#define STA_ERR_HIP (-3)
#define STA_ERR_ARG (-4)
#define STA_ERR_UNSORTED (-5)
#define STA_ERR_UNSUPPORTED (-6)
#define STA_ERR_IO (-7)

/* mplp_conf_t.flag bits, same values as bam_plcmd.c:173-201 */
#define STA_MPLP_NO_ORPHAN       (1 << 3)
#define STA_MPLP_REALN           (1 << 4)
#define STA_MPLP_REDO_BAQ        (1 << 6)
#define STA_MPLP_ILLUMINA13      (1 << 7)
#define STA_MPLP_SMART_OVERLAPS  (1 << 10)
#define STA_MPLP_PRINT_MAPQ_CHAR (1 << 11)
#define STA_MPLP_PRINT_QPOS      (1 << 12)
#define STA_MPLP_PRINT_QNAME     (1 << 13)
#define STA_MPLP_PRINT_FLAG      (1 << 14)
#define STA_MPLP_PRINT_RNAME     (1 << 15)
#define STA_MPLP_PRINT_POS       (1 << 16)
#define STA_MPLP_PRINT_MAPQ      (1 << 17)
#define STA_MPLP_PRINT_RNEXT     (1 << 19)   /* text comes from sta_reads.xcol_* column 0 */
#define STA_MPLP_PRINT_PNEXT     (1 << 20)
#define STA_MPLP_PRINT_RLEN      (1 << 24)
#define STA_MPLP_OUTPUT_MODS     (1 << 25)   /* -M / --output-mods: sta_reads.mod_* carries the text */
#define STA_MPLP_PRINT_QPOS5     (1 << 26)

/* per-read aux summary bits (sta_reads.aux) computed while decoding */
#define STA_AUX_HAS_BQ   1   /* BQ:Z present (and usable)         */
#define STA_AUX_HAS_ZQ   2   /* ZQ:Z present                      */
#define STA_AUX_SKIP     4   /* excluded by -G read-group list    */
#define STA_AUX_ACCEPTED 8   /* carried over from an earlier window of the same iterator, where the -d cap (bam_plp_push)
                              * already accepted it: the cap is not re-tested, the read still counts as live */
#define STA_AUX_ZQ_RESTORE 16 /* calmd -r without -A: the read's slice of the bq pool holds its ZQ:Z string; the qualities get the
                              * stored difference back and the tag becomes BQ:Z (realn.c's ZQ -> BQ branch)          */

/* where the arrays of a sta_reads / sta_window live */
#define STA_MEM_HOST   0     /* engine copies them to HBM (pinned staging) */
#define STA_MEM_DEVICE 1     /* already resident in HBM; used in place     */

/*
 * Pre-decoded alignment records of ONE input file for ONE window, structure
 * of arrays, sorted by pos (file order preserved among equal pos).  Record
 * field meaning follows bam1_core_t.
 *
 *  pos[i]        0-based leftmost coordinate MINUS window origin (int32)
 *  cig_off[i]    index of read i's first op in cigar[]; cig_off[n] = total ops
 *  base_off8[i]  (offset of read i's first base)/8 in qual[]/bq[]; the 4-bit
 *                packed sequence starts at seq[base_off8[i]*4].  Every read's
 *                base run is padded to a multiple of 8 bases.
 *  name_off[i]   offset of read i's qname in names[]; name_off[n] = total
 */
typedef struct sta_reads {
    int64_t n_reads;
    const int32_t *pos;
    const uint16_t *flag;
    const uint8_t *mapq;
    const uint8_t *aux;
    const int32_t *l_qseq;
    const uint32_t *cig_off;     /* n_reads + 1 */
    const uint32_t *base_off8;   /* n_reads */
    const int32_t *mtid;
    const int64_t *mpos;         /* absolute mate position (bam1_core_t.mpos) */
    const int32_t *isize;        /* clamped to int32 */
    const uint32_t *name_off;    /* n_reads + 1 */
    const uint32_t *cigar;       /* len<<4|op, op in MIDNSHP=XB */
    const uint8_t *seq;          /* 4-bit packed, high nibble first */
    const uint8_t *qual;         /* Phred, 0xff = absent */
    const uint8_t *bq;           /* optional BQ:Z values ('@' = 64 where absent), NULL if no read has one */
    const char *names;           /* NUL-terminated qnames */
    uint64_t n_cigar_total, n_bases_total /* padded */, n_name_bytes;
    /* optional host-formatted text columns (--output-extra RNEXT and aux tags, bam_plcmd.c:779-784,:798-852): n_xcols
     * columns per read; the text of (read i, column c) is xcol_text[xcol_off[i*n_xcols+c] .. xcol_off[i*n_xcols+c+1]).
     * Column 0 is RNEXT when STA_MPLP_PRINT_RNEXT is set, the tags follow in --output-extra order.  n_xcols = 0: none. */
    int32_t n_xcols;
    const uint32_t *xcol_off;    /* n_reads * n_xcols + 1 */
    const char *xcol_text;
    uint64_t n_xcol_bytes;
    /* optional base modifications for --output-mods (bam_plcmd.c:86-109 prints what HTSlib's bam_mods_at_qpos returns; here the
     * MM / ML tags are evaluated on the host while staging): for every modified base of every read the text pileup_seq appends,
     * e.g. "[+m128-h7]".  Entries of read i: mod_off[i] .. mod_off[i+1], sorted by mod_qpos (query position); text of entry e:
     * mod_text[mod_toff[e] .. mod_toff[e+1]).  mod_off == NULL: none. */
    const uint32_t *mod_off;     /* n_reads + 1 */
    const uint32_t *mod_qpos;    /* n_mod_entries */
    const uint32_t *mod_toff;    /* n_mod_entries + 1 */
    const char *mod_text;
    uint64_t n_mod_entries, n_mod_bytes;
    /* optional (STA_MEM_HOST only): reads [raw_first, n_reads) are staged on the DEVICE out of raw BAM records.  Their scalars and
     * offsets (pos .. name_off) are filled in as for any read, but their slices of the pools cigar / seq / qual / names are left
     * unwritten by the host: the engine uploads the inflated BAM bytes they lie in (`raw_pieces`, concatenated in this order)
     * and a HIP kernel copies every record's CIGAR, 4-bit bases, qualities and name from its alignment record (SAM spec 4.2:
     * block_size | refID pos l_read_name mapq bin n_cigar_op flag l_seq next_refID next_pos tlen | read_name | cigar | seq | qual | aux)
     * to the pool offsets cig_off / base_off8 / name_off name, zero-padding the bases to 8 (the feed this replaces on the host:
     * sam_read1 in bam_plcmd.c:409, bam2depth.c:541-543).  raw_rec_off[i - raw_first] = byte offset, inside the concatenated pieces,
     * of read i's refID field.  The pools' host-written prefixes are cigar[0 .. cig_off[raw_first]), bases [0 .. 8 * base_off8[raw_first]),
     * names[0 .. name_off[raw_first]).  raw_first == n_reads or n_raw_pieces == 0: everything was staged by the host (the default).
     * A read whose CIGAR lives in a CG:B,I tag, or a window with BQ:Z values, must not be staged this way. */
    int64_t raw_first;
    int32_t n_raw_pieces;
    const struct sta_raw_piece *raw_pieces;
    const uint32_t *raw_rec_off;
    int32_t raw_verify;          /* != 0: the host pools ARE fully written as well; the engine compares its device-built pools with them and fails on a difference (tests) */
    /* optional: what every read found in the reference's read-name hash, worked out by the caller in FILE ORDER over the whole input.
     * Both hashes are sequential state (an entry put by a record a million columns upstream decides what a later record of the
     * template finds), so a caller that cuts the input into windows keeps them itself and hands over the outcome:
     *   olap_clip[i]  depth -s (bam2depth.c:598-623): absolute column below which read i is not counted -- the end position the
     *                 first-seen record of its template left in the hash -- or 0.  Used by sta_depth_plan when remove_overlaps is set.
     *   olap_mate[i]  mpileup with overlap detection (HTSlib overlap_push, enabled at bam_plcmd.c:586): index, in this file's reads,
     *                 of the record whose hash entry read i found at its push -- the pair tweak_overlap_quality(olap_mate[i], i)
     *                 resolves -- or -1.  Used by the plans that resolve overlaps (STA_MPLP_SMART_OVERLAPS, sta_plp_plan).
     * NULL (the default): the engine replays the hash itself from the names of the staged reads, which is exact when the window
     * holds every record of the templates it touches (a whole contig staged at once, as bench.py does), and not otherwise. */
    const int64_t *olap_clip;
    const int32_t *olap_mate;
} sta_reads;

typedef struct sta_raw_piece { const uint8_t *bytes; uint64_t n_bytes; } sta_raw_piece;

/* One window of reference columns on one contig, with every read (of every
 * input file) whose reference span can touch it. */
typedef struct sta_window {
    int32_t tid;
    int64_t origin;              /* absolute coordinate of relative column 0 */
    int32_t col_beg, col_end;    /* relative columns to produce: [col_beg, col_end) */
    const char *tname;           /* contig name (host pointer, always) */
    int64_t tlen;                /* contig length from the header */
    int32_t n_files;
    const sta_reads *files;      /* host array of n_files descriptors */
    int32_t mem;                 /* STA_MEM_HOST or STA_MEM_DEVICE: where files[].arrays live */
    /* -l/-b BED: merged, sorted, disjoint intervals of this contig (absolute, host pointers) */
    int32_t has_bed;
    int64_t n_bed;
    const int64_t *bed_beg, *bed_end;
    /* -r region clip (absolute, half open); has_reg = 0 for none */
    int32_t has_reg;
    int64_t reg_beg, reg_end;
} sta_window;

/* subset of mplp_conf_t (bam_plcmd.c:206-216) the device path consumes */
typedef struct sta_mplp_params {
    int32_t min_mq, min_baseQ, capQ_thres, max_depth;
    int32_t all, rev_del;
    int32_t rflag_require, rflag_filter;
    int32_t flag;                /* STA_MPLP_* */
    int32_t no_ins, no_del, no_ends;
    int32_t has_fai;             /* a FASTA was given with -f (BAQ/ref column need it) */
    int32_t n_tags;              /* aux-tag columns after the fixed extra columns (text staged in sta_reads.xcol_*) */
    int32_t tag_sep;             /* --output-sep character between the entries of a tag column (',' by default) */
    int32_t min_qlen;            /* coverage -l: drop reads whose bam_cigar2qlen is smaller (0 = off) */
    int32_t no_ins_mods;         /* --no-output-ins-mods: no modification text inside inserted sequences */
} sta_mplp_params;

/* subset of depth_opt (bam2depth.c:72-86) */
typedef struct sta_depth_params {
    int32_t flag, incl_flag, require_flag;
    int32_t min_qual, min_mqual, min_len;
    int32_t skip_del, all_pos, remove_overlaps;
} sta_depth_params;

/* what a plan call learned about the window */
typedef struct sta_plan_info {
    uint64_t out_bytes;          /* text bytes the emit call will produce          */
    uint64_t n_lines;            /* output rows                                    */
    uint64_t n_data_cols;        /* columns with >=1 read (mpileup) / covered (depth) */
    uint64_t n_kept_reads;       /* reads that passed the read-level filters       */
    uint64_t piled_bases;        /* sum of reference span of kept reads in-window  */
    uint64_t n_maxcnt_dropped;   /* reads dropped by the -d cap                    */
} sta_plan_info;

typedef struct sta_engine sta_engine;

/* ---- engine ---- */
int sta_engine_create(sta_engine **out, int device, void *hip_stream);
void sta_engine_destroy(sta_engine *e);
const char *sta_last_error(const sta_engine *e);
/* number of usable HIP devices (0 when there is none); never throws */
int sta_device_count(void);
const char *sta_version(void);

/* ---- reference sequence (mplp_get_ref) ---- */
/* Copies contig `tid` (raw FASTA characters) to HBM; kept until replaced or
 * sta_clear_references().  mem selects host or device source pointer. */
int sta_set_reference(sta_engine *e, int32_t tid, const char *seq, int64_t len, int32_t mem);
void sta_clear_references(sta_engine *e);

/* ---- staging ---- */
/* Makes `w` the current window: uploads (STA_MEM_HOST) or adopts
 * (STA_MEM_DEVICE) the read arrays.  Asynchronous on the engine stream. */
int sta_stage_window(sta_engine *e, const sta_window *w);

/* ---- mpileup ---- */
/* Runs read prep (filters, reference span), quality prep (-6, BQ tags), BAQ,
 * -d cap, mate-overlap resolution and the per-column measuring pass; returns
 * sizes.  Synchronises the stream. */
int sta_mpileup_plan(sta_engine *e, const sta_mplp_params *p, sta_plan_info *info);
/* Writes the pileup text of the planned window into dev_out (device pointer,
 * capacity >= out_bytes); dev_out == NULL uses an engine-owned buffer that
 * sta_fetch_output() copies back.  Asynchronous on the engine stream. */
int sta_mpileup_emit(sta_engine *e, void *dev_out, uint64_t capacity);
/* Plan and emit in one call: the staged window's text goes to dev_out (device pointer; the exact size is only known once
 * the window is planned, so a capacity that turns out too small is STA_ERR_ARG) or, with dev_out == NULL, to an engine-owned
 * buffer (read it with sta_fetch_output).  info receives the same counters as sta_mpileup_plan.  Synchronises the stream. */
int sta_mpileup_run(sta_engine *e, const sta_mplp_params *p, void *dev_out, uint64_t capacity, sta_plan_info *info);

/* ---- binary per-column pileup entries (the bam_plp_* / bam_mplp_* surface is built on these; see
 *      samtools_amd_plp.h).  One input file per window.  Replaces bam_plp_push's admission rules
 *      (unmapped reads dropped, -d cap, mate-overlap resolution) + bam_plp64_next/resolve_cigar2. ---- */
typedef struct sta_plp_entry {
    int32_t read;      /* index into the staged reads of the window                      */
    int32_t qpos;      /* bam_pileup1_t.qpos                                             */
    int32_t indel;     /* bam_pileup1_t.indel                                            */
    uint32_t bits;     /* 1 is_del, 2 is_head, 4 is_tail, 8 is_refskip, bits >> 4 = cigar_ind */
} sta_plp_entry;
/* info->out_bytes = 16 * entries, n_lines = columns with >= 1 entry */
int sta_plp_plan(sta_engine *e, int32_t max_depth, int32_t overlaps, sta_plan_info *info);
int sta_plp_emit(sta_engine *e, void *dev_entries, uint64_t capacity);     /* NULL: engine buffer, read with sta_fetch_output */
/* reads whose pools the engine has cut out of raw BAM records on the device since it was created (sta_reads.raw_*): lets a caller
 * (and the tests) see that the device staging path, not the host copy, fed the windows */
uint64_t sta_stage_raw_reads(sta_engine *e);

/* ---- BGZF blocks inflated on the device (csrc/kernels_inflate.hip, one wave per block; the decoder the drivers' BAM reader feeds
 *      the compressed file to).  Replaces HTSlib bgzf.c inflate_block() under sam_read1 (bam_plcmd.c:409, bam2depth.c:541-543).
 *      This entry takes host buffers: block b's raw deflate data comp[comp_off, comp_off + clen) -> out[out_off, out_off + isize);
 *      status[b] = 0, or why the wave gave up on the block (damaged stream, table overflow, size mismatch: a caller then inflates
 *      that block with zlib, whose verdict counts).  CRC-32s are the caller's to check.  kernel_ms (may be NULL): the launch alone. ---- */
typedef struct sta_bgzf_block { uint64_t comp_off; uint32_t clen, isize; uint64_t out_off; } sta_bgzf_block;
int sta_bgzf_inflate(int32_t device, const uint8_t *comp, uint64_t comp_bytes, const sta_bgzf_block *blocks, int32_t n_blocks,
                     uint8_t *out, uint64_t out_bytes, uint32_t *status, double *kernel_ms);
/* first n (<= columns + 1) exclusive column offsets of the planned window (bytes for text plans, entries for sta_plp_plan) */
int sta_fetch_col_offsets(sta_engine *e, uint64_t *host_offs, uint64_t n);
/* per-read state after a plan: info words (bit 1 = read is in the pileup) and the working quality
 * pool (mate-overlap / BAQ adjusted), laid out like sta_reads.qual.  Either pointer may be NULL. */
int sta_fetch_read_state(sta_engine *e, int32_t file, uint32_t *host_info, uint8_t *host_qual);
/* Overlap detection where only the device knows who reaches bam_plp_push (-C: sam_cap_mapq reads the BAQ-adjusted qualities) or
 * who bam_plp_push turns away (a -d cap that triggers): a caller that keeps the overlap hash itself (sta_reads.olap_mate) cannot
 * fill olap_mate in before the plan.  With a resolver installed, sta_mpileup_plan / sta_plp_plan stop in front of the overlap pass,
 * hand the resolver every file's read states (bit 0 = reached bam_plp_push, bit 1 = in the pileup; a pushed read with a reference
 * span that is not in the pileup was turned away by the cap) and take olap_mate from it (mate_out[i] as in sta_reads.olap_mate; the
 * resolver returns 0, anything else fails the plan).  It is called on the thread that runs the plan; one host round trip per
 * window, so callers install it only for such windows.  fn == NULL removes it. */
typedef int (*sta_mate_resolver)(void *user, int32_t file, const uint32_t *read_state, int64_t n_reads, int32_t *mate_out);
int sta_set_mate_resolver(sta_engine *e, sta_mate_resolver fn, void *user);
/* mate-overlap visibility records of the last plan (one per read; fix_y[i] = -1: none): the query index whose quality a
 * deletion / ref-skip placeholder of read i shows, that quality before the pair was resolved, and the mate's read index.
 * HTSlib resolves a pair when the second mate is pushed, so columns handed out earlier still see the old value
 * (DESIGN.md section 2); the iterator surface uses these to present b->qual[] as of each column. */
int sta_fetch_overlap_fixups(sta_engine *e, int32_t file, int32_t *fix_y, int32_t *fix_mate, uint8_t *fix_q);

/* ---- coverage / bedcov: per-window column reductions (coverage.c:621-672, bedcov.c:316-333) ---- */
typedef struct sta_cov_params {
    int32_t mode;                /* 0 coverage, 1 bedcov */
    int32_t min_baseQ;           /* coverage -Q */
    int32_t min_depth;           /* coverage --min-depth (>= 1) / bedcov -d (-1 = off) */
    int32_t skip_dn;             /* bedcov -j */
    int32_t max_depth;           /* iterator depth cap (bam_mplp_set_maxcnt) */
    int32_t min_mq, rflag_require, rflag_filter, min_qlen;     /* read filters of the commands' read_bam callbacks */
    /* coverage -m / -D (coverage.c:632-668): hist_bins > 0 adds every column at or after hist_beg to bin
     * (pos - hist_beg) / hist_bin_width of the histogram opened with sta_cov_hist_begin -- one per column that counts
     * (breadth), or with hist_depth the column's filtered depth summed over the files */
    int32_t hist_bins, hist_depth;
    int64_t hist_beg, hist_bin_width;
} sta_cov_params;
typedef struct sta_cov_totals {  /* coverage: sums over the window's columns (and all files) */
    uint64_t n_covered_bases, summed_coverage, summed_baseQ, quality_bases, missing_qual;
} sta_cov_totals;
/* per_file: [n_files][2] = {sum of per-column depth, columns at or above min_depth} (bedcov); may be NULL for coverage.
 * info->n_kept_reads = reads that entered the pileup (bedcov -c).  Synchronises the stream. */
int sta_cov_plan(sta_engine *e, const sta_cov_params *p, sta_cov_totals *totals, uint64_t *per_file, sta_plan_info *info);
/* The histogram lives on the device across the windows of one contig: begin zeroes n_bins 32-bit counters (they wrap like the
 * reference's uint32_t array), fetch copies them out.  Synchronise the stream. */
int sta_cov_hist_begin(sta_engine *e, int32_t n_bins);
int sta_cov_hist_fetch(sta_engine *e, uint32_t *hist, int32_t n_bins);

/* ---- `samtools stats`, the coverage distribution (COV section): stats.c:311-391 (coverage_idx, the pileup round buffer),
 * :1452-1508 (the aligned blocks of a read go in), :1884-1892 (the lines).  The device does not keep a ring: an aligned block is a
 * pair of marks (+1 at its first position, -1 behind its last) and a batch of marks SORTED BY POSITION becomes the histogram
 * directly -- depth after mark i = carry_in + the sum of delta[0..i], it holds for pos[i+1] - pos[i] positions, and every
 * non-zero depth adds that many positions to bin coverage_idx(depth).  The last mark of a batch only closes the last run (its
 * delta is not applied: it opens the next batch).  The ring's aliasing (positions folding back modulo 5 x the longest read, the
 * slot that misses a whole-buffer flush, the short copy when the buffer grows) is applied by the caller to the marks; the
 * stats driver does so (driver_stats.cpp), DESIGN.md section 7 states the rules. ---- */
typedef struct sta_statcov_params { int32_t cov_min, cov_max, cov_step; } sta_statcov_params;   /* after stats.c:2398-2405 */
int sta_statcov_begin(sta_engine *e, const sta_statcov_params *p, int32_t *ncov);            /* zeroes 3 + (max-min)/step bins */
int sta_statcov_add(sta_engine *e, const int64_t *pos, const int32_t *delta, int64_t n, int64_t carry_in, int mem);
int sta_statcov_fetch(sta_engine *e, uint64_t *cov, int32_t ncov);

/* ---- per-column genotype-likelihood packer (bcf_call_init / bcf_call_glfgen, bam2bcf.c:38-48,65-123; errmod_cal in
 * HTSlib; the consensus call of bam_tview.c:194-212).  Columns come from the plain iterator (no filters, no overlaps), as
 * tview drives it through bam_lplbuf.  The reference FASTA set with sta_set_reference supplies the column's base ('N' if
 * none).  Columns with more than 255 counted bases are flagged: HTSlib subsamples them with its process-wide random stream,
 * the engine keeps the first 255 in pileup order. ---- */
typedef struct sta_glf_params {
    int32_t min_baseQ;           /* bcf_call_init(theta, min_baseQ): tview passes 13 */
    int32_t max_depth;           /* iterator depth cap (8000 = bam_plp default) */
    double theta;                /* <= 0: 0.83 (CALL_DEFTHETA); errmod_init(1 - theta) */
} sta_glf_params;
#define STA_GLF_CUT 1            /* sta_glf_col.flags: more than 255 counted bases */
typedef struct sta_glf_col {     /* what bcf_call_glfgen returns for one column of one file */
    int32_t n_plp;               /* entries the iterator handed over (0 = the iterator would not have returned the column) */
    int32_t n;                   /* return value: bases that passed the filters */
    int32_t flags;
    float qsum[4];               /* bcf_callret1_t (bam2bcf.h:41-46) */
    float p[25];
} sta_glf_col;
/* plans and runs the window: info->out_bytes = (col_end - col_beg) * n_files * sizeof(sta_glf_col), laid out [column][file];
 * read them back with sta_fetch_output */
int sta_glf_plan(sta_engine *e, const sta_glf_params *p, sta_plan_info *info);
/* bam_tview.c:194-212: consensus character (",ACMGRSVTWYHKDBN" alphabet) and its quality for one column */
int sta_glf_consensus(const sta_glf_col *c, char ref_base, char *call_char);

/* ---- calmd's per-record arithmetic: MD / NM recomputation (bam_fillmd1_core, bam_md.c:64-224) and the BAQ tag writer
 * (sam_prob_realn call, bam_md.c:474-479).  Works on file 0 of the staged window (records are independent: the window is only
 * the coordinate frame; the contig set with sta_set_reference is the FASTA calmd was given). ---- */
#define STA_CALMD_REALN     1    /* -r: run BAQ */
#define STA_CALMD_APPLY     2    /* -A: BAQ changes the qualities (tag ZQ:Z) instead of only writing BQ:Z */
#define STA_CALMD_EXTENDED  4    /* -E */
#define STA_CALMD_USE_EQUAL 8    /* -e: matching bases become '=' */
#define STA_CALMD_BIN_QUAL  16   /* -q */
typedef struct sta_calmd_params {
    int32_t flag; int32_t max_nm; /* -n, 0 = off */
    int32_t capQ;                 /* -C: > 10 = sam_cap_mapq's coefficient (bam_md.c:480-483): the caps are computed on the qualities the record
                                     carries after -r [-A], before -q / -n touch them; fetch with sta_fetch_calmd_mapq_cap */
} sta_calmd_params;
#define STA_CALMD_HAS_MD    1    /* state[]: NM / MD are valid (mapped record with a sequence on a contig that has a reference) */
#define STA_CALMD_NEW_TAG   2    /*          BAQ was computed: tag[] holds the BQ:Z (or, with -A, ZQ:Z) string               */
#define STA_CALMD_BQ_TO_ZQ  4    /*          an existing BQ:Z was applied to the qualities (realn.c renames it ZQ:Z)          */
#define STA_CALMD_ZQ_TO_BQ  8    /*          an existing ZQ:Z was taken back out of the qualities (renamed BQ:Z; STA_AUX_ZQ_RESTORE) */
/* info->out_bytes = bytes of MD text */
int sta_calmd_plan(sta_engine *e, const sta_calmd_params *p, sta_plan_info *info);
/* Results of the planned window (any pointer may be NULL): nm[n]; md_off[n+1] into md_text; state[n]; the quality, 4-bit
 * sequence and tag pools in the staged layout (sta_reads.base_off8; seq at half the byte offset). */
int sta_fetch_calmd(sta_engine *e, int32_t *nm, uint64_t *md_off, char *md_text, uint8_t *state, uint8_t *qual_pool,
                    uint8_t *seq_pool, uint8_t *tag_pool);
/* cap[n]: sam_cap_mapq of every read of the planned window (-1: the read's mismatch score is beyond the coefficient); only after a plan
 * with capQ > 10.  calmd then does `if (b->core.qual > q) b->core.qual = q` (a -1 lands in the unsigned field as 255). */
int sta_fetch_calmd_mapq_cap(sta_engine *e, int16_t *cap);

/* ---- consensus: `samtools consensus`' own column iterator (consensus_pileup.c:69-608: one column per
 * reference position plus one per inserted base, pads for the reads without the insertion) and its two callers, the
 * frequency caller (bam_consensus.c:1907-2014) and the Bayesian one (:1258-1880 with the per-read preparation :1012-1206).
 * Works on file 0 of the staged window.  The MD:Z text the Bayesian mode wants per read travels as text column 0 of
 * sta_reads.xcol_* (n_xcols >= 1; "*" or empty = no tag). ---- */
#define STA_CONS_SIMPLE    0     /* -m simple */
#define STA_CONS_BAYES_116 1     /* -m bayesian_116 */
#define STA_CONS_RECALL    2     /* -m bayesian (default) */
#define STA_CONS_PRECISE   3     /* -m bayesian_p */
#define STA_CONS_MIXED     4     /* -m bayesian_m */
typedef struct sta_cons_params {    /* consensus_opts (bam_consensus.c:211-260), same names */
    int32_t mode, use_qual, min_qual, adj_qual, use_mqual, nm_adjust, nm_halo, sc_cost, low_mqual, high_mqual, min_depth;
    int32_t cons_cutoff, ambig, default_qual, excl_flags, incl_flags, min_mqual;
    int32_t want_pileup;         /* also produce the per-column base / quality characters of `-f pileup` */
    double scale_mqual, call_fract, het_fract, P_het, P_indel, het_scale, homopoly_fix, homopoly_redux;
    int32_t qcal[3][101];        /* qcal_t smap / umap / omap (bam_consensus.c:191-195); identity = :flat */
} sta_cons_params;
typedef struct sta_cons_col {    /* one (position, nth) column */
    int32_t depth;               /* reads in the column (0: the iterator has no such column) */
    int32_t base;                /* consensus_base(): call character */
    int32_t qual;                /*                   and its quality */
} sta_cons_col;
typedef struct sta_cons_info {
    uint64_t n_cols;             /* (position, nth) columns of [col_beg, col_end), in order: position p has 1 + ins[p] of them */
    uint64_t n_entries;          /* sum of depth over the columns */
    uint64_t n_kept_reads;
} sta_cons_info;
/* runs the staged window.  Synchronises the stream. */
int sta_consensus_run(sta_engine *e, const sta_cons_params *p, sta_cons_info *info);
/* results of the last sta_consensus_run (any pointer may be NULL): ins[col_end - col_beg] = inserted columns after each
 * position; cols[n_cols]; with want_pileup col_off[n_cols + 1] into seq_chars / qual_chars[n_entries] (pileup_t.base as
 * `-f pileup` prints it: reverse-strand bases lower-cased, '#' for a reverse-strand pad; min(qual, 93) + '!'). */
int sta_fetch_consensus(sta_engine *e, int32_t *ins, sta_cons_col *cols, uint64_t *col_off, char *seq_chars, char *qual_chars);
/* The iterator half alone -- what pileup_loop() / get_next_base() hand to seq_column (consensus_pileup.c:69-608) -- for callers
 * that keep their own per-column code (samtools_amd_cons.h builds pileup_loop on it).  No read filters besides the unmapped
 * flag.  info->n_entries = entry words.  Per read i of the staged file: its first / last column index in the window
 * (last < first: not in it) and entry_off[i]; its entry for column c is entries[entry_off[i] + c - first_col[i]]:
 * bits 0-4 4-bit base code (16 = pad / deletion), 5-12 quality, 0x2000 ref_skip, 0x4000 reverse strand, 0x8000 the column
 * lies in a reference skip (base '.'), 0x10000 padding; seq_offs[] = pileup_t.seq_offset of the same entry. */
int sta_cons_entries_run(sta_engine *e, sta_cons_info *info);
int sta_fetch_cons_entries(sta_engine *e, int32_t *ins, int32_t *first_col, int32_t *last_col, uint64_t *entry_off, uint32_t *entries, uint32_t *seq_offs);
/* ---- depth ---- */
int sta_depth_plan(sta_engine *e, const sta_depth_params *p, sta_plan_info *info);
int sta_depth_emit(sta_engine *e, void *dev_out, uint64_t capacity);
/* plan + emit in one call (same contract as sta_mpileup_run) */
int sta_depth_run(sta_engine *e, const sta_depth_params *p, void *dev_out, uint64_t capacity, sta_plan_info *info);
/* binary per-column counts of the planned window: int32 [n_files][col_end-col_beg]
 * (device pointer owned by the engine, valid until the next plan call) */
const int32_t *sta_depth_counts_dev(sta_engine *e);

/* ---- results ---- */
int sta_fetch_output(sta_engine *e, char *host_out, uint64_t n);   /* D2H + sync */
/* n bytes of the output from byte `offset` on: a caller that writes the text out piece by piece through a small page-locked buffer
 * (the drivers' text ring, csrc/driver_pipeline.h) instead of holding a window's whole text on the host.  D2H + sync. */
int sta_fetch_output_at(sta_engine *e, char *host_out, uint64_t offset, uint64_t n);
int sta_sync(sta_engine *e);

/* ---- measurement support (bench.py) ---- */
/* When enabled, every kernel launch is bracketed by HIP events on the engine
 * stream and accumulated per kernel name. */
void sta_profile_enable(sta_engine *e, int on);
/* Only the launches accumulated under `name` are bracketed (NULL or "": all of them again).  An event pair costs the stream
 * 3-6 us, two dozen pairs a step were 0.07 ms of bench.py's timed region: the timed steps time the roofline's kernel only,
 * the per-kernel table comes from a pass of its own. */
void sta_profile_only(sta_engine *e, const char *name);
void sta_profile_reset(sta_engine *e);
/* Copies up to cap entries; returns the number of distinct kernels. */
typedef struct sta_kernel_time { char name[48]; uint64_t launches; double total_ms; } sta_kernel_time;
int sta_profile_get(sta_engine *e, sta_kernel_time *out, int cap);

/* ---- samtools-compatible command drivers (host side, C++) ---- */
/* argv[0] = "mpileup" / "depth"; same options, stdout/stderr text and exit
 * status as the reference sub-commands (bamtk.c:248,270 would dispatch here). */
int sta_main_mpileup(int argc, char **argv);
int sta_main_depth(int argc, char **argv);
/* A program that ends with the driver's exit status (csrc/main.cpp) may ask the drivers to end the PROCESS as soon as their output is
 * written and flushed -- _exit(status) -- instead of returning through the teardown of page-locked pools, engine and HIP runtime
 * (~0.28 s of a 0.7 s run).  Never set by an in-process caller. */
void sta_exit_after_main(int on);
/* The same two drivers with the text they would print handed back in memory (argv[0] selects "mpileup" / "depth"; a -o option
 * in argv still wins).  *text is malloc'ed: release it with sta_capture_free.  Returns the driver's exit status, or
 * STA_ERR_ARG / STA_ERR_IO.  Used by the sharded launcher (samtools_amd/shard.py): a rank's block of columns goes from the
 * driver straight into the gather, no temporary file.  No reference counterpart (bam_plcmd.c:663-868 prints as it goes). */
int sta_main_capture(int argc, char **argv, char **text, uint64_t *n_bytes);
void sta_capture_free(char *text);
/* The same with the windows' text left in device memory (a sharded run gathers it GPU to GPU: samtools_amd/shard.py).  What the driver
 * itself writes (depth -H's header line) comes back as host text and goes first in the output; the windows' n_dev_bytes are kept until
 * sta_capture_device_take() copies them (device to device) to dev_dst and releases them.  One capture at a time per process; needs
 * STA_DEV_THREADS=1 (the default).  host_text: sta_capture_free(). */
int sta_main_capture_device(int argc, char **argv, uint64_t *n_dev_bytes, char **host_text, uint64_t *n_host_bytes);
int sta_capture_device_take(void *dev_dst, uint64_t capacity);
/* `glf [-Q min_baseQ] [-t theta] [-f ref.fa] in.bam`: one text line per column (what tests diff against the oracle) */
int sta_main_glf(int argc, char **argv);
/* `calmd [-erAEqdNQ] [-n max_nm] [--no-PG] in.bam ref.fa`: bam_fillmd (bam_md.c:346-520); the records are written as SAM text
 * with the header, or as BAM with -b / -u (-C is refused) */
int sta_main_calmd(int argc, char **argv);
/* `stats [-c min,max,step] [-f INT] [-F INT] [-d] [-l INT] [-I ID] [-t targets] in.bam [region ...]`: the coverage distribution of `samtools stats`
 * (the comment line and the COV lines of stats.c:1884-1892; nothing else of that report).  --marks-out FILE writes the sorted
 * marks the device would be given ("epoch<TAB>position<TAB>delta" lines) instead, without touching a device. */
int sta_main_stats(int argc, char **argv);
/* `consensus [options] in.bam`: options, text and exit status of `samtools consensus` (bam_consensus.c:3082-3593; -X presets
 * and the named calibration tables other than :flat are not built) */
int sta_main_consensus(int argc, char **argv);

/* ---- host input plumbing (needs no device) ----
 * The drivers' SAM / BAM reader (stands where sam_open / sam_read1 stand for bam_plcmd.c:500-569): BGZF blocks are inflated
 * by `threads` workers and records are parsed one batch ahead of the consumer.  sta_io_scan decodes a whole file through it
 * (`path` may hold several paths separated by '\n': the drivers' multi-file windows)
 * and returns the record count and an order-dependent checksum over every decoded field (threads <= 0: the drivers'
 * default, $STA_IO_THREADS or 4..8 depending on the machine; 1 worker is still a separate thread).  stage != 0 additionally pushes the records through
 * the drivers' window pump and SoA stager (what runs between the reader and sta_stage_window) and checksums every staged
 * window: 1 = one decoded record at a time (host_pump.h), 2 = chunk slices decoded on `threads` parser threads
 * (host_chunk.h); both lanes must give the same checksum.  Returns 0, or <0 on error. */
int sta_io_scan(const char *path, int threads, int stage, uint64_t *n_records, uint64_t *checksum);
/* The records of one region ("chr", "chr:beg-end") of one file as a `-r` run reads them (stands where sam_itr_querys stands at
 * bam_plcmd.c:550 / bam2depth.c:961-975): with use_index != 0 and a .bai beside the BAM the reader starts at the linear
 * index's virtual offset for the region start and stops at the first record beyond the region; otherwise it filters the whole
 * file.  Same record count and checksum either way; *used_index = 1 when the index was used.  Host only. */
int sta_io_scan_region(const char *path, const char *region, int threads, int use_index, uint64_t *n_records, uint64_t *checksum, int *used_index);
/* Text of an 'f' / 'd' aux value as `mpileup --output-extra TAG` prints it: HTSlib's kputd (bam_plcmd.c:838-840), which is not
 * printf("%g") -- six significant digits, half rounded UP on the truncated decimal expansion inside [0.0001, 999999], "%g"
 * outside.  Host only.  Returns the length written (NUL-terminated), or -1 when cap is too small. */
int sta_format_aux_float(double v, char *buf, int cap);
/* Every record of `path` read with the drivers' reader and written back to out_path as SAM text behind the header, the way
 * sam_write1 writes a record calmd did not change (bam_md.c:486-489): integer aux fields of every width as `i`, floats through
 * kputd, B arrays comma separated.  Host only; returns 0 or <0. */
int sta_io_write_sam(const char *path, const char *out_path);
/* The same as BAM (SAM spec 4.1 / 4.2; what calmd -b / -u write): level 0 = stored BGZF blocks, otherwise compressed. */
int sta_io_write_bam(const char *path, const char *out_path, int level);
/* Every contig of a reference FASTA through the drivers' loader (stands where fai_load / faidx_fetch_seq64 stand at
 * bam_plcmd.c:289-352): with a .fai beside a plain file only the index is read up front and a contig's bases are read when first
 * asked for (the next contig of the file on a thread of its own meanwhile); otherwise, or with STA_FASTA_WHOLE=1, or when the file does
 * not fit its index, the whole file is parsed.  Count, total bases, a checksum over names and bases in file order (order != 0: fetched
 * from the last contig to the first), *lazy = 1 when the index was used.  Host only. */
int sta_io_fasta_scan(const char *path, int order, uint64_t *n_contigs, uint64_t *n_bases, uint64_t *checksum, int *lazy);

#ifdef __cplusplus
}
#endif
#endif
