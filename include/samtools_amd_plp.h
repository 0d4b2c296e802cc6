/*
 * samtools_amd_plp.h -- the per-column pileup callback surface of the MI355X engine.
 *
 * C-ABI replacement for the HTSlib pileup iterator that samtools drives one column at a time
 * (SURVEY.md section 8b).  Function names, argument meaning, ownership and error behaviour are
 * HTSlib's (htslib/sam.h, release 1.23.1 -- not in the reference tree; the call sites are):
 *
 *   reference call site (file:line)                              entry points used there
 *   -----------------------------------------------------------  ----------------------------------------
 *   bam_plcmd.c:581-607,922  (mpileup)                           bam_mplp_init, bam_mplp_constructor/destructor,
 *                                                                bam_mplp_init_overlaps, bam_mplp_set_maxcnt,
 *                                                                bam_mplp64_auto, bam_mplp_destroy
 *   bam_plbuf.c:40-69        (bam_plbuf_* wrapper)               bam_plp_init, bam_plp_push, bam_plp64_next,
 *                                                                bam_plp_reset, bam_plp_destroy
 *   bedcov.c:303-335, coverage.c:572-589, cut_target.c:223-248   bam_mplp_init/_auto, bam_plp_init/_auto, set_maxcnt
 *   bam_plcmd.c:119, bam_tview.c:223,255                         bam_plp_insertion_mod, bam_plp_insertion
 *   bam_plbuf.h:45-51                                            bam_plbuf_init/_push/_reset/_destroy
 *
 * Every symbol is exported with an `sta_` prefix so that the library can be loaded next to a real
 * libhts; define STA_PLP_DROPIN before including this header to get the unprefixed HTSlib names as
 * macros, which is all a samtools source file needs to switch engines.
 *
 * How it works: the iterator reads AHEAD through the caller's bam_plp_auto_f callback (legal: the
 * callback is the only producer), copies the records (bam_copy1 semantics: the iterator owns the
 * copies), stages a window of them into HBM, lets the device resolve every (read, column) pair
 * (sta_plp_plan / sta_plp_emit in samtools_amd.h) and then hands out bam_pileup1_t arrays column by
 * column.  Arrays and the bam1_t they point to are valid until the next *_auto/_next call, as in
 * HTSlib.  With overlaps enabled, b->qual[] of a returned entry carries the mate-overlap adjusted
 * qualities for the columns of the current window.  There is no CPU fallback: without a HIP device
 * the iterators return NULL / -1.
 */
#ifndef SAMTOOLS_AMD_PLP_H
#define SAMTOOLS_AMD_PLP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- record and pileup types: HTSlib's, declared here only when htslib/sam.h is not in use ---- */
#ifndef HTSLIB_SAM_H
typedef int64_t hts_pos_t;

typedef struct bam1_core_t {
    hts_pos_t pos;
    int32_t tid;
    uint16_t bin;
    uint8_t qual;
    uint8_t l_extranul;
    uint16_t flag;
    uint16_t l_qname;
    uint32_t n_cigar;
    int32_t l_qseq;
    int32_t mtid;
    hts_pos_t mpos;
    hts_pos_t isize;
} bam1_core_t;

typedef struct bam1_t {
    bam1_core_t core;
    uint64_t id;
    uint8_t *data;          /* qname | cigar | seq (4-bit) | qual | aux */
    int l_data;
    uint32_t m_data;
    uint32_t mempolicy:2, :30;
} bam1_t;

typedef union { void *p; int64_t i; double f; } bam_pileup_cd;

typedef struct bam_pileup1_t {
    bam1_t *b;
    int32_t qpos;
    int indel, level;
    uint32_t is_del:1, is_head:1, is_tail:1, is_refskip:1, :1, aux:27;
    bam_pileup_cd cd;
    int cigar_ind;
} bam_pileup1_t;

typedef struct kstring_t { size_t l, m; char *s; } kstring_t;
typedef struct hts_base_mod_state hts_base_mod_state;      /* the MM / ML evaluation of one read: opaque, as in HTSlib */
typedef struct hts_base_mod {                               /* htslib/sam.h: one modification of one base */
    int modified_base;      /* the code letter ('m', 'h', ...), or the ChEBI number negated */
    int canonical_base;     /* the MM entry's base letter ('C', 'N', ...) */
    int strand;             /* 0 '+', 1 '-' */
    int qual;               /* the ML probability 0..255, -1 without an ML value */
} hts_base_mod;

#define bam_get_qname(b) ((char *)(b)->data)
#define bam_get_cigar(b) ((uint32_t *)((b)->data + (b)->core.l_qname))
#define bam_get_seq(b) ((b)->data + ((b)->core.n_cigar << 2) + (b)->core.l_qname)
#define bam_get_qual(b) ((b)->data + ((b)->core.n_cigar << 2) + (b)->core.l_qname + (((b)->core.l_qseq + 1) >> 1))
#define bam_seqi(s, i) ((s)[(i) >> 1] >> ((~(i) & 1) << 2) & 0xf)
#endif /* HTSLIB_SAM_H */

typedef int (*sta_bam_plp_auto_f)(void *data, bam1_t *b);     /* >=0 ok, -1 EOF, < -1 error; fills caller-owned b */
typedef struct sta_bam_plp *sta_bam_plp_t;
typedef struct sta_bam_mplp *sta_bam_mplp_t;

/* ---- single-file iterator (HTSlib bam_plp_*) ---- */
sta_bam_plp_t sta_bam_plp_init(sta_bam_plp_auto_f func, void *data);
void sta_bam_plp_destroy(sta_bam_plp_t iter);
/* b == NULL marks end of input.  <0 on error (unsorted input, allocation). */
int sta_bam_plp_push(sta_bam_plp_t iter, const bam1_t *b);
/* NULL with *n_plp == 0: more input needed / end; NULL with *n_plp == -1: error */
const bam_pileup1_t *sta_bam_plp_next(sta_bam_plp_t iter, int *tid, int *pos, int *n_plp);
const bam_pileup1_t *sta_bam_plp64_next(sta_bam_plp_t iter, int *tid, hts_pos_t *pos, int *n_plp);
const bam_pileup1_t *sta_bam_plp_auto(sta_bam_plp_t iter, int *tid, int *pos, int *n_plp);
const bam_pileup1_t *sta_bam_plp64_auto(sta_bam_plp_t iter, int *tid, hts_pos_t *pos, int *n_plp);
void sta_bam_plp_set_maxcnt(sta_bam_plp_t iter, int maxcnt);
void sta_bam_plp_reset(sta_bam_plp_t iter);
int sta_bam_plp_init_overlaps(sta_bam_plp_t iter);
void sta_bam_plp_constructor(sta_bam_plp_t iter, int (*func)(void *data, const bam1_t *b, bam_pileup_cd *cd));
void sta_bam_plp_destructor(sta_bam_plp_t iter, int (*func)(void *data, const bam1_t *b, bam_pileup_cd *cd));
/* inserted sequence after p (pads as '*'), returns its length incl. pads or <0; *del_len = deletion that follows it */
int sta_bam_plp_insertion(const bam_pileup1_t *p, kstring_t *ins, int *del_len);
/* the form bam_plcmd.c:119 calls: m == NULL (no --output-mods) is bam_plp_insertion; with a state (below) every inserted base is followed
 * by the "[...]" text of its modifications; a state that bam_parse_basemod never filled is refused (< 0, message on stderr) */
int sta_bam_plp_insertion_mod(const bam_pileup1_t *p, hts_base_mod_state *m, kstring_t *ins, int *del_len);

/* ---- base modifications: the four HTSlib calls of bam_plcmd.c:86-109 and :356-369 (sam_mods.c is not in the reference tree: this
 * library's own MM / ML evaluation, SAM tags specification 1.7, the one `samtools-amd mpileup --output-mods` prints from) ---- */
hts_base_mod_state *sta_hts_base_mod_state_alloc(void);
void sta_hts_base_mod_state_free(hts_base_mod_state *state);
/* evaluates the record's MM:Z / ML:B:C aux fields; 0 on success (also without an MM tag), -1 on a malformed tag */
int sta_bam_parse_basemod(const bam1_t *b, hts_base_mod_state *state);
/* the modifications of query position qpos in MM order: fills up to n_mods entries, returns their number (0: none, < 0: error) */
int sta_bam_mods_at_qpos(const bam1_t *b, int qpos, hts_base_mod_state *state, hts_base_mod *mods, int n_mods);

/* ---- multi-file iterator (HTSlib bam_mplp_*) ---- */
sta_bam_mplp_t sta_bam_mplp_init(int n, sta_bam_plp_auto_f func, void **data);
void sta_bam_mplp_destroy(sta_bam_mplp_t iter);
void sta_bam_mplp_set_maxcnt(sta_bam_mplp_t iter, int maxcnt);
int sta_bam_mplp_init_overlaps(sta_bam_mplp_t iter);
void sta_bam_mplp_reset(sta_bam_mplp_t iter);
void sta_bam_mplp_constructor(sta_bam_mplp_t iter, int (*func)(void *data, const bam1_t *b, bam_pileup_cd *cd));
void sta_bam_mplp_destructor(sta_bam_mplp_t iter, int (*func)(void *data, const bam1_t *b, bam_pileup_cd *cd));
/* >0: number of files with data at (*tid,*pos); 0: end; <0: error */
int sta_bam_mplp_auto(sta_bam_mplp_t iter, int *tid, int *pos, int *n_plp, const bam_pileup1_t **plp);
int sta_bam_mplp64_auto(sta_bam_mplp_t iter, int *tid, hts_pos_t *pos, int *n_plp, const bam_pileup1_t **plp);

/* ---- samtools' own push-style wrapper (bam_plbuf.h:30-51) ---- */
typedef int (*sta_bam_pileup_f)(uint32_t tid, hts_pos_t pos, int n, const bam_pileup1_t *pl, void *data);
typedef struct sta_bam_plbuf sta_bam_plbuf_t;
sta_bam_plbuf_t *sta_bam_plbuf_init(sta_bam_pileup_f func, void *data);
void sta_bam_plbuf_destroy(sta_bam_plbuf_t *buf);
void sta_bam_plbuf_reset(sta_bam_plbuf_t *buf);
int sta_bam_plbuf_push(const bam1_t *b, sta_bam_plbuf_t *buf);

/* ---- tuning (not in HTSlib) ---- */
/* records pulled per device window (default 65536, env STA_PLP_BATCH) */
void sta_bam_plp_set_batch(sta_bam_plp_t iter, int n_records);

#ifdef STA_PLP_DROPIN
#define bam_plp_auto_f sta_bam_plp_auto_f
#define bam_plp_t sta_bam_plp_t
#define bam_mplp_t sta_bam_mplp_t
#define bam_plp_init sta_bam_plp_init
#define bam_plp_destroy sta_bam_plp_destroy
#define bam_plp_push sta_bam_plp_push
#define bam_plp_next sta_bam_plp_next
#define bam_plp64_next sta_bam_plp64_next
#define bam_plp_auto sta_bam_plp_auto
#define bam_plp64_auto sta_bam_plp64_auto
#define bam_plp_set_maxcnt sta_bam_plp_set_maxcnt
#define bam_plp_reset sta_bam_plp_reset
#define bam_plp_init_overlaps sta_bam_plp_init_overlaps
#define bam_plp_constructor sta_bam_plp_constructor
#define bam_plp_destructor sta_bam_plp_destructor
#define bam_plp_insertion sta_bam_plp_insertion
#define bam_plp_insertion_mod sta_bam_plp_insertion_mod
#ifndef STA_PLP_KEEP_HTSLIB_MODS      /* (with HTSlib linked as well: define this to keep its own sam_mods.c calls and state type) */
#define hts_base_mod_state_alloc sta_hts_base_mod_state_alloc
#define hts_base_mod_state_free sta_hts_base_mod_state_free
#define bam_parse_basemod sta_bam_parse_basemod
#define bam_mods_at_qpos sta_bam_mods_at_qpos
#endif
#define bam_mplp_init sta_bam_mplp_init
#define bam_mplp_destroy sta_bam_mplp_destroy
#define bam_mplp_set_maxcnt sta_bam_mplp_set_maxcnt
#define bam_mplp_init_overlaps sta_bam_mplp_init_overlaps
#define bam_mplp_reset sta_bam_mplp_reset
#define bam_mplp_constructor sta_bam_mplp_constructor
#define bam_mplp_destructor sta_bam_mplp_destructor
#define bam_mplp_auto sta_bam_mplp_auto
#define bam_mplp64_auto sta_bam_mplp64_auto
#define bam_pileup_f sta_bam_pileup_f
#define bam_plbuf_t sta_bam_plbuf_t
#define bam_plbuf_init sta_bam_plbuf_init
#define bam_plbuf_destroy sta_bam_plbuf_destroy
#define bam_plbuf_reset sta_bam_plbuf_reset
#define bam_plbuf_push sta_bam_plbuf_push
#endif

#ifdef __cplusplus
}
#endif
#endif
