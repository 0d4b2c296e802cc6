/*
 * oracle/o_plp.h -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Restatement of the HTSlib 1.23.1 pileup iterator surface used by samtools
 * (bam_plp_* / bam_mplp_*; callers bam_plcmd.c:581-607, bam_plbuf.c:40-66).
 * HTSlib's sam.c is absent from /root/reference; semantics follow
 * SURVEY.md Appendix A.1-A.3 and are pinned by test/mpileup/expected goldens.
 */
#ifndef O_PLP_H
#define O_PLP_H
#include "o_common.h"

typedef int (*oplp_auto_f)(void *data, orec_t *b);   /* >=0 ok, -1 EOF, < -1 error */

typedef struct {
    orec_t *b;
    int32_t qpos;
    int indel, level;
    uint32_t is_del:1, is_head:1, is_tail:1, is_refskip:1, aux:28;
    int cigar_ind;
} opileup1_t;

typedef struct oplp oplp_t;
typedef struct omplp omplp_t;

oplp_t *oplp_init(oplp_auto_f func, void *data);
void oplp_destroy(oplp_t *iter);
int oplp_push(oplp_t *iter, const orec_t *b);
const opileup1_t *oplp_next(oplp_t *iter, int *tid, hpos_t *pos, int *n_plp);
const opileup1_t *oplp_auto(oplp_t *iter, int *tid, hpos_t *pos, int *n_plp);
void oplp_set_maxcnt(oplp_t *iter, int maxcnt);
int oplp_init_overlaps(oplp_t *iter);

omplp_t *omplp_init(int n, oplp_auto_f func, void **data);
void omplp_destroy(omplp_t *iter);
void omplp_set_maxcnt(omplp_t *iter, int maxcnt);
int omplp_init_overlaps(omplp_t *iter);
/* bam_plp_constructor / bam_mplp_constructor: called when a read enters the pileup (cd is passed as NULL) */
void oplp_constructor(oplp_t *iter, int (*func)(void *data, const orec_t *b, void *cd));
void omplp_constructor(omplp_t *iter, int (*func)(void *data, const orec_t *b, void *cd));
int omplp_auto(omplp_t *iter, int *tid, hpos_t *pos, int *n_plp, const opileup1_t **plp);

/* bam_plp_insertion: inserted sequence following p (pads as '*'); returns
 * length incl. pads, *del_len = length of a deletion following the insertion */
int oplp_insertion(const opileup1_t *p, ostr_t *ins, int *del_len);

/* realn.c / probaln.c restatement (o_baq.c) */
int o_prob_realn(orec_t *b, const char *ref, hpos_t ref_len, int flag);

/* ---- base modifications (o_mods.c): MM / ML tags of one record, grouped by query position ---- */
typedef struct { int code; int strand; int qual; } omod1_t;     /* code: letter, or -ChEBI number; qual -1 = no ML value */
typedef struct { int n, l; int *start; omod1_t *ent; } omods_t; /* start[q] .. start[q+1]: entries of query position q */
int omods_parse(const orec_t *b, omods_t *m);
void omods_free(omods_t *m);
void omods_put(const omods_t *m, int qpos, ostr_t *out);
/* bam_plp_insertion_mod: the inserted sequence with the modifications of every inserted base (m may be NULL) */
int oplp_insertion_mod(const opileup1_t *p, const omods_t *m, ostr_t *ins, int *del_len);

#endif
