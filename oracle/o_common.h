/*
 * oracle/o_common.h -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Plain-C restatement of the samtools mpileup/depth hot path, used only by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the
 * checker.  Nothing under samtools_amd/ may include, link or execute it.
 *
 * This header: record / header / reader / FASTA / BED / region helpers that
 * stand in for the HTSlib 1.23.1 I/O layer (absent from /root/reference; see
 * SURVEY.md section 1 fact 1).  Record layout follows the BAM spec / bam1_t
 * (SURVEY.md section 8b).
 */
#ifndef O_COMMON_H
#define O_COMMON_H

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef int64_t hpos_t;
#define HPOS_MAX ((((int64_t)INT32_MAX)<<32)|UINT32_MAX)

/* BAM flag bits */
#define F_PAIRED 1
#define F_PROPER_PAIR 2
#define F_UNMAP 4
#define F_MUNMAP 8
#define F_REVERSE 16
#define F_READ1 64
#define F_READ2 128
#define F_SECONDARY 256
#define F_QCFAIL 512
#define F_DUP 1024
#define F_SUPPLEMENTARY 2048

/* CIGAR ops: MIDNSHP=XB */
enum { C_M = 0, C_I, C_D, C_N, C_S, C_H, C_P, C_EQ, C_X, C_B };
#define cig_op(c) ((c) & 0xf)
#define cig_len(c) ((c) >> 4)

/* ---- growable string (kstring stand-in) ---- */
typedef struct { size_t l, m; char *s; } ostr_t;
void os_reserve(ostr_t *s, size_t extra);
static inline void os_putc(ostr_t *s, int c) { if (s->l + 2 > s->m) os_reserve(s, 2); s->s[s->l++] = (char)c; s->s[s->l] = 0; }
void os_putsn(ostr_t *s, const char *p, size_t n);
static inline void os_puts(ostr_t *s, const char *p) { os_putsn(s, p, strlen(p)); }
void os_putll(ostr_t *s, long long v);      /* kputll / kputw / kputuw */
static inline void os_clear(ostr_t *s) { s->l = 0; if (s->s) s->s[0] = 0; }

/* ---- alignment record (bam1_t stand-in) ---- */
typedef struct {
    hpos_t pos, mpos, isize;
    int32_t tid, mtid;
    uint16_t flag;
    uint8_t mapq;
    int32_t l_qseq;
    uint32_t n_cigar;
    char *qname;      /* NUL terminated */
    uint32_t *cigar;  /* len<<4|op */
    uint8_t *seq;     /* 4-bit packed, high nibble first */
    uint8_t *qual;    /* l_qseq bytes; 0xff = absent */
    uint8_t *aux;     /* BAM-encoded aux block */
    int l_aux;
    /* backing store */
    uint8_t *data; size_t m_data;
} orec_t;

#define rec_seqi(s, i) ((s)[(i) >> 1] >> ((~(i) & 1) << 2) & 0xf)
#define rec_is_rev(b) (((b)->flag & F_REVERSE) != 0)

void rec_free(orec_t *r);
int rec_copy(orec_t *dst, const orec_t *src);   /* bam_copy1 */
hpos_t rec_rlen(const orec_t *r);               /* bam_cigar2rlen */
hpos_t rec_endpos(const orec_t *r);             /* bam_endpos: pos + max(rlen,1) */
const uint8_t *rec_aux_get(const orec_t *r, const char tag[2]); /* -> type byte */
const uint8_t *rec_aux_next(const uint8_t *p, const uint8_t *end);   /* tag start -> next tag start */
int rec_aux_del(orec_t *r, const uint8_t *v);                    /* bam_aux_del */
void o_put_double(ostr_t *s, double d);                          /* HTSlib kputd (o_mpileup.c) */

/* ---- header ---- */
typedef struct {
    int n_ref;
    char **name;
    hpos_t *len;
    char *text;   /* raw header text */
} ohdr_t;
int hdr_name2tid(const ohdr_t *h, const char *name);
void hdr_free(ohdr_t *h);

/* ---- reader (SAM text, gz SAM, BAM) ---- */
typedef struct oreader oreader_t;
oreader_t *rd_open(const char *fn);            /* "-" = stdin */
ohdr_t *rd_header(oreader_t *r);               /* owned by reader */
/* region restriction, emulating sam_itr_querys(): only records on tid whose
 * [pos, endpos) overlaps [beg,end) are returned. */
void rd_set_region(oreader_t *r, int tid, hpos_t beg, hpos_t end);
int rd_next(oreader_t *r, orec_t *rec);        /* 0 ok, -1 EOF, < -1 error */
void rd_close(oreader_t *r);

/* hts_parse_reg-like: "chr", "chr:beg", "chr:beg-end" with thousands commas.
 * Returns 0 and fills tid/beg/end (0-based half open) or -1. */
int parse_region(const ohdr_t *h, const char *reg, int *tid, hpos_t *beg, hpos_t *end);

/* ---- FASTA (faidx stand-in: whole file in memory) ---- */
typedef struct {
    int n;
    char **name;
    char **seq;
    hpos_t *len;
} ofasta_t;
ofasta_t *fa_load(const char *fn);
const char *fa_fetch(const ofasta_t *fa, const char *name, hpos_t *len);
void fa_free(ofasta_t *fa);

/* ---- BED (bedidx.c:102-197,258-364) ---- */
typedef struct obed obed_t;
obed_t *bed_load(const char *fn);
int bed_olap(const obed_t *b, const char *chr, hpos_t beg, hpos_t end);
void bed_free(obed_t *b);

/* ---- small tables (hts.c) ---- */
extern const char nt16_str[];          /* "=ACMGRSVTWYHKDBN" */
extern const unsigned char nt16_table[256];
extern const int nt16_int[16];

int str2flag(const char *s);           /* bam_str2flag */

/* o_mpileup.c: HTSlib realn.c sam_cap_mapq */
int o_cap_mapq(orec_t *b, const char *ref, hpos_t ref_len, int thres);

#endif
