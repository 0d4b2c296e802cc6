/*
 * samtools_amd_cons.h -- `samtools consensus`' column iterator, pileup_loop(), on the MI355X engine.
 *
 * C-ABI replacement for consensus_pileup.h:78-107 (implementation consensus_pileup.c:69-608): same arguments, same callback
 * protocol, same pileup_t layout (consensus_pileup.h:41-75), so that bam_consensus.c:2961-2972 (and any other client of that
 * header) links against it unchanged:
 *
 *     #define STA_CONS_DROPIN
 *     #include "samtools_amd_cons.h"          // pileup_t -> sta_pileup_t, pileup_loop -> sta_pileup_loop
 *     ...
 *     pileup_loop(fp, h, readaln2, nm_init, basic_fasta, nm_free, &ctx);
 *
 * How it works: records are pulled through seq_fetch a batch ahead (the callback is the only producer), seq_init runs on
 * every record as it is pulled, the batch is staged to HBM and the device resolves every (read, column) pair including the
 * insertion columns (sta_cons_entries_run in samtools_amd.h); seq_column is then called column by column with the linked
 * list of pileup_t in the reference's order, and seq_free when a read leaves.  Per column the fields the reference's callbacks
 * read are filled -- next, cd, b, b_is_rev, b_qual / b_seq / b_cigar, base, base4, qual, ref_skip, padding, pos, nth,
 * seq_offset -- while the cursor internals (start, eof, first_del, cigar_ind / cigar_op / cigar_len, eofn / eofl) are left 0.
 * Differences: seq_init of a batch runs before the seq_column calls of the columns that precede those reads (the reference
 * interleaves them); is_insert is the number of inserted columns still to come at the position, which is what the reference
 * passes when insertions are single CIGAR operations; a return of 1 from seq_column stops the loop as in the reference.
 * There is no CPU fallback: without a HIP device the function prints why and returns -1.
 */
#ifndef SAMTOOLS_AMD_CONS_H
#define SAMTOOLS_AMD_CONS_H
#include "samtools_amd_plp.h"
#ifdef __cplusplus
extern "C" {
#endif
#ifndef HTSLIB_SAM_H
typedef struct htsFile samFile;          /* opaque: only handed back to the callbacks */
typedef struct sam_hdr_t sam_hdr_t;
#endif

typedef struct sta_pileup {
    struct sta_pileup *next;  /* a link list, for active seqs */
    void *cd;                 /* per-sequence client data (seq_init / seq_free) */
    int eof;
    int qual;                 /* current quality */
    char start;
    char base;                /* current base, ASCII ('*' pad / deletion, '.' reference skip) */
    char ref_skip;
    char padding;             /* the base was added because of another sequence's insertion */
    int base4;                /* 4-bit code, 16 for '*' */
    hts_pos_t pos;            /* current unpadded position (1-based) */
    int nth;                  /* nth base at that position */
    int b_is_rev;
    int seq_offset;           /* current index into the read's bases */
    unsigned char *b_qual;    /* cached bam_get_qual / bam_get_seq */
    unsigned char *b_seq;
    struct sta_pileup *eofn, *eofl;
    uint32_t *b_cigar;
    int cigar_ind, cigar_op, cigar_len;
    int first_del;
    bam1_t b;
} sta_pileup_t;

int sta_pileup_loop(samFile *fp, sam_hdr_t *h,
                    int (*seq_fetch)(void *client_data, samFile *fp, sam_hdr_t *h, bam1_t *b),
                    int (*seq_init)(void *client_data, samFile *fp, sam_hdr_t *h, sta_pileup_t *p),
                    int (*seq_column)(void *client_data, samFile *fp, sam_hdr_t *h, sta_pileup_t *p, int depth, hts_pos_t pos, int nth, int is_insert),
                    void (*seq_free)(void *client_data, samFile *fp, sam_hdr_t *h, sta_pileup_t *p),
                    void *client_data);

#ifdef STA_CONS_DROPIN
#define pileup_t sta_pileup_t
#define pileup sta_pileup
#define pileup_loop sta_pileup_loop
#endif
#ifdef __cplusplus
}
#endif
#endif
