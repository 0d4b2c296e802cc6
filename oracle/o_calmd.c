/*
 * oracle/o_calmd.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * calmd's per-record arithmetic -- the MD/NM recomputation and the BAQ tag writer.
 *   bam_fillmd1_core                 bam_md.c:64-224   (in the reference tree: restated line by line)
 *   per-record sequence of calls     bam_md.c:457-497  (sam_prob_realn with -r, then bam_fillmd1_core)
 *   sam_prob_realn                   HTSlib realn.c (absent): o_baq.c, pinned on the mpileup BAQ goldens; its tag-writing
 *                                    tail (BQ:Z without -A, ZQ:Z with -A) is restated here from the published behaviour
 *   aux bookkeeping + sam_write1     bam_md.c:156-193 (NM / MD kept, replaced or appended), :195-199 (-d), :486-489; the SAM
 *                                    text of a record is HTSlib's sam_format1 (absent: SAM spec section 1.4/1.5 is the format,
 *                                    integers of every width print as `i`, floats through kputd)
 * PINNING: the reference's own calmd test only checks the container magic (test/test.pl:3652-3661), but its test inputs carry
 * MD:Z / NM:i tags written by the aligner against the very FASTA the test uses: a record whose stored tags are right must
 * leave calmd byte for byte as it came in (bam_md.c:158-191 touches nothing then), so tests/test_oracle_goldens.py requires
 * `calmd in.sam ref.fa` to reproduce test/dat/mpileup.{1,2,3}.sam themselves (1 034 mapped records).  The BQ/ZQ strings are
 * pinned only through o_baq.c's mpileup goldens (qualities after BAQ), not as tags.
 *
 * Output: SAM text with the input's header (no @PG line is added: as --no-PG).  -b / -u (BAM output) are refused.  -C (round 5) lowers
 * MAPQ by sam_cap_mapq as bam_md.c:480-483 does, -1 return included (it lands in the unsigned field as 255).
 */
#include "o_plp.h"
#include <ctype.h>
#include <getopt.h>

#define USE_EQUAL 1
#define DROP_TAG 2
#define BIN_QUAL 4
#define UPDATE_NM 8
#define UPDATE_MD 16

typedef struct { int nm; ostr_t md; int has; } mdres_t;

/* bam_md.c:64-224 without the aux bookkeeping: computes NM and MD, applies -e, -n and -q to seq / qual */
static int fillmd1_core(orec_t *b, const char *ref, hpos_t ref_len, int flag, int max_nm, mdres_t *out, unsigned *skipped)
{
    uint8_t *seq = b->seq;
    uint32_t *cigar = b->cigar;
    int i, qpos, matched = 0, nm = 0;
    hpos_t rpos;
    out->has = 0; os_clear(&out->md);
    if (b->l_qseq == 0) { if (skipped) (*skipped)++; return 0; }
    for (i = qpos = 0, rpos = b->pos; i < (int)b->n_cigar; ++i) {
        int j, oplen = (int)cig_len(cigar[i]), op = cig_op(cigar[i]);
        if (op == C_M || op == C_EQ || op == C_X) {
            for (j = 0; j < oplen; ++j) {
                int c1, c2, z = qpos + j;
                if (rpos + j >= ref_len || z >= b->l_qseq || ref[rpos + j] == '\0') break;
                c1 = rec_seqi(seq, z);
                c2 = nt16_table[(uint8_t)ref[rpos + j]];
                if ((c1 == c2 && c1 != 15 && c2 != 15) || c1 == 0) {
                    if (flag & USE_EQUAL) seq[z / 2] &= (z & 1) ? 0xf0 : 0x0f;
                    ++matched;
                } else {
                    os_putll(&out->md, matched);
                    os_putc(&out->md, toupper((unsigned char)ref[rpos + j]));
                    matched = 0; ++nm;
                }
            }
            if (j < oplen) break;
            rpos += oplen; qpos += oplen;
        } else if (op == C_D) {
            os_putll(&out->md, matched);
            os_putc(&out->md, '^');
            for (j = 0; j < oplen; ++j) {
                if (rpos + j >= ref_len || ref[rpos + j] == '\0') break;
                os_putc(&out->md, toupper((unsigned char)ref[rpos + j]));
            }
            matched = 0;
            rpos += j; nm += j;
            if (j < oplen) break;
        } else if (op == C_I || op == C_S) {
            qpos += oplen;
            if (op == C_I) nm += oplen;
        } else if (op == C_N) rpos += oplen;
    }
    os_putll(&out->md, matched);
    if (max_nm > 0 && nm >= max_nm) {
        for (i = qpos = 0, rpos = b->pos; i < (int)b->n_cigar; ++i) {
            int j, oplen = (int)cig_len(cigar[i]), op = cig_op(cigar[i]);
            if (op == C_M || op == C_EQ || op == C_X) {
                for (j = 0; j < oplen; ++j) {
                    int c1, c2, z = qpos + j;
                    if (rpos + j >= ref_len || z >= b->l_qseq || ref[rpos + j] == '\0') break;
                    c1 = rec_seqi(seq, z);
                    c2 = nt16_table[(uint8_t)ref[rpos + j]];
                    if ((c1 == c2 && c1 != 15 && c2 != 15) || c1 == 0) {
                        seq[z / 2] |= (z & 1) ? 0x0f : 0xf0;
                        b->qual[z] = 0;
                    }
                }
                if (j < oplen) break;
                rpos += oplen; qpos += oplen;
            } else if (op == C_D || op == C_N) rpos += oplen;
            else if (op == C_I || op == C_S) qpos += oplen;
        }
    }
    if ((flag & (UPDATE_NM | UPDATE_MD)) && !(b->flag & F_UNMAP)) { out->has = 1; out->nm = nm; }
    if (flag & BIN_QUAL)
        for (i = 0; i < b->l_qseq; ++i) if (b->qual[i] >= 3) b->qual[i] = (uint8_t)(b->qual[i] / 10 * 10 + 7);
    return 0;
}

/* the appended tags live behind the record's own aux block (bam_aux_append always writes at the end) */
static void app_tag(ostr_t *x, const char tag[2], char type, const void *data, size_t len)
{
    os_putsn(x, tag, 2); os_putc(x, type); os_putsn(x, (const char *)data, len);
}

/* sam_format1's aux part: every integer width prints as i, f / d through kputd, B arrays comma separated */
static int put_aux_text(ostr_t *o, const uint8_t *p, const uint8_t *end)
{
    while (p + 3 <= end) {
        const uint8_t *nx = rec_aux_next(p, end);
        if (!nx) return -1;
        int t = p[2];
        const uint8_t *v = p + 3;
        os_putc(o, '\t'); os_putc(o, p[0]); os_putc(o, p[1]); os_putc(o, ':');
        if (t == 'A') { os_puts(o, "A:"); os_putc(o, v[0]); }
        else if (t == 'c') { os_puts(o, "i:"); os_putll(o, (int8_t)v[0]); }
        else if (t == 'C') { os_puts(o, "i:"); os_putll(o, v[0]); }
        else if (t == 's') { int16_t x; memcpy(&x, v, 2); os_puts(o, "i:"); os_putll(o, x); }
        else if (t == 'S') { uint16_t x; memcpy(&x, v, 2); os_puts(o, "i:"); os_putll(o, x); }
        else if (t == 'i') { int32_t x; memcpy(&x, v, 4); os_puts(o, "i:"); os_putll(o, x); }
        else if (t == 'I') { uint32_t x; memcpy(&x, v, 4); os_puts(o, "i:"); os_putll(o, x); }
        else if (t == 'f') { float x; memcpy(&x, v, 4); os_puts(o, "f:"); o_put_double(o, x); }
        else if (t == 'd') { double x; memcpy(&x, v, 8); os_puts(o, "d:"); o_put_double(o, x); }
        else if (t == 'Z' || t == 'H') { os_putc(o, t); os_putc(o, ':'); os_puts(o, (const char *)v); }
        else if (t == 'B') {
            int sub = v[0]; uint32_t n; memcpy(&n, v + 1, 4);
            const uint8_t *q = v + 5;
            os_puts(o, "B:"); os_putc(o, sub);
            for (uint32_t k = 0; k < n; ++k) {
                os_putc(o, ',');
                switch (sub) {
                case 'c': os_putll(o, (int8_t)q[0]); q += 1; break;
                case 'C': os_putll(o, q[0]); q += 1; break;
                case 's': { int16_t x; memcpy(&x, q, 2); os_putll(o, x); q += 2; break; }
                case 'S': { uint16_t x; memcpy(&x, q, 2); os_putll(o, x); q += 2; break; }
                case 'i': { int32_t x; memcpy(&x, q, 4); os_putll(o, x); q += 4; break; }
                case 'I': { uint32_t x; memcpy(&x, q, 4); os_putll(o, x); q += 4; break; }
                case 'f': { float x; memcpy(&x, q, 4); o_put_double(o, x); q += 4; break; }
                default: return -1;
                }
            }
        } else return -1;
        p = nx;
    }
    return 0;
}

/* one SAM record line (sam_format1): the eleven mandatory fields, then the record's aux block and what was appended to it */
static int put_sam_record(ostr_t *o, const ohdr_t *h, const orec_t *b, const ostr_t *extra)
{
    os_puts(o, b->qname); os_putc(o, '\t');
    os_putll(o, b->flag); os_putc(o, '\t');
    os_puts(o, b->tid >= 0 && b->tid < h->n_ref ? h->name[b->tid] : "*"); os_putc(o, '\t');
    os_putll(o, (long long)b->pos + 1); os_putc(o, '\t');
    os_putll(o, b->mapq); os_putc(o, '\t');
    if (b->n_cigar) for (uint32_t k = 0; k < b->n_cigar; ++k) { os_putll(o, cig_len(b->cigar[k])); os_putc(o, "MIDNSHP=XB"[cig_op(b->cigar[k])]); }
    else os_putc(o, '*');
    os_putc(o, '\t');
    if (b->mtid < 0) os_putc(o, '*');
    else if (b->mtid == b->tid) os_putc(o, '=');
    else os_puts(o, b->mtid < h->n_ref ? h->name[b->mtid] : "*");
    os_putc(o, '\t');
    os_putll(o, (long long)b->mpos + 1); os_putc(o, '\t');
    os_putll(o, (long long)b->isize); os_putc(o, '\t');
    if (b->l_qseq) {
        for (int i = 0; i < b->l_qseq; ++i) os_putc(o, nt16_str[rec_seqi(b->seq, i)]);
        os_putc(o, '\t');
        if (b->qual[0] == 0xff) os_putc(o, '*'); else for (int i = 0; i < b->l_qseq; ++i) os_putc(o, b->qual[i] + 33);
    } else os_puts(o, "*\t*");
    if (put_aux_text(o, b->aux, b->aux + b->l_aux) < 0) return -1;
    if (extra->l && put_aux_text(o, (const uint8_t *)extra->s, (const uint8_t *)extra->s + extra->l) < 0) return -1;
    os_putc(o, '\n');
    return 0;
}

int o_main_calmd(int argc, char *argv[])
{
    int c, flt_flag = UPDATE_NM | UPDATE_MD, is_realn = 0, baq_flag = 0, max_nm = 0, quiet = 0, capQ = 0;
    static const struct option lopts[] = { { "no-PG", no_argument, NULL, 1 }, { NULL, 0, NULL, 0 } };
    optind = 1;
    while ((c = getopt_long(argc, argv, "EqQreuNhbSC:n:Ad", lopts, NULL)) >= 0) {
        switch (c) {
        case 'e': flt_flag |= USE_EQUAL; break;
        case 'd': flt_flag |= DROP_TAG; break;
        case 'N': flt_flag &= ~(UPDATE_MD | UPDATE_NM); break;
        case 'r': is_realn = 1; break;
        case 'A': baq_flag |= 1; break;
        case 'E': baq_flag |= 2; break;
        case 'q': flt_flag |= BIN_QUAL; break;
        case 'n': max_nm = atoi(optarg); break;
        case 'C': capQ = atoi(optarg); break;
        case 'Q': quiet = 1; break;
        case 'h': case 'S': case 1: break;
        default: fprintf(stderr, "[calmd] option -%c is not part of the restated rows\n", c); return 1;
        }
    }
    if (argc - optind != 2) { fprintf(stderr, "usage: oracle_samtools calmd [-erAEqdNQ] [-n max_nm] in.sam ref.fa\n"); return 1; }
    oreader_t *rd = rd_open(argv[optind]);
    if (!rd) { fprintf(stderr, "[calmd] failed to open %s\n", argv[optind]); return 1; }
    ohdr_t *h = rd_header(rd);
    ofasta_t *fa = fa_load(argv[optind + 1]);
    if (!fa) { fprintf(stderr, "[calmd] failed to open reference file '%s'\n", argv[optind + 1]); return 1; }
    if (h->text && h->text[0]) { fputs(h->text, stdout); if (h->text[strlen(h->text) - 1] != '\n') putchar('\n'); }
    orec_t b; memset(&b, 0, sizeof b);
    mdres_t md; memset(&md, 0, sizeof md);
    ostr_t extra = { 0, 0, NULL }, line = { 0, 0, NULL };
    unsigned skipped = 0;
    int r, last_tid = -2, ret = 0;
    const char *ref = NULL; hpos_t ref_len = 0;
    uint8_t *q0 = NULL; size_t q0_m = 0;
    while ((r = rd_next(rd, &b)) >= 0) {
        md.has = 0;
        os_clear(&extra);
        if ((size_t)b.l_qseq + 1 > q0_m) { q0_m = (size_t)b.l_qseq * 2 + 64; q0 = (uint8_t *)realloc(q0, q0_m); }
        if (b.tid >= 0) {
            if (b.tid != last_tid) {
                ref = fa_fetch(fa, h->name[b.tid], &ref_len);
                last_tid = b.tid;
                if (!ref) {
                    fprintf(stderr, "[bam_fillmd] fail to find sequence '%s' in the reference.\n", h->name[b.tid]);
                    if (is_realn || capQ > 10) { ret = 1; break; }     /* bam_md.c:471 */
                }
            }
            if (is_realn) {
                /* sam_prob_realn (bam_md.c:474-479): with a BQ:Z / ZQ:Z tag in the record it converts, renames or leaves (o_baq.c); a
                   fresh computation ends with bq[i] = 64 + (quality as read - quality after BAQ) appended as ZQ:Z (-A, qualities
                   changed) or BQ:Z (qualities as read) */
                const int tagged = rec_aux_get(&b, "BQ") || rec_aux_get(&b, "ZQ");
                if (tagged) o_prob_realn(&b, ref, ref_len, baq_flag);
                else {
                    if (b.l_qseq) memcpy(q0, b.qual, (size_t)b.l_qseq);
                    if (o_prob_realn(&b, ref, ref_len, baq_flag | 1) == 0) {
                        for (int i = 0; i < b.l_qseq; ++i) {
                            uint8_t adj = b.qual[i], t = (uint8_t)(64 + (q0[i] - adj));
                            if (!(baq_flag & 1)) b.qual[i] = q0[i];
                            q0[i] = t;
                        }
                        q0[b.l_qseq] = 0;
                        app_tag(&extra, (baq_flag & 1) ? "ZQ" : "BQ", 'Z', q0, (size_t)b.l_qseq + 1);
                    } else if (b.l_qseq) memcpy(b.qual, q0, (size_t)b.l_qseq);
                }
            }
            if (capQ > 10) {            /* bam_md.c:480-483; a -1 lands in the unsigned field as 255 */
                int q = o_cap_mapq(&b, ref, ref_len, capQ);
                if ((int)b.mapq > q) b.mapq = (uint8_t)q;
            }
            if (ref) {
                if (b.l_qseq == 0 && !quiet)
                    fprintf(stderr, "[bam_fillmd1] no sequence in alignment record for '%s' at %s:%lld, skipped\n", b.qname, h->name[b.tid], (long long)b.pos + 1);
                const int binq = flt_flag & BIN_QUAL;
                if (fillmd1_core(&b, ref, ref_len, flt_flag & ~BIN_QUAL, max_nm, &md, &skipped) < 0) { ret = 1; break; }
                if (b.l_qseq) {
                    /* bam_md.c:156-193 */
                    if ((flt_flag & UPDATE_NM) && !(b.flag & F_UNMAP)) {
                        const uint8_t *old_nm = rec_aux_get(&b, "NM");
                        int32_t nm = md.nm;
                        if (!old_nm) app_tag(&extra, "NM", 'i', &nm, 4);
                        else {
                            long long old = 0;
                            switch (old_nm[0]) {
                            case 'c': old = (int8_t)old_nm[1]; break;
                            case 'C': old = old_nm[1]; break;
                            case 's': { int16_t x; memcpy(&x, old_nm + 1, 2); old = x; break; }
                            case 'S': { uint16_t x; memcpy(&x, old_nm + 1, 2); old = x; break; }
                            case 'i': { int32_t x; memcpy(&x, old_nm + 1, 4); old = x; break; }
                            case 'I': { uint32_t x; memcpy(&x, old_nm + 1, 4); old = x; break; }
                            }
                            if ((int32_t)old != nm) {
                                if (!quiet) fprintf(stderr, "[bam_fillmd1] different NM for read '%s': %d -> %d\n", b.qname, (int)old, nm);
                                rec_aux_del(&b, old_nm);
                                app_tag(&extra, "NM", 'i', &nm, 4);
                            }
                        }
                    }
                    if ((flt_flag & UPDATE_MD) && !(b.flag & F_UNMAP)) {
                        const uint8_t *old_md = rec_aux_get(&b, "MD");
                        const char *ns = md.md.s ? md.md.s : "";
                        if (!old_md) app_tag(&extra, "MD", 'Z', ns, md.md.l + 1);
                        else {
                            int is_diff = 0;
                            if (strlen((const char *)old_md + 1) == md.md.l) {
                                size_t k;
                                for (k = 0; k < md.md.l; ++k) if (toupper(old_md[k + 1]) != toupper((unsigned char)ns[k])) break;
                                if (k < md.md.l) is_diff = 1;
                            } else is_diff = 1;
                            if (is_diff) {
                                if (!quiet) fprintf(stderr, "[bam_fillmd1] different MD for read '%s': '%s' -> '%s'\n", b.qname, old_md + 1, ns);
                                rec_aux_del(&b, old_md);
                                app_tag(&extra, "MD", 'Z', ns, md.md.l + 1);
                            }
                        }
                    }
                    if (flt_flag & DROP_TAG) {
                        /* bam_aux_drop_other: nothing but the RG tag stays */
                        const uint8_t *rg = rec_aux_get(&b, "RG");
                        if (rg) {
                            const uint8_t *beg = rg - 2, *nx = rec_aux_next(beg, b.aux + b.l_aux);
                            size_t n = (size_t)(nx - beg);
                            memmove(b.aux, beg, n); b.l_aux = (int)n;
                        } else b.l_aux = 0;
                        os_clear(&extra);
                    }
                    if (binq) for (int i = 0; i < b.l_qseq; ++i) if (b.qual[i] >= 3) b.qual[i] = (uint8_t)(b.qual[i] / 10 * 10 + 7);
                }
            }
        }
        os_clear(&line);
        if (put_sam_record(&line, h, &b, &extra) < 0) { fprintf(stderr, "[calmd] Corrupt aux data\n"); ret = 1; break; }
        fwrite(line.s, 1, line.l, stdout);
    }
    if (r < -1) { fprintf(stderr, "[bam_fillmd] Error reading input.\n"); ret = 1; }
    if (skipped) fprintf(stderr, "[calmd] Warning: %u records skipped due to no query sequence\n", skipped);
    rec_free(&b); free(md.md.s); free(q0); free(extra.s); free(line.s);
    fa_free(fa); rd_close(rd);
    return ret;
}
