/*
 * oracle/o_mods.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Base modifications for `mpileup --output-mods` (bam_plcmd.c:86-109 prints them, :356-369 parses them per read through
 * the iterator's constructor hook).  The parsing lives in HTSlib (sam_mods.c: bam_parse_basemod, bam_mods_at_qpos), which
 * is absent from the reference tree; restated here from the SAM tags specification (MM / ML, section 1.7 "Base
 * modifications") and pinned on the reference's goldens test/mpileup/expected/mp2.out and mp2-noins.out (input mod1.sam:
 * forward and reverse reads, multi-code entries "C+mh", the "N" canonical base, soft clips, an insertion carrying a mod).
 *
 *   MM:Z:  ([ACGTUN][-+]([a-z]+|[0-9]+)[.?]?(,[0-9]+)*;)*      (also spelled Mm)
 *   ML:B:C one probability per (position, code), in MM order     (also spelled Ml)
 * A delta d skips d bases of the entry's canonical kind (any base for N) before the next modified one, counted along the
 * ORIGINAL read: for a reverse-strand record from the end of SEQ, on the complemented bases.
 */
#include "o_plp.h"
#include <ctype.h>

static int comp16(int c)      /* complement of a 4-bit base code (seq_nt16 bit order: A=1 C=2 G=4 T=8) */
{
    return ((c & 1) << 3) | ((c & 2) << 1) | ((c & 4) >> 1) | ((c & 8) >> 3);
}

void omods_free(omods_t *m) { free(m->start); free(m->ent); m->start = NULL; m->ent = NULL; m->n = 0; }

/* every modification of record b, grouped by query position, MM order inside a position; 0 ok, <0 malformed tag */
int omods_parse(const orec_t *b, omods_t *m)
{
    memset(m, 0, sizeof *m);
    const uint8_t *mm = rec_aux_get(b, "MM"), *ml = rec_aux_get(b, "ML");
    if (!mm) mm = rec_aux_get(b, "Mm");
    if (!ml) ml = rec_aux_get(b, "Ml");
    const int L = b->l_qseq;
    m->start = (int *)calloc((size_t)L + 2, sizeof(int));
    if (!mm || mm[0] != 'Z' || L == 0) return 0;
    const uint8_t *mlv = NULL; uint32_t n_ml = 0;
    if (ml && ml[0] == 'B' && (ml[1] == 'C' || ml[1] == 'c')) { memcpy(&n_ml, ml + 2, 4); mlv = ml + 6; }
    const int rev = rec_is_rev(b);
    /* pass 1 collects (qpos, entry) in tag order; pass 2 buckets by position (stable) */
    size_t cap = 64, n = 0;
    omod1_t *tmp = (omod1_t *)malloc(cap * sizeof *tmp);
    int *tq = (int *)malloc(cap * sizeof *tq);
    uint32_t ml_i = 0;
    const char *p = (const char *)mm + 1;
    while (*p) {
        /* canonical base, strand */
        int base = toupper((unsigned char)*p++);
        if (!base || (*p != '+' && *p != '-')) goto bad;
        int strand = *p++ == '-';
        int want = base == 'N' ? 15 : nt16_table[base];
        /* codes: letters (one modification each) or a ChEBI number */
        int codes[64], n_codes = 0;
        if (isdigit((unsigned char)*p)) { codes[n_codes++] = -(int)strtol(p, (char **)&p, 10); }
        else while (*p >= 'a' && *p <= 'z' && n_codes < 64) codes[n_codes++] = *p++;
        if (!n_codes) goto bad;
        if (*p == '?' || *p == '.') ++p;
        /* candidate positions along the original read */
        int k = -1, cand = rev ? L : -1;
        while (*p == ',') {
            long d = strtol(p + 1, (char **)&p, 10);
            k += (int)d + 1;
            /* advance `cand` to the k-th base of the wanted kind (cumulative: continue from the previous hit) */
            int need = (int)d + 1, q = -1;
            while (need > 0) {
                cand += rev ? -1 : 1;
                if (cand < 0 || cand >= L) { cand = rev ? -1 : L; break; }
                int c = rec_seqi(b->seq, cand);
                if (rev) c = comp16(c);
                if (want == 15 || c == want) --need;
            }
            if (need == 0) q = cand;
            for (int c = 0; c < n_codes; ++c) {
                int qual = mlv && ml_i < n_ml ? mlv[ml_i] : -1;
                ++ml_i;
                if (q < 0) continue;
                if (n == cap) { cap *= 2; tmp = (omod1_t *)realloc(tmp, cap * sizeof *tmp); tq = (int *)realloc(tq, cap * sizeof *tq); }
                tmp[n].code = codes[c]; tmp[n].strand = strand; tmp[n].qual = qual; tq[n] = q; ++n;
            }
        }
        (void)k;
        if (*p == ';') ++p; else if (*p) goto bad;
    }
    for (size_t i = 0; i < n; ++i) m->start[tq[i] + 1]++;
    for (int i = 0; i < L; ++i) m->start[i + 1] += m->start[i];
    m->ent = (omod1_t *)malloc((n ? n : 1) * sizeof *m->ent);
    {
        int *fill = (int *)calloc((size_t)L + 1, sizeof(int));
        for (size_t i = 0; i < n; ++i) { int q = tq[i]; m->ent[m->start[q] + fill[q]++] = tmp[i]; }
        free(fill);
    }
    m->n = (int)n; m->l = L;
    free(tmp); free(tq);
    return 0;
bad:
    free(tmp); free(tq);
    m->n = 0; m->l = L;
    return -1;
}

/* "[+m128-h7]" for query position qpos appended to out; nothing when the base carries no modification (bam_plcmd.c:86-109) */
void omods_put(const omods_t *m, int qpos, ostr_t *out)
{
    if (!m || !m->start || qpos < 0 || qpos >= m->l || m->start[qpos + 1] == m->start[qpos]) return;
    os_putc(out, '[');
    for (int i = m->start[qpos]; i < m->start[qpos + 1] && i < m->start[qpos] + 256; ++i) {
        const omod1_t *e = &m->ent[i];
        os_putc(out, "+-"[e->strand]);
        if (e->code < 0) { os_putc(out, '('); os_putll(out, -e->code); os_putc(out, ')'); }
        else os_putc(out, e->code);
        if (e->qual >= 0) os_putll(out, e->qual);
    }
    os_putc(out, ']');
}
