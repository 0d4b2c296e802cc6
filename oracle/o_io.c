/*
 * oracle/o_io.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Stand-in for the HTSlib 1.23.1 I/O used on the mpileup/depth path
 * (sam_open/sam_hdr_read/sam_read1/sam_itr_next, fai_load/faidx_fetch_seq64,
 * hts_parse_reg, bam_str2flag) plus a restatement of samtools bedidx.c.
 * HTSlib is not in /root/reference; formats follow the SAM/BAM specification.
 */
#include "o_common.h"
#include <zlib.h>
#include <ctype.h>
#include <errno.h>

const char nt16_str[] = "=ACMGRSVTWYHKDBN";
const int nt16_int[16] = { 4, 0, 1, 4, 2, 4, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4 };
/* seq_nt16_table: IUPAC -> 4 bit; everything else 15 */
const unsigned char nt16_table[256] = {
#define X15 15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15
    X15, X15,
    15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,
    1, 2, 4, 8, 15,15,15,15, 15,15,15,15,15, 0 /*=*/,15,15,
    15, 1,14, 2, 13,15,15, 4, 11,15,15,12, 15, 3,15,15,
    15,15, 5, 6,  8,15, 7, 9, 15,10,15,15, 15,15,15,15,
    15, 1,14, 2, 13,15,15, 4, 11,15,15,12, 15, 3,15,15,
    15,15, 5, 6,  8,15, 7, 9, 15,10,15,15, 15,15,15,15,
    X15, X15, X15, X15, X15, X15, X15, X15
#undef X15
};

/* ---------------- ostr ---------------- */
void os_reserve(ostr_t *s, size_t extra)
{
    size_t need = s->l + extra + 1;
    if (need <= s->m) return;
    size_t m = s->m ? s->m : 64;
    while (m < need) m <<= 1;
    s->s = (char *)realloc(s->s, m);
    if (!s->s) { perror("realloc"); exit(2); }
    s->m = m;
}
void os_putsn(ostr_t *s, const char *p, size_t n)
{
    os_reserve(s, n + 1);
    memcpy(s->s + s->l, p, n);
    s->l += n;
    s->s[s->l] = 0;
}
void os_putll(ostr_t *s, long long v)
{
    char buf[32]; int n = 0;
    unsigned long long x = v < 0 ? 0ULL - (unsigned long long)v : (unsigned long long)v;
    do { buf[n++] = (char)('0' + x % 10); x /= 10; } while (x);
    if (v < 0) buf[n++] = '-';
    os_reserve(s, (size_t)n + 1);
    while (n) s->s[s->l++] = buf[--n];
    s->s[s->l] = 0;
}

/* ---------------- records ---------------- */
void rec_free(orec_t *r) { free(r->data); memset(r, 0, sizeof(*r)); }

static void rec_layout(orec_t *r, size_t l_qname, size_t n_cigar, size_t l_qseq, size_t l_aux)
{
    /* data = qname | pad to 4 | cigar | seq | qual | aux */
    size_t o_cig = (l_qname + 1 + 3) & ~(size_t)3;
    size_t o_seq = o_cig + 4 * n_cigar;
    size_t o_qual = o_seq + (l_qseq + 1) / 2;
    size_t o_aux = o_qual + l_qseq;
    size_t tot = o_aux + l_aux + 8;
    if (tot > r->m_data) {
        size_t m = r->m_data ? r->m_data : 256;
        while (m < tot) m <<= 1;
        r->data = (uint8_t *)realloc(r->data, m);
        if (!r->data) { perror("realloc"); exit(2); }
        r->m_data = m;
    }
    r->qname = (char *)r->data;
    r->cigar = (uint32_t *)(r->data + o_cig);
    r->seq = r->data + o_seq;
    r->qual = r->data + o_qual;
    r->aux = r->data + o_aux;
    r->n_cigar = (uint32_t)n_cigar;
    r->l_qseq = (int32_t)l_qseq;
    r->l_aux = (int)l_aux;
}

int rec_copy(orec_t *dst, const orec_t *src)
{
    size_t lq = strlen(src->qname);
    uint8_t *d = dst->data; size_t m = dst->m_data;
    *dst = *src;
    dst->data = d; dst->m_data = m;
    rec_layout(dst, lq, src->n_cigar, (size_t)src->l_qseq, (size_t)src->l_aux);
    memcpy(dst->qname, src->qname, lq + 1);
    memcpy(dst->cigar, src->cigar, 4 * (size_t)src->n_cigar);
    memcpy(dst->seq, src->seq, ((size_t)src->l_qseq + 1) / 2);
    memcpy(dst->qual, src->qual, (size_t)src->l_qseq);
    memcpy(dst->aux, src->aux, (size_t)src->l_aux);
    return 0;
}

hpos_t rec_rlen(const orec_t *r)
{
    hpos_t l = 0;
    for (uint32_t k = 0; k < r->n_cigar; ++k) {
        int op = cig_op(r->cigar[k]);
        if (op == C_M || op == C_D || op == C_N || op == C_EQ || op == C_X) l += cig_len(r->cigar[k]);
    }
    return l;
}

hpos_t rec_endpos(const orec_t *r)
{
    hpos_t rlen = (r->flag & F_UNMAP) ? 0 : rec_rlen(r);
    if (rlen == 0) rlen = 1;
    return r->pos + rlen;
}

static int aux_type_size(int t)
{
    switch (t) {
    case 'A': case 'c': case 'C': return 1;
    case 's': case 'S': return 2;
    case 'i': case 'I': case 'f': return 4;
    case 'd': return 8;
    default: return 0;
    }
}

const uint8_t *rec_aux_get(const orec_t *r, const char tag[2])
{
    const uint8_t *p = r->aux, *end = r->aux + r->l_aux;
    while (p + 3 <= end) {
        int hit = (p[0] == (uint8_t)tag[0] && p[1] == (uint8_t)tag[1]);
        int t = p[2];
        const uint8_t *v = p + 2;
        p += 3;
        if (t == 'Z' || t == 'H') { while (p < end && *p) ++p; ++p; }
        else if (t == 'B') {
            if (p + 5 > end) return NULL;
            int sz = aux_type_size(p[0]);
            uint32_t n; memcpy(&n, p + 1, 4);
            p += 5 + (size_t)sz * n;
        } else {
            int sz = aux_type_size(t);
            if (!sz) return NULL;
            p += sz;
        }
        if (hit) return v;
    }
    return NULL;
}

/* one field further: p at a tag's first byte -> the next tag's first byte (NULL on damage) */
const uint8_t *rec_aux_next(const uint8_t *p, const uint8_t *end)
{
    if (p + 3 > end) return NULL;
    int t = p[2];
    p += 3;
    if (t == 'Z' || t == 'H') { while (p < end && *p) ++p; ++p; }
    else if (t == 'B') {
        if (p + 5 > end) return NULL;
        int sz = aux_type_size(p[0]);
        uint32_t n; memcpy(&n, p + 1, 4);
        if (!sz) return NULL;
        p += 5 + (size_t)sz * n;
    } else {
        int sz = aux_type_size(t);
        if (!sz) return NULL;
        p += sz;
    }
    return p <= end ? p : NULL;
}

/* bam_aux_del: v = what rec_aux_get returned (the type byte); the field is cut out of the block */
int rec_aux_del(orec_t *r, const uint8_t *v)
{
    uint8_t *beg = (uint8_t *)v - 2, *end = r->aux + r->l_aux;
    const uint8_t *nx = rec_aux_next(beg, end);
    if (!nx) return -1;
    memmove(beg, nx, (size_t)(end - nx));
    r->l_aux -= (int)(nx - beg);
    return 0;
}

/* ---------------- header ---------------- */
int hdr_name2tid(const ohdr_t *h, const char *name)
{
    for (int i = 0; i < h->n_ref; ++i) if (strcmp(h->name[i], name) == 0) return i;
    return -1;
}
void hdr_free(ohdr_t *h)
{
    if (!h) return;
    for (int i = 0; i < h->n_ref; ++i) free(h->name[i]);
    free(h->name); free(h->len); free(h->text); free(h);
}
static void hdr_add_ref(ohdr_t *h, const char *name, size_t ln, hpos_t len)
{
    h->name = (char **)realloc(h->name, sizeof(char *) * (size_t)(h->n_ref + 1));
    h->len = (hpos_t *)realloc(h->len, sizeof(hpos_t) * (size_t)(h->n_ref + 1));
    h->name[h->n_ref] = (char *)malloc(ln + 1);
    memcpy(h->name[h->n_ref], name, ln); h->name[h->n_ref][ln] = 0;
    h->len[h->n_ref] = len;
    h->n_ref++;
}
static void hdr_parse_sq(ohdr_t *h)
{
    const char *p = h->text;
    while (p && *p) {
        const char *eol = strchr(p, '\n');
        size_t ll = eol ? (size_t)(eol - p) : strlen(p);
        if (ll >= 3 && memcmp(p, "@SQ", 3) == 0) {
            const char *sn = NULL; size_t snl = 0; hpos_t ln = 0;
            const char *q = p + 3, *e = p + ll;
            while (q < e) {
                if (*q == '\t') { ++q; continue; }
                const char *f = q;
                while (q < e && *q != '\t') ++q;
                if (q - f >= 3 && f[2] == ':') {
                    if (f[0] == 'S' && f[1] == 'N') { sn = f + 3; snl = (size_t)(q - f - 3); }
                    else if (f[0] == 'L' && f[1] == 'N') ln = strtoll(f + 3, NULL, 10);
                }
            }
            if (sn) hdr_add_ref(h, sn, snl, ln);
        }
        p = eol ? eol + 1 : NULL;
    }
}

/* ---------------- reader ---------------- */
struct oreader {
    gzFile fp;
    int is_bam;
    ohdr_t *hdr;
    ostr_t line;          /* SAM: current line */
    int have_line;        /* SAM: first record line already read */
    uint8_t *blk; size_t m_blk;
    int reg_tid; hpos_t reg_beg, reg_end; int has_reg;
    /* buffered input */
    uint8_t buf[1 << 16]; int buf_l, buf_p; int eof;
};

static int rd_fill(oreader_t *r)
{
    if (r->eof) return 0;
    int n = gzread(r->fp, r->buf, sizeof(r->buf));
    if (n <= 0) { r->eof = 1; r->buf_l = r->buf_p = 0; return 0; }
    r->buf_l = n; r->buf_p = 0;
    return n;
}
static int rd_bytes(oreader_t *r, void *dst, size_t n)
{
    uint8_t *d = (uint8_t *)dst;
    size_t got = 0;
    while (got < n) {
        if (r->buf_p >= r->buf_l && !rd_fill(r)) break;
        size_t k = (size_t)(r->buf_l - r->buf_p);
        if (k > n - got) k = n - got;
        memcpy(d + got, r->buf + r->buf_p, k);
        r->buf_p += (int)k; got += k;
    }
    return (int)got;
}
static int rd_line(oreader_t *r, ostr_t *s)
{
    os_clear(s);
    int any = 0;
    for (;;) {
        if (r->buf_p >= r->buf_l && !rd_fill(r)) break;
        uint8_t *b = r->buf + r->buf_p;
        int k = r->buf_l - r->buf_p;
        uint8_t *nl = (uint8_t *)memchr(b, '\n', (size_t)k);
        any = 1;
        if (nl) {
            os_putsn(s, (char *)b, (size_t)(nl - b));
            r->buf_p += (int)(nl - b) + 1;
            if (s->l && s->s[s->l - 1] == '\r') s->s[--s->l] = 0;
            return 1;
        }
        os_putsn(s, (char *)b, (size_t)k);
        r->buf_p = r->buf_l;
    }
    return any;
}

oreader_t *rd_open(const char *fn)
{
    oreader_t *r = (oreader_t *)calloc(1, sizeof(*r));
    r->fp = strcmp(fn, "-") ? gzopen(fn, "rb") : gzdopen(fileno(stdin), "rb");
    if (!r->fp) { free(r); return NULL; }
    gzbuffer(r->fp, 1 << 17);
    r->hdr = (ohdr_t *)calloc(1, sizeof(ohdr_t));
    rd_fill(r);
    if (r->buf_l >= 4 && memcmp(r->buf, "BAM\1", 4) == 0) {
        r->is_bam = 1;
        r->buf_p = 4;
        int32_t l_text, n_ref;
        if (rd_bytes(r, &l_text, 4) != 4) goto fail;
        r->hdr->text = (char *)malloc((size_t)l_text + 1);
        if (rd_bytes(r, r->hdr->text, (size_t)l_text) != l_text) goto fail;
        r->hdr->text[l_text] = 0;
        if (rd_bytes(r, &n_ref, 4) != 4) goto fail;
        for (int i = 0; i < n_ref; ++i) {
            int32_t l_name, l_ref;
            if (rd_bytes(r, &l_name, 4) != 4) goto fail;
            char *nm = (char *)malloc((size_t)l_name + 1);
            if (rd_bytes(r, nm, (size_t)l_name) != l_name) { free(nm); goto fail; }
            nm[l_name] = 0;
            if (rd_bytes(r, &l_ref, 4) != 4) { free(nm); goto fail; }
            hdr_add_ref(r->hdr, nm, strlen(nm), l_ref);
            free(nm);
        }
        /* @SQ LN in the text header overrides the 32-bit binary length
         * (HTSlib long-reference convention, test/large_pos) */
        {
            ohdr_t tmp; memset(&tmp, 0, sizeof(tmp));
            tmp.text = r->hdr->text;
            hdr_parse_sq(&tmp);
            for (int i = 0; i < tmp.n_ref; ++i) {
                int t = hdr_name2tid(r->hdr, tmp.name[i]);
                if (t >= 0 && tmp.len[i] > r->hdr->len[t]) r->hdr->len[t] = tmp.len[i];
                free(tmp.name[i]);
            }
            free(tmp.name); free(tmp.len);
        }
    } else {
        ostr_t text = { 0, 0, NULL };
        while (rd_line(r, &r->line)) {
            if (r->line.l == 0) continue;
            if (r->line.s[0] != '@') { r->have_line = 1; break; }
            os_putsn(&text, r->line.s, r->line.l);
            os_putc(&text, '\n');
        }
        r->hdr->text = text.s ? text.s : strdup("");
        hdr_parse_sq(r->hdr);
    }
    return r;
fail:
    rd_close(r);
    return NULL;
}

ohdr_t *rd_header(oreader_t *r) { return r->hdr; }

void rd_set_region(oreader_t *r, int tid, hpos_t beg, hpos_t end)
{
    r->has_reg = 1; r->reg_tid = tid; r->reg_beg = beg; r->reg_end = end;
}

void rd_close(oreader_t *r)
{
    if (!r) return;
    if (r->fp) gzclose(r->fp);
    hdr_free(r->hdr);
    free(r->line.s); free(r->blk);
    free(r);
}

/* append one SAM aux field "TG:T:value" as BAM binary */
static void aux_append_text(ostr_t *out, const char *f, size_t n)
{
    if (n < 5 || f[2] != ':' || f[4] != ':') return;
    char type = f[3];
    const char *v = f + 5; size_t vl = n - 5;
    os_putsn(out, f, 2);
    if (type == 'A') { os_putc(out, 'A'); os_putc(out, vl ? v[0] : ' '); }
    else if (type == 'i') {
        long long x = strtoll(v, NULL, 10);
        if (x < 0) {
            if (x >= INT8_MIN) { os_putc(out, 'c'); int8_t y = (int8_t)x; os_putsn(out, (char *)&y, 1); }
            else if (x >= INT16_MIN) { os_putc(out, 's'); int16_t y = (int16_t)x; os_putsn(out, (char *)&y, 2); }
            else { os_putc(out, 'i'); int32_t y = (int32_t)x; os_putsn(out, (char *)&y, 4); }
        } else {
            if (x <= UINT8_MAX) { os_putc(out, 'C'); uint8_t y = (uint8_t)x; os_putsn(out, (char *)&y, 1); }
            else if (x <= UINT16_MAX) { os_putc(out, 'S'); uint16_t y = (uint16_t)x; os_putsn(out, (char *)&y, 2); }
            else { os_putc(out, 'I'); uint32_t y = (uint32_t)x; os_putsn(out, (char *)&y, 4); }
        }
    } else if (type == 'f') { os_putc(out, 'f'); float y = strtof(v, NULL); os_putsn(out, (char *)&y, 4); }
    else if (type == 'd') { os_putc(out, 'd'); double y = strtod(v, NULL); os_putsn(out, (char *)&y, 8); }
    else if (type == 'Z' || type == 'H') { os_putc(out, type); os_putsn(out, v, vl); out->s[out->l++] = 0; os_reserve(out, 1); out->s[out->l] = 0; }
    else if (type == 'B' && vl >= 1) {
        char sub = v[0];
        int sz = aux_type_size(sub);
        if (!sz) { out->l -= 2; return; }
        os_putc(out, 'B'); os_putc(out, sub);
        size_t cnt_off = out->l; uint32_t cnt = 0;
        os_putsn(out, "\0\0\0\0", 4);
        const char *p = v + 1, *e = v + vl;
        while (p < e) {
            if (*p == ',') { ++p; continue; }
            char *q;
            if (sub == 'f') { float y = strtof(p, &q); os_putsn(out, (char *)&y, 4); }
            else {
                long long x = strtoll(p, &q, 10);
                if (sz == 1) { uint8_t y = (uint8_t)x; os_putsn(out, (char *)&y, 1); }
                else if (sz == 2) { uint16_t y = (uint16_t)x; os_putsn(out, (char *)&y, 2); }
                else { uint32_t y = (uint32_t)x; os_putsn(out, (char *)&y, 4); }
            }
            if (q == p) break;
            p = q; ++cnt;
        }
        memcpy(out->s + cnt_off, &cnt, 4);
    } else out->l -= 2;
}

static int parse_sam_line(oreader_t *r, orec_t *rec)
{
    char *f[12]; size_t fl[12]; int nf = 0;
    char *p = r->line.s, *e = r->line.s + r->line.l;
    char *aux_start = NULL;
    while (nf < 11) {
        char *t = (char *)memchr(p, '\t', (size_t)(e - p));
        f[nf] = p; fl[nf] = t ? (size_t)(t - p) : (size_t)(e - p);
        ++nf;
        if (!t) { p = e; break; }
        p = t + 1;
    }
    if (nf < 11) return -2;
    aux_start = p <= e ? p : e;
    for (int i = 0; i < 11; ++i) f[i][fl[i]] = 0;
    /* cigar count */
    size_t n_cig = 0;
    if (!(fl[5] == 1 && f[5][0] == '*'))
        for (char *c = f[5]; *c; ++c) if (!isdigit((unsigned char)*c)) ++n_cig;
    size_t l_seq = (fl[9] == 1 && f[9][0] == '*') ? 0 : fl[9];
    /* aux -> binary */
    ostr_t aux = { 0, 0, NULL };
    for (char *a = aux_start; a < e;) {
        char *t = (char *)memchr(a, '\t', (size_t)(e - a));
        size_t n = t ? (size_t)(t - a) : (size_t)(e - a);
        aux_append_text(&aux, a, n);
        if (!t) break;
        a = t + 1;
    }
    rec_layout(rec, fl[0], n_cig, l_seq, aux.l);
    memcpy(rec->qname, f[0], fl[0] + 1);
    rec->flag = (uint16_t)strtol(f[1], NULL, 0);
    rec->tid = (fl[2] == 1 && f[2][0] == '*') ? -1 : hdr_name2tid(r->hdr, f[2]);
    rec->pos = strtoll(f[3], NULL, 10) - 1;
    rec->mapq = (uint8_t)strtol(f[4], NULL, 10);
    {
        char *c = f[5]; size_t k = 0;
        while (k < n_cig) {
            char *q;
            unsigned long len = strtoul(c, &q, 10);
            int op;
            const char *ops = "MIDNSHP=XB";
            const char *o = strchr(ops, *q);
            if (!o || !*q) return -2;
            op = (int)(o - ops);
            rec->cigar[k++] = (uint32_t)(len << 4 | (unsigned)op);
            c = q + 1;
        }
    }
    if (fl[6] == 1 && f[6][0] == '=') rec->mtid = rec->tid;
    else if (fl[6] == 1 && f[6][0] == '*') rec->mtid = -1;
    else rec->mtid = hdr_name2tid(r->hdr, f[6]);
    rec->mpos = strtoll(f[7], NULL, 10) - 1;
    rec->isize = strtoll(f[8], NULL, 10);
    memset(rec->seq, 0, (l_seq + 1) / 2);
    for (size_t i = 0; i < l_seq; ++i)
        rec->seq[i >> 1] |= (uint8_t)(nt16_table[(unsigned char)f[9][i]] << ((~i & 1) << 2));
    if (fl[10] == 1 && f[10][0] == '*') memset(rec->qual, 0xff, l_seq);
    else {
        if (fl[10] != l_seq) { free(aux.s); return -2; }
        for (size_t i = 0; i < l_seq; ++i) rec->qual[i] = (uint8_t)(f[10][i] - 33);
    }
    if (aux.l) memcpy(rec->aux, aux.s, aux.l);
    free(aux.s);
    return 0;
}

static int parse_bam_rec(oreader_t *r, orec_t *rec)
{
    int32_t bs;
    int n = rd_bytes(r, &bs, 4);
    if (n == 0) return -1;
    if (n != 4 || bs < 32) return -2;
    if ((size_t)bs > r->m_blk) { r->m_blk = (size_t)bs * 2; r->blk = (uint8_t *)realloc(r->blk, r->m_blk); }
    if (rd_bytes(r, r->blk, (size_t)bs) != bs) return -2;
    const uint8_t *b = r->blk;
    int32_t refID, pos, l_seq, nref, npos, tlen; uint8_t l_rn, mapq; uint16_t n_cig, flag;
    memcpy(&refID, b, 4); memcpy(&pos, b + 4, 4); l_rn = b[8]; mapq = b[9];
    memcpy(&n_cig, b + 12, 2); memcpy(&flag, b + 14, 2); memcpy(&l_seq, b + 16, 4);
    memcpy(&nref, b + 20, 4); memcpy(&npos, b + 24, 4); memcpy(&tlen, b + 28, 4);
    size_t o = 32;
    size_t l_aux = (size_t)bs - 32 - l_rn - 4 * (size_t)n_cig - ((size_t)l_seq + 1) / 2 - (size_t)l_seq;
    if ((size_t)bs < 32 + l_rn + 4 * (size_t)n_cig + ((size_t)l_seq + 1) / 2 + (size_t)l_seq) return -2;
    rec_layout(rec, l_rn ? l_rn - 1u : 0u, n_cig, (size_t)l_seq, l_aux);
    memcpy(rec->qname, b + o, l_rn); rec->qname[l_rn ? l_rn - 1 : 0] = 0; o += l_rn;
    memcpy(rec->cigar, b + o, 4 * (size_t)n_cig); o += 4 * (size_t)n_cig;
    memcpy(rec->seq, b + o, ((size_t)l_seq + 1) / 2); o += ((size_t)l_seq + 1) / 2;
    memcpy(rec->qual, b + o, (size_t)l_seq); o += (size_t)l_seq;
    memcpy(rec->aux, b + o, l_aux);
    rec->tid = refID; rec->pos = pos; rec->mapq = mapq; rec->flag = flag;
    rec->mtid = nref; rec->mpos = npos; rec->isize = tlen;
    return 0;
}

int rd_next(oreader_t *r, orec_t *rec)
{
    for (;;) {
        int ret;
        if (r->is_bam) ret = parse_bam_rec(r, rec);
        else {
            if (r->have_line) r->have_line = 0;
            else {
                do { if (!rd_line(r, &r->line)) return -1; } while (r->line.l == 0);
            }
            ret = parse_sam_line(r, rec);
        }
        if (ret < 0) return ret;
        if (r->has_reg) {
            if (rec->tid != r->reg_tid) continue;
            if (rec->pos >= r->reg_end) continue;
            if (rec_endpos(rec) <= r->reg_beg) continue;
        }
        return 0;
    }
}

/* ---------------- region ---------------- */
/* HTSlib hts_parse_decimal: digits with thousands commas, optional fraction / exponent, suffixes k M G ("1M", "1.5k", "2,500,000") */
static long long parse_decimal(const char *s, char **endp)
{
    long long n = 0; int decimals = 0, e = 0, digits = 0, neg = 0;
    const char *p = s;
    while (isspace((unsigned char)*p)) ++p;
    if (*p == '-' && isdigit((unsigned char)p[1])) { neg = 1; ++p; } else if (*p == '+') ++p;
    for (; isdigit((unsigned char)*p) || (*p == ',' && digits); ++p) if (*p != ',') { if (n < LLONG_MAX / 10 - 1) n = n * 10 + (*p - '0'); digits = 1; }
    if (*p == '.') { ++p; for (; isdigit((unsigned char)*p); ++p) { if (n < LLONG_MAX / 10 - 1) { n = n * 10 + (*p - '0'); ++decimals; } digits = 1; } }
    if (!digits) { *endp = (char *)s; return 0; }
    if ((*p == 'e' || *p == 'E') && (isdigit((unsigned char)p[1]) || ((p[1] == '+' || p[1] == '-') && isdigit((unsigned char)p[2])))) { char *q; e = (int)strtol(p + 1, &q, 10); p = q; }
    if (*p == 'k' || *p == 'K') { e += 3; ++p; } else if (*p == 'm' || *p == 'M') { e += 6; ++p; } else if (*p == 'g' || *p == 'G') { e += 9; ++p; }
    e -= decimals;
    while (e > 0) { if (n < LLONG_MAX / 10 - 1) n *= 10; --e; }       /* (absurd coordinates saturate instead of overflowing) */
    while (e < 0) { n /= 10; ++e; }
    *endp = (char *)p;
    return neg ? -n : n;
}

int parse_region(const ohdr_t *h, const char *reg, int *tid, hpos_t *beg, hpos_t *end)
{
    int t = hdr_name2tid(h, reg);
    *beg = 0; *end = HPOS_MAX;
    if (t >= 0) { *tid = t; return 0; }
    const char *colon = strrchr(reg, ':');
    if (!colon) return -1;
    char *name = (char *)malloc((size_t)(colon - reg) + 1);
    memcpy(name, reg, (size_t)(colon - reg)); name[colon - reg] = 0;
    t = hdr_name2tid(h, name);
    free(name);
    if (t < 0) return -1;
    const char *num = colon + 1;
    char *q;
    long long b = parse_decimal(num, &q);
    if (q == num) {
        if (*q == '-') { b = 1; } else return -1;
    }
    if (b < 0 && !*q) { *tid = t; *beg = 0; *end = -b; return 0; }      /* hts_parse_region: chr:-100 is chr:1-100 */
    long long e = HPOS_MAX;
    if (*q == '-') { if (q[1]) { char *q2; e = parse_decimal(q + 1, &q2); if (q2 == q + 1 || *q2) return -1; } }
    else if (*q) return -1;
    *tid = t;
    *beg = b > 0 ? b - 1 : 0;
    *end = e;
    if (*beg >= *end) return -1;
    return 0;
}

/* ---------------- FASTA ---------------- */
ofasta_t *fa_load(const char *fn)
{
    gzFile fp = gzopen(fn, "rb");
    if (!fp) return NULL;
    ofasta_t *fa = (ofasta_t *)calloc(1, sizeof(*fa));
    size_t cap = 0, l = 0; char *cur = NULL;
    char *buf = (char *)malloc(1 << 16);
    int at_line_start = 1, in_name = 0;
    ostr_t nm = { 0, 0, NULL };
    int n;
    while ((n = gzread(fp, buf, 1 << 16)) > 0) {
        for (int i = 0; i < n; ++i) {
            char c = buf[i];
            if (in_name) {
                if (c == '\n') {
                    in_name = 0; at_line_start = 1;
                    /* name = up to first whitespace */
                    size_t k = 0; while (k < nm.l && !isspace((unsigned char)nm.s[k])) ++k;
                    fa->name = (char **)realloc(fa->name, sizeof(char *) * (size_t)(fa->n + 1));
                    fa->seq = (char **)realloc(fa->seq, sizeof(char *) * (size_t)(fa->n + 1));
                    fa->len = (hpos_t *)realloc(fa->len, sizeof(hpos_t) * (size_t)(fa->n + 1));
                    fa->name[fa->n] = (char *)malloc(k + 1);
                    memcpy(fa->name[fa->n], nm.s ? nm.s : "", k); fa->name[fa->n][k] = 0;
                    fa->n++;
                    cur = NULL; cap = l = 0;
                } else os_putc(&nm, c);
                continue;
            }
            if (at_line_start && c == '>') {
                if (fa->n) { fa->seq[fa->n - 1] = cur; fa->len[fa->n - 1] = (hpos_t)l; if (cur) cur[l] = 0; }
                in_name = 1; os_clear(&nm);
                continue;
            }
            if (c == '\n') { at_line_start = 1; continue; }
            at_line_start = 0;
            if (!isgraph((unsigned char)c)) continue;
            if (!fa->n) continue;
            if (l + 2 > cap) { cap = cap ? cap * 2 : 1024; cur = (char *)realloc(cur, cap); }
            cur[l++] = c;
        }
    }
    if (fa->n) {
        if (!cur) { cur = (char *)malloc(1); }
        cur[l] = 0;
        fa->seq[fa->n - 1] = cur; fa->len[fa->n - 1] = (hpos_t)l;
    }
    for (int i = 0; i < fa->n; ++i) if (!fa->seq[i]) { fa->seq[i] = strdup(""); fa->len[i] = 0; }
    free(buf); free(nm.s);
    gzclose(fp);
    return fa;
}
const char *fa_fetch(const ofasta_t *fa, const char *name, hpos_t *len)
{
    for (int i = 0; i < fa->n; ++i)
        if (strcmp(fa->name[i], name) == 0) { *len = fa->len[i]; return fa->seq[i]; }
    return NULL;
}
void fa_free(ofasta_t *fa)
{
    if (!fa) return;
    for (int i = 0; i < fa->n; ++i) { free(fa->name[i]); free(fa->seq[i]); }
    free(fa->name); free(fa->seq); free(fa->len); free(fa);
}

/* ---------------- BED (bedidx.c:102-197, 258-364) ---------------- */
#define LIDX_SHIFT 13
typedef struct { hpos_t beg, end; } bpair_t;
typedef struct {
    char *chr;
    int n, m;
    bpair_t *a;
    int *idx; hpos_t max_idx;
} breg_t;
struct obed { int n; breg_t *r; };

static int bpair_cmp(const void *x, const void *y)
{
    const bpair_t *a = (const bpair_t *)x, *b = (const bpair_t *)y;
    if (a->beg != b->beg) return a->beg < b->beg ? -1 : 1;
    return 0;
}

/* bedidx.c:102-140 bed_index_core */
static void bed_index_core(breg_t *p)
{
    int *idx = NULL; size_t m = 0; hpos_t last_end = 0;
    for (int i = 0; i < p->n; ++i) {
        hpos_t beg = p->a[i].beg >= 0 ? p->a[i].beg >> LIDX_SHIFT : 0;
        hpos_t end = p->a[i].end >= 0 ? p->a[i].end >> LIDX_SHIFT : 0;
        if (end < last_end) continue;
        if ((size_t)end + 1 > m) { m = ((size_t)end + 1) * 2; idx = (int *)realloc(idx, m * sizeof(int)); }
        hpos_t j;
        for (j = last_end; j < beg; j++) idx[j] = i > 0 ? i - 1 : 0;
        for (; j <= end; j++) idx[j] = i;
        last_end = end + 1;
    }
    p->idx = idx; p->max_idx = last_end;
}

obed_t *bed_load(const char *fn)
{
    gzFile fp = gzopen(fn, "rb");
    if (!fp) return NULL;
    obed_t *b = (obed_t *)calloc(1, sizeof(*b));
    char line[65536];
    while (gzgets(fp, line, sizeof(line))) {
        size_t l = strlen(line);
        while (l && (line[l - 1] == '\n' || line[l - 1] == '\r')) line[--l] = 0;
        if (!l) continue;
        char *ref = line;
        while (*ref && isspace((unsigned char)*ref)) ref++;
        if (!*ref || *ref == '#') continue;
        char *re = ref;
        while (*re && !isspace((unsigned char)*re)) re++;
        unsigned long long beg = 0, end = 0; int num = 0;
        if (*re) { *re = 0; num = sscanf(re + 1, "%llu %llu", &beg, &end); }
        if (num == 1) end = beg--;
        if (num < 1 || end < beg) {
            if (!strcmp(ref, "browser") || !strcmp(ref, "track")) continue;
            fprintf(stderr, "[bed_read] Parse error reading \"%s\"\n", fn);
            gzclose(fp); bed_free(b); errno = 0;
            return NULL;
        }
        breg_t *p = NULL;
        for (int i = 0; i < b->n; ++i) if (!strcmp(b->r[i].chr, ref)) { p = &b->r[i]; break; }
        if (!p) {
            b->r = (breg_t *)realloc(b->r, sizeof(breg_t) * (size_t)(b->n + 1));
            p = &b->r[b->n++];
            memset(p, 0, sizeof(*p));
            p->chr = strdup(ref);
        }
        if (p->n == p->m) { p->m = p->m ? p->m << 1 : 4; p->a = (bpair_t *)realloc(p->a, sizeof(bpair_t) * (size_t)p->m); }
        p->a[p->n].beg = (hpos_t)beg; p->a[p->n++].end = (hpos_t)end;
    }
    gzclose(fp);
    for (int i = 0; i < b->n; ++i) {
        /* ks_introsort on beg only; equal-beg order is irrelevant to overlap tests */
        qsort(b->r[i].a, (size_t)b->r[i].n, sizeof(bpair_t), bpair_cmp);
        bed_index_core(&b->r[i]);
    }
    return b;
}

/* bedidx.c:142-197 bed_minoff + bed_overlap_core + bed_overlap */
int bed_olap(const obed_t *b, const char *chr, hpos_t beg, hpos_t end)
{
    if (!b) return 0;
    const breg_t *p = NULL;
    for (int i = 0; i < b->n; ++i) if (!strcmp(b->r[i].chr, chr)) { p = &b->r[i]; break; }
    if (!p || p->n == 0) return 0;
    int min_off = 0;
    if (p->idx && p->max_idx > 0 && beg >= 0)
        min_off = (beg >> LIDX_SHIFT >= p->max_idx) ? p->idx[p->max_idx - 1] : p->idx[beg >> LIDX_SHIFT];
    for (int i = min_off; i < p->n; ++i) {
        if (p->a[i].beg >= end) break;
        if (p->a[i].end > beg && p->a[i].beg < end) return 1;
    }
    return 0;
}

void bed_free(obed_t *b)
{
    if (!b) return;
    for (int i = 0; i < b->n; ++i) { free(b->r[i].chr); free(b->r[i].a); free(b->r[i].idx); }
    free(b->r); free(b);
}

/* ---------------- flags (HTSlib bam_str2flag) ---------------- */
int str2flag(const char *s)
{
    char *end;
    long v = strtol(s, &end, 0);
    if (end != s && *end == 0) return v < 0 ? -1 : (int)v;
    static const struct { const char *n; int f; } names[] = {
        { "PAIRED", 1 }, { "PROPER_PAIR", 2 }, { "UNMAP", 4 }, { "MUNMAP", 8 }, { "REVERSE", 16 },
        { "MREVERSE", 32 }, { "READ1", 64 }, { "READ2", 128 }, { "SECONDARY", 256 },
        { "QCFAIL", 512 }, { "DUP", 1024 }, { "SUPPLEMENTARY", 2048 }, { NULL, 0 } };
    int flag = 0;
    const char *p = s;
    while (*p) {
        const char *e = p;
        while (*e && *e != ',') ++e;
        int hit = 0;
        for (int i = 0; names[i].n; ++i)
            if (strlen(names[i].n) == (size_t)(e - p) && strncasecmp(p, names[i].n, (size_t)(e - p)) == 0) { flag |= names[i].f; hit = 1; break; }
        if (!hit) return -1;
        p = *e ? e + 1 : e;
    }
    return flag;
}
