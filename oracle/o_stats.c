/*
 * oracle/o_stats.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * SURVEY.md 8(f) row 3, second half: the coverage distribution of `samtools stats` (the COV section), i.e. the pileup round
 * buffer of stats.c restated as it is -- quirks included, because the numbers depend on them:
 *   coverage_idx / round_buffer_lidx2ridx / round_buffer_flush / round_buffer_insert_read   stats.c:311-391
 *   the read filters in front of it and the CIGAR walk that feeds it                           stats.c:1212-1273, :1330-1340, :1380-1393, :1452-1508
 *   realloc_buffers' copy of the ring into a bigger one (byte counts where element counts were meant)   stats.c:690-692, :766-778
 *   bin setup and the printed lines                                                            stats.c:2396-2411, :1884-1892
 * The ring's aliasing rules that follow from that code (DESIGN.md section 7 spells them out for the device version):
 *   1. an aligned block [from, to) of a read at pos P lands in slots (from - P) mod size .. (to - P) mod size: reference
 *      positions further than `size` = 5 x (longest read so far, at least 300) behind the read's start fold back onto it;
 *   2. when the next read starts `size` or more positions later (or at a contig change / the end), all but the LAST slot is
 *      counted; what the last slot holds is added to the depth of the next read's first position (or, across a contig
 *      change, of the position its slot index maps to there; at the end of the input it is lost);
 *   3. when a read at least as long as the per-cycle arrays arrives, the ring is copied into a bigger one with memcpy(n) where
 *      n counts elements: only the first quarter of the pending positions (and of the wrapped part) survives.
 *   target regions: -t file (init_regions :1954-2043), region arguments (replicate_regions :2104-2149), is_in_regions :2067-2102,
 *   and the chunk-clipped CIGAR walk :1454-1487
 *   -p: remove_overlaps :1088-1210 (the part of a second mate that the first mate's blocks already cover is not counted) with the
 *   pair table's clean-up schedule :1392-1402 (it decides which line of a template counts as the first)
 * PINNING: the COV lines of test/stat/{1..8,11,12,14,15}*.expected and the `.large` variants (test/test.pl:3394-3429; every run
 * there that does not use -S): tests/test_stats_cov.py.
 *
 * Options restated: -c min,max,step  -f / -F flags  -d  -l readlen  -I read group or sample  -t targets  region arguments (the
 * reference reads those through the index: here the whole file is read and is_in_regions does the filtering, which gives the
 * same section)  -p; -r -q -i -m -x -s -g are accepted and have no effect on this section.  -S is refused.
 */
#include "o_plp.h"
#include <ctype.h>
#include <getopt.h>
#include <limits.h>

typedef struct { int32_t *buffer; int start, size; hpos_t pos; } rbuf_t;

typedef struct { hpos_t beg, end; } ival_t;                 /* 1-based, both ends included (stats.c regions_t.pos) */
typedef struct { int npos, mpos, cpos; ival_t *pos; } regs_t;

typedef struct {
    regs_t *regions; int nregions;            /* per tid; NULL = no target regions */
    ival_t *chunks; int nchunks, mchunks;     /* the regions the current read overlaps, clipped to it */
    int cov_min, cov_max, cov_step, ncov;
    uint64_t *cov;
    rbuf_t rb;
    int nbases, is_sorted, tid;
    hpos_t pos;
} cstat_t;

static int coverage_idx(int min, int max, int n, int step, int depth)
{
    if (depth < min) return 0;
    if (depth > max) return n - 1;
    return 1 + (depth - min) / step;
}

static int lidx2ridx(int offset, int size, hpos_t refpos, hpos_t pos) { return (int)((offset + (pos - refpos) % size) % size); }

static int rb_flush(cstat_t *s, hpos_t pos)
{
    int ibuf, idp;
    if (pos == s->rb.pos) return 0;
    hpos_t new_pos = pos;
    if (pos == -1 || pos - s->rb.pos >= s->rb.size) pos = s->rb.pos + s->rb.size - 1;      /* the whole buffer -- but for its last slot */
    if (pos < s->rb.pos) { fprintf(stderr, "Expected coordinates in ascending order, got %lld after %lld\n", (long long)pos, (long long)s->rb.pos); return -1; }
    int ifrom = s->rb.start;
    int ito = lidx2ridx(s->rb.start, s->rb.size, s->rb.pos, pos - 1);
    if (ifrom > ito) {
        for (ibuf = ifrom; ibuf < s->rb.size; ibuf++) {
            if (!s->rb.buffer[ibuf]) continue;
            idp = coverage_idx(s->cov_min, s->cov_max, s->ncov, s->cov_step, s->rb.buffer[ibuf]);
            s->cov[idp]++;
            s->rb.buffer[ibuf] = 0;
        }
        ifrom = 0;
    }
    for (ibuf = ifrom; ibuf <= ito; ibuf++) {
        if (!s->rb.buffer[ibuf]) continue;
        idp = coverage_idx(s->cov_min, s->cov_max, s->ncov, s->cov_step, s->rb.buffer[ibuf]);
        s->cov[idp]++;
        s->rb.buffer[ibuf] = 0;
    }
    s->rb.start = (new_pos == -1) ? 0 : lidx2ridx(s->rb.start, s->rb.size, s->rb.pos, pos);
    s->rb.pos = new_pos;
    return 0;
}

static int rb_insert(rbuf_t *rb, hpos_t from, hpos_t to)
{
    if (to - from > rb->size) { fprintf(stderr, "The read length too big (%lld), please increase the buffer length (currently %d)\n", (long long)(to - from), rb->size); return -1; }
    if (from < rb->pos) { fprintf(stderr, "The reads are not sorted (%lld comes after %lld).\n", (long long)from, (long long)rb->pos); return -1; }
    int ifrom = lidx2ridx(rb->start, rb->size, rb->pos, from), ito = lidx2ridx(rb->start, rb->size, rb->pos, to), ibuf;
    if (ifrom > ito) {
        for (ibuf = ifrom; ibuf < rb->size; ibuf++) rb->buffer[ibuf]++;
        ifrom = 0;
    }
    for (ibuf = ifrom; ibuf < ito; ibuf++) rb->buffer[ibuf]++;
    return 0;
}

/* stats.c:690-692 + :766-778 (the other per-cycle arrays the function grows do not touch this section) */
static void grow(cstat_t *s, int seq_len)
{
    int n = 2 * (1 + seq_len - s->nbases) + s->nbases;
    s->nbases = n;
    int32_t *rbuffer = (int32_t *)calloc(sizeof(int32_t), (size_t)seq_len * 5);
    n = s->rb.size - s->rb.start;
    memcpy(rbuffer, s->rb.buffer + s->rb.start, (size_t)n);                      /* n BYTES, as the reference has it */
    if (s->rb.start > 1) memcpy(rbuffer + n, s->rb.buffer, (size_t)s->rb.start);
    s->rb.start = 0;
    free(s->rb.buffer);
    s->rb.buffer = rbuffer;
    s->rb.size = seq_len * 5;
}

static int ival_lt(const void *a, const void *b)
{
    const ival_t *x = (const ival_t *)a, *y = (const ival_t *)b;
    return x->beg > y->beg ? 1 : x->beg < y->beg ? -1 : x->end > y->end ? 1 : x->end < y->end ? -1 : 0;
}

static void regs_add(cstat_t *s, int tid, hpos_t beg, hpos_t end)
{
    if (tid >= s->nregions) {
        s->regions = (regs_t *)realloc(s->regions, sizeof(regs_t) * (size_t)(tid + 10));
        memset(s->regions + s->nregions, 0, sizeof(regs_t) * (size_t)(tid + 10 - s->nregions));
        s->nregions = tid + 10;
    }
    regs_t *r = &s->regions[tid];
    if (r->npos >= r->mpos) { r->mpos = r->npos + 1000; r->pos = (ival_t *)realloc(r->pos, sizeof(ival_t) * (size_t)r->mpos); }
    r->pos[r->npos].beg = beg; r->pos[r->npos].end = end; r->npos++;
}

/* stats.c:2018-2031: per contig, sorted, overlapping intervals merged */
static void regs_finish(cstat_t *s)
{
    s->mchunks = 1;
    for (int t = 0; t < s->nregions; ++t) {
        regs_t *r = &s->regions[t];
        if (r->npos > 1) {
            qsort(r->pos, (size_t)r->npos, sizeof(ival_t), ival_lt);
            int n = 0;
            for (int p = 1; p < r->npos; ++p) {
                if (r->pos[n].end < r->pos[p].beg) r->pos[++n] = r->pos[p];
                else if (r->pos[n].end < r->pos[p].end) r->pos[n].end = r->pos[p].end;
            }
            r->npos = n + 1;
        }
        if (r->npos > s->mchunks) s->mchunks = r->npos;
    }
    s->chunks = (ival_t *)calloc((size_t)s->mchunks, sizeof(ival_t));
}

/* stats.c:1954-2016 */
static int regs_from_file(cstat_t *s, const ohdr_t *h, const char *file)
{
    FILE *fp = fopen(file, "r");
    if (!fp) { fprintf(stderr, "%s: cannot open\n", file); return -1; }
    char line[4096];
    int warned = 0, prev_tid = -1; hpos_t prev_pos = -1;
    while (fgets(line, sizeof line, fp)) {
        if (line[0] == '#') continue;
        size_t l = strlen(line);
        while (l && (line[l - 1] == '\n' || line[l - 1] == '\r')) line[--l] = 0;
        size_t i = 0;
        while (i < l && !isspace((unsigned char)line[i])) i++;
        if (i >= l) { fprintf(stderr, "Could not parse the file: %s [%s]\n", file, line); fclose(fp); return -1; }
        line[i] = 0;
        const int tid = hdr_name2tid(h, line);
        if (tid < 0) {
            if (!warned) fprintf(stderr, "Warning: Some sequences not present in the BAM, e.g. \"%s\". This message is printed only once.\n", line);
            warned = 1;
            continue;
        }
        long long b, e;
        if (sscanf(line + i + 1, "%lld %lld", &b, &e) != 2) { fprintf(stderr, "Could not parse the region [%s]\n", line + i + 1); fclose(fp); return -1; }
        if (prev_tid == -1 || prev_tid != tid) { prev_tid = tid; prev_pos = b; }
        if (prev_pos > b) { fprintf(stderr, "The positions are not in chromosomal order (%s:%lld comes after %lld)\n", line, b, (long long)prev_pos); fclose(fp); return -1; }
        regs_add(s, tid, b, e);
    }
    fclose(fp);
    if (!s->regions) { fprintf(stderr, "Unable to map the -t sequences to the BAM sequences.\n"); return -1; }
    regs_finish(s);
    return 0;
}

static hpos_t endpos_of(const orec_t *b)
{
    hpos_t rlen = 0;
    if (!(b->flag & F_UNMAP))
        for (uint32_t k = 0; k < b->n_cigar; ++k) { int op = cig_op(b->cigar[k]); if (op == C_M || op == C_D || op == C_N || op == C_EQ || op == C_X) rlen += cig_len(b->cigar[k]); }
    return b->pos + (rlen > 0 ? rlen : 1);
}

/* stats.c:2067-2102; 1 = counted (and s->chunks / nchunks set), 0 = not in a region, -1 = error */
static int is_in_regions(cstat_t *s, const orec_t *b)
{
    if (!s->regions) return 1;
    if (b->tid >= s->nregions || b->tid < 0) return 0;
    if (!s->is_sorted) { fprintf(stderr, "The BAM must be sorted in order for -t to work.\n"); return -1; }
    regs_t *reg = &s->regions[b->tid];
    if (reg->cpos == reg->npos) return 0;
    int i = reg->cpos;
    while (i < reg->npos && reg->pos[i].end <= b->pos) i++;
    if (i >= reg->npos) { reg->cpos = reg->npos; return 0; }
    const hpos_t endpos = endpos_of(b);
    if (endpos < reg->pos[i].beg) return 0;
    reg->cpos = i;
    s->nchunks = 0;
    while (i < reg->npos) {
        if (b->pos < reg->pos[i].end && endpos >= reg->pos[i].beg) {
            s->chunks[s->nchunks].beg = b->pos + 1 > reg->pos[i].beg ? b->pos + 1 : reg->pos[i].beg;
            s->chunks[s->nchunks].end = endpos < reg->pos[i].end ? endpos : reg->pos[i].end;
            s->nchunks++;
        }
        i++;
    }
    return 1;
}

/* ---- -p: the pair table (khash qn2pair in the reference; a chained table here) ---- */
typedef struct pair_s { char *name; ival_t *chunks; int n, m; unsigned first; struct pair_s *next; } pair_t;
#define PAIR_BUCKETS 65536
typedef struct { pair_t *b[PAIR_BUCKETS]; } pairtab_t;
static unsigned pair_hash(const char *s) { unsigned h = 2166136261u; while (*s) { h ^= (unsigned char)*s++; h *= 16777619u; } return h & (PAIR_BUCKETS - 1); }
static pair_t **pair_find(pairtab_t *t, const char *name) { pair_t **pp = &t->b[pair_hash(name)]; while (*pp && strcmp((*pp)->name, name)) pp = &(*pp)->next; return pp; }
static void pair_del(pair_t **pp) { pair_t *p = *pp; *pp = p->next; free(p->name); free(p->chunks); free(p); }

/* stats.c:1056-1085: entries whose last block ends before `max` */
static int cleanup_overlaps(pairtab_t *t, hpos_t max)
{
    int count = 0;
    for (int k = 0; k < PAIR_BUCKETS; ++k)
        for (pair_t **pp = &t->b[k]; *pp;) {
            if ((*pp)->chunks[(*pp)->n - 1].end < max) { pair_del(pp); count++; }
            else pp = &(*pp)->next;
        }
    return count;
}

/* stats.c:1088-1210; [pmin, pmax) 0-based half open; pmin == -1: the line is finished */
static int remove_overlaps(cstat_t *s, pairtab_t *t, unsigned *pair_count, const orec_t *b, hpos_t pmin, hpos_t pmax)
{
    const unsigned order = ((b->flag & 64) ? 1u : 0u) + ((b->flag & 128) ? 2u : 0u);
    long long isz = b->isize < 0 ? -(long long)b->isize : (long long)b->isize;
    if (!(b->flag & 1) || (b->flag & 8) || isz >= 2ll * b->l_qseq || (order != 1 && order != 2)) {
        if (pmin >= 0) return rb_insert(&s->rb, pmin, pmax);
        return 0;
    }
    pair_t **pp = pair_find(t, b->qname);
    if (!*pp) {
        if (pmin == -1) return 0;
        pair_t *pc = (pair_t *)calloc(1, sizeof(pair_t));
        pc->name = strdup(b->qname); pc->m = 8; pc->chunks = (ival_t *)calloc((size_t)pc->m, sizeof(ival_t));
        pc->chunks[0].beg = pmin; pc->chunks[0].end = pmax; pc->n = 1; pc->first = order;
        *pp = pc;
        (*pair_count)++;
    } else {
        pair_t *pc = *pp;
        if (order == pc->first) {
            if (pmin == -1) return 0;
            if (pc->n == pc->m) { pc->m <<= 1; pc->chunks = (ival_t *)realloc(pc->chunks, sizeof(ival_t) * (size_t)pc->m); }
            pc->chunks[pc->n].beg = pmin; pc->chunks[pc->n].end = pmax; pc->n++;
        } else {
            if (pmin == -1) { pair_del(pp); (*pair_count)--; return 0; }
            for (int i = 0; i < pc->n; i++) {
                if (pmin >= pc->chunks[i].end) continue;
                if (pmax <= pc->chunks[i].beg) break;
                if (pmin < pc->chunks[i].beg) { if (rb_insert(&s->rb, pmin, pc->chunks[i].beg) < 0) return -1; pmin = pc->chunks[i].beg; }
                if (pmax <= pc->chunks[i].end) return 0;
                pmin = pc->chunks[i].end;
            }
        }
    }
    return rb_insert(&s->rb, pmin, pmax);
}

static int unclipped_length(const orec_t *b)
{
    int len = b->l_qseq;
    for (uint32_t k = 0; k < b->n_cigar; ++k) if (cig_op(b->cigar[k]) == C_H) len += (int)cig_len(b->cigar[k]);
    return len;
}

int o_main_stats(int argc, char *argv[])
{
    int c, flag_require = 0, flag_filter = 0, filter_readlen = -1, tmp, remove_olap = 0;
    static pairtab_t pairs; unsigned pair_count = 0, last_read_flush = 0; int last_pair_tid = -2;
    const char *group_id = NULL, *targets = NULL;
    cstat_t st; memset(&st, 0, sizeof st);
    st.cov_min = 1; st.cov_max = 1000; st.cov_step = 1;
    static const struct option lopts[] = {
        { "coverage", required_argument, NULL, 'c' }, { "required-flag", required_argument, NULL, 'f' }, { "filtering-flag", required_argument, NULL, 'F' },
        { "remove-dups", no_argument, NULL, 'd' }, { "read-length", required_argument, NULL, 'l' }, { "id", required_argument, NULL, 'I' },
        { "ref-seq", required_argument, NULL, 'r' }, { "insert-size", required_argument, NULL, 'i' }, { "most-inserts", required_argument, NULL, 'm' },
        { "trim-quality", required_argument, NULL, 'q' }, { "sparse", no_argument, NULL, 'x' }, { "sam", no_argument, NULL, 's' },
        { "target-regions", required_argument, NULL, 't' }, { "cov-threshold", required_argument, NULL, 'g' }, { NULL, 0, NULL, 0 } };
    optind = 1;
    while ((c = getopt_long(argc, argv, "dsxr:c:l:i:m:q:f:F:I:t:g:pS:", lopts, NULL)) >= 0) {
        switch (c) {
        case 'f': if ((tmp = str2flag(optarg)) < 0) { fprintf(stderr, "samtools stats: Unknown flag '%s'\n", optarg); return 1; } flag_require = tmp; break;
        case 'F': if ((tmp = str2flag(optarg)) < 0) { fprintf(stderr, "samtools stats: Unknown flag '%s'\n", optarg); return 1; } flag_filter |= tmp; break;
        case 'd': flag_filter |= F_DUP; break;
        case 'c': if (sscanf(optarg, "%d,%d,%d", &st.cov_min, &st.cov_max, &st.cov_step) != 3) { fprintf(stderr, "Unable to parse -c %s\n", optarg); return 1; } break;
        case 'l': filter_readlen = atoi(optarg); break;
        case 'I': group_id = optarg; break;
        case 't': targets = optarg; break;
        case 'p': remove_olap = 1; break;
        case 'r': case 'i': case 'm': case 'q': case 'x': case 's': case 'g': break;
        default: fprintf(stderr, "[stats] option -%c is not part of the restated section (COV)\n", c); return 1;
        }
    }
    if (argc - optind < 1) { fprintf(stderr, "usage: oracle_samtools stats [-c min,max,step] [-f INT] [-F INT] [-d] [-l INT] [-I ID] [-t targets] in.bam [region ...]\n"); return 1; }
    oreader_t *rd = rd_open(argv[optind]);
    if (!rd) { fprintf(stderr, "samtools stats: failed to open \"%s\"\n", argv[optind]); return 1; }
    ohdr_t *h = rd_header(rd);
    /* stats.c:2396-2411 */
    if (st.cov_step > st.cov_max - st.cov_min + 1) { st.cov_step = st.cov_max - st.cov_min; if (st.cov_step <= 0) st.cov_step = 1; }
    st.ncov = 3 + (st.cov_max - st.cov_min) / st.cov_step;
    st.cov_max = st.cov_min + ((st.cov_max - st.cov_min) / st.cov_step + 1) * st.cov_step - 1;
    st.cov = (uint64_t *)calloc(sizeof(uint64_t), (size_t)st.ncov);
    st.nbases = 300;
    st.rb.size = st.nbases * 5;
    st.rb.buffer = (int32_t *)calloc(sizeof(int32_t), (size_t)st.rb.size);
    st.is_sorted = 1; st.tid = -1; st.pos = -1;
    if (targets) { if (regs_from_file(&st, h, targets) < 0) return 1; }
    else if (argc - optind > 1) {
        /* stats.c:2104-2149: the regions of the command line (the reference gets them, merged, from the multi-region iterator) */
        for (int a = optind + 1; a < argc; ++a) {
            int t; hpos_t rb, re;
            if (parse_region(h, argv[a], &t, &rb, &re) < 0) { fprintf(stderr, "Multi-region iterator could not be created\n"); return 1; }
            regs_add(&st, t, rb + 1, re);
        }
        regs_finish(&st);
    }
    /* stats.c:2151-2177: the read groups whose ID or SM is `group_id` */
    char **rg_ok = NULL; int n_rg_ok = 0;
    if (group_id) {
        for (const char *l = h->text; l && *l; ) {
            const char *e = strchr(l, '\n'); size_t n = e ? (size_t)(e - l) : strlen(l);
            if (n > 4 && strncmp(l, "@RG\t", 4) == 0) {
                char *line = strndup(l, n), *id = NULL, *sm = NULL, *save = NULL;
                for (char *f = strtok_r(line + 4, "\t", &save); f; f = strtok_r(NULL, "\t", &save)) {
                    if (strncmp(f, "ID:", 3) == 0 && !id) id = f + 3;
                    if (strncmp(f, "SM:", 3) == 0 && !sm) sm = f + 3;
                }
                if (id && (strcmp(id, group_id) == 0 || (sm && strcmp(sm, group_id) == 0))) {
                    rg_ok = (char **)realloc(rg_ok, sizeof(char *) * (size_t)(n_rg_ok + 1)); rg_ok[n_rg_ok++] = strdup(id);
                }
                free(line);
            }
            l = e ? e + 1 : NULL;
        }
    }
    orec_t b; memset(&b, 0, sizeof b);
    int r, ret = 0;
    while ((r = rd_next(rd, &b)) >= 0) {
        /* stats.c:1212-1273 */
        { const int in = is_in_regions(&st, &b); if (in < 0) { ret = 1; break; } if (!in) continue; }
        if (group_id) {
            const uint8_t *rg = rec_aux_get(&b, "RG");
            if (!rg) continue;
            int ok = 0;
            for (int k = 0; k < n_rg_ok; ++k) if (strcmp(rg_ok[k], (const char *)rg + 1) == 0) { ok = 1; break; }
            if (!ok) continue;
        }
        if (flag_require && (b.flag & flag_require) != flag_require) continue;
        if (flag_filter && (b.flag & flag_filter)) continue;
        if (filter_readlen != -1 && b.l_qseq != filter_readlen) continue;
        if (b.flag & F_SECONDARY) continue;
        if (!b.l_qseq) continue;
        int read_len = unclipped_length(&b);
        if (read_len >= st.nbases) grow(&st, read_len);
        if (b.flag & F_UNMAP) continue;
        if (b.n_cigar == 0) { fprintf(stderr, "FIXME: mapped read with no cigar?\n"); ret = 1; break; }
        /* stats.c:1380-1393 */
        if (st.tid == b.tid && b.pos < st.pos) st.is_sorted = 0;
        st.pos = b.pos;
        if (!st.is_sorted) continue;
        if (st.tid == -1 || st.tid != b.tid) { if (rb_flush(&st, -1) < 0) { ret = 1; break; } }
        /* stats.c:1392-1402: the pair table is thinned out every 10 000 reads once it holds 10 000 pairs, and emptied at a contig change */
        last_read_flush++;
        if (pair_count > 10000 && last_read_flush > 10000) { pair_count -= (unsigned)cleanup_overlaps(&pairs, b.pos); last_read_flush = 0; }
        if (last_pair_tid != b.tid) { pair_count -= (unsigned)cleanup_overlaps(&pairs, HPOS_MAX - 1); last_pair_tid = b.tid; last_read_flush = 0; }
        st.tid = b.tid;
        /* stats.c:1452-1508 */
        if (rb_flush(&st, b.pos) < 0) { ret = 1; break; }
        hpos_t p = b.pos;
        int bad = 0;
        if (st.regions) {
            /* stats.c:1454-1487: every aligned block clipped to the chunks; a block that reaches beyond a chunk is looked at again with the next */
            uint32_t j = 0; int i = 0;
            while (j < b.n_cigar && i < st.nchunks) {
                int op = cig_op(b.cigar[j]), oplen = (int)cig_len(b.cigar[j]);
                if (op == C_M || op == C_EQ || op == C_X) {
                    hpos_t pmin = p > st.chunks[i].beg - 1 ? p : st.chunks[i].beg - 1, pmax = p + oplen < st.chunks[i].end ? p + oplen : st.chunks[i].end;
                    if (pmax > pmin && (remove_olap ? remove_overlaps(&st, &pairs, &pair_count, &b, pmin, pmax) : rb_insert(&st.rb, pmin, pmax)) < 0) { bad = 1; break; }
                }
                hpos_t pnew = p + ((op == C_M || op == C_D || op == C_N || op == C_EQ || op == C_X) ? oplen : 0);
                if (pnew >= st.chunks[i].end) i++;
                else { j++; p = pnew; }
            }
        } else
        for (uint32_t j = 0; j < b.n_cigar; ++j) {
            int op = cig_op(b.cigar[j]), oplen = (int)cig_len(b.cigar[j]);
            if (op == C_M || op == C_EQ || op == C_X) { if ((remove_olap ? remove_overlaps(&st, &pairs, &pair_count, &b, p, p + oplen) : rb_insert(&st.rb, p, p + oplen)) < 0) { bad = 1; break; } }
            if (op == C_M || op == C_D || op == C_N || op == C_EQ || op == C_X) p += oplen;
        }
        if (bad) { ret = 1; break; }
        if (remove_olap) remove_overlaps(&st, &pairs, &pair_count, &b, -1, -1);        /* the line is finished (stats.c:1509-1510) */
    }
    if (r < -1) { fprintf(stderr, "Failure while decoding file\n"); ret = 1; }
    if (!ret) {
        rb_flush(&st, -1);
        if (st.is_sorted) {
            /* stats.c:1884-1892 */
            printf("# Coverage distribution. Use `grep ^COV | cut -f 2-` to extract this part.\n");
            if (st.cov[0]) printf("COV\t[<%d]\t%d\t%ld\n", st.cov_min, st.cov_min - 1, (long)st.cov[0]);
            for (int i = 1; i < st.ncov - 1; i++)
                if (st.cov[i]) printf("COV\t[%d-%d]\t%d\t%ld\n", st.cov_min + (i - 1) * st.cov_step, st.cov_min + i * st.cov_step - 1, st.cov_min + i * st.cov_step - 1, (long)st.cov[i]);
            if (st.cov[st.ncov - 1]) printf("COV\t[%d<]\t%d\t%ld\n", st.cov_min + (st.ncov - 2) * st.cov_step - 1, st.cov_min + (st.ncov - 2) * st.cov_step - 1, (long)st.cov[st.ncov - 1]);
        }
    }
    rec_free(&b); free(st.cov); free(st.rb.buffer);
    for (int k = 0; k < n_rg_ok; ++k) free(rg_ok[k]);
    free(rg_ok);
    rd_close(rd);
    return ret;
}
