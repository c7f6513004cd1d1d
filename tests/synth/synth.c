/*
 * synth.c -- seeded synthetic alignment generator (see include/crumble_synth.h).
 * Bench/test utility: it stands in for the BAM files named by BASELINE.json's
 * configs, none of which exist offline.  Pure host C + OpenMP.
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <omp.h>
#include "crumble_synth.h"

#define WIN 16384            /* start positions per generation window */

static inline uint64_t mix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
static inline uint64_t h3(uint64_t seed, uint64_t a, uint64_t b)
{ return mix64(mix64(seed ^ (a * 0xD6E8FEB86659FD93ULL)) + b); }
static inline double u01(uint64_t h) { return (h >> 11) * (1.0 / 9007199254740992.0); }

typedef struct { uint64_t s; } rng_t;
static inline uint64_t rnd(rng_t *r) { r->s += 0x9E3779B97F4A7C15ULL; return mix64(r->s); }
static inline double rndu(rng_t *r) { return u01(rnd(r)); }
static inline int rndi(rng_t *r, int n) { return (int)(rnd(r) % (uint64_t)n); }

struct crumble_synth {
    crumble_batch b;
    int32_t *pos; uint8_t *mapq; uint16_t *flag;
    int64_t *seq_off, *cig_off; uint32_t *cigar; uint8_t *bases, *quals;
};

typedef struct {
    int64_t n_reads, n_cig, n_bases, cap_reads, cap_cig, cap_bases;
    int32_t *pos; uint8_t *mapq; uint16_t *flag; int32_t *lq, *ncig;
    uint32_t *cigar; uint8_t *bases, *quals;
} win_t;

static void win_reserve(win_t *w, int reads, int cig, int bases)
{
    if (w->n_reads + reads > w->cap_reads) {
        w->cap_reads = (w->n_reads + reads) * 2 + 64;
        w->pos = realloc(w->pos, w->cap_reads * 4); w->mapq = realloc(w->mapq, w->cap_reads);
        w->flag = realloc(w->flag, w->cap_reads * 2); w->lq = realloc(w->lq, w->cap_reads * 4);
        w->ncig = realloc(w->ncig, w->cap_reads * 4);
    }
    if (w->n_cig + cig > w->cap_cig) {
        w->cap_cig = (w->n_cig + cig) * 2 + 64;
        w->cigar = realloc(w->cigar, w->cap_cig * 4);
    }
    if (w->n_bases + bases > w->cap_bases) {
        w->cap_bases = (w->n_bases + bases) * 2 + 4096;
        w->bases = realloc(w->bases, w->cap_bases); w->quals = realloc(w->quals, w->cap_bases);
    }
}

typedef struct {
    const crumble_synth_cfg *c;
    uint8_t *ref;     /* 0..3 */
    uint8_t *vtype;   /* 0 none 1 hetSNP 2 homSNP 3 hetDel 4 homDel 5 hetIns 6 homIns */
    uint8_t *vlen;    /* SNP: alt shift 1..3; indel: length */
    uint8_t *instr;   /* STR period at this position (0 outside) */
    int64_t glen;
} genome_t;

static const uint8_t NT16[4] = {1, 2, 4, 8};
static double pe_tab[128];


static void build_genome(genome_t *g)
{
    const crumble_synth_cfg *c = g->c;
    int64_t L = g->glen;
    int64_t blk, nblk;
    #pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < L; p++) {
        g->ref[p] = (uint8_t)(h3(c->seed, 1, p) & 3);
        g->instr[p] = 0;
    }
    if (c->str_every > 250) {
        nblk = L / c->str_every;
        #pragma omp parallel for schedule(static)
        for (blk = 0; blk < nblk; blk++) {
            uint64_t h = h3(c->seed, 2, blk);
            int period = 1 + (int)(h % 6);
            int copies = 5 + (int)((h >> 8) % 26);
            int64_t start = blk * c->str_every + (int64_t)((h >> 20) % (uint64_t)(c->str_every - 200));
            uint8_t unit[6];
            int k, same = 1;
            for (k = 0; k < period; k++) unit[k] = (uint8_t)((h >> (32 + 2 * k)) & 3);
            for (k = 1; k < period; k++) if (unit[k] != unit[0]) same = 0;
            if (period > 1 && same) unit[period - 1] = (unit[0] + 1) & 3;
            for (k = 0; k < period * copies && start + k < L; k++) {
                g->ref[start + k] = unit[k % period];
                g->instr[start + k] = (uint8_t)period;
            }
        }
    }
    #pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < L; p++) {
        uint64_t h = h3(c->seed, 3, p);
        double u = u01(h);
        double ind = c->indel_var * (g->instr[p] ? 30.0 : 1.0);
        g->vtype[p] = 0; g->vlen[p] = 0;
        if (u < c->het_snp) { g->vtype[p] = 1; g->vlen[p] = 1 + (uint8_t)((h >> 3) % 3); }
        else if (u < c->het_snp + c->hom_snp) { g->vtype[p] = 2; g->vlen[p] = 1 + (uint8_t)((h >> 3) % 3); }
        else if (u < c->het_snp + c->hom_snp + ind) {
            int hom = (h >> 5) & 1, ins = (h >> 6) & 1, len;
            if (g->instr[p]) len = g->instr[p];
            else { len = 1; while (len < 8 && ((h >> (7 + len)) & 1)) len++; }
            g->vtype[p] = (uint8_t)(3 + 2 * ins + hom);
            g->vlen[p] = (uint8_t)len;
        }
    }
}

static inline void push_op(uint32_t *ops, int *nops, int op, int len)
{
    if (len <= 0) return;
    if (*nops && (int)(ops[*nops - 1] & 0xf) == op) ops[*nops - 1] += (uint32_t)len << 4;
    else ops[(*nops)++] = ((uint32_t)len << 4) | (uint32_t)op;
}

static inline uint8_t gen_qual(const crumble_synth_cfg *c, rng_t *r, double t)
{
    double u = rndu(r);
    if (c->qual_model == 1) return (uint8_t)(20 + rndi(r, 22));
    if (c->qual_model == 2) {
        if (u < 0.80) return 93;
        if (u < 0.95) return (uint8_t)(40 + rndi(r, 53));
        return (uint8_t)(1 + rndi(r, 60));
    }
    {
        double t2 = t * t;
        double w37 = 0.88 - 0.45 * t2, w25 = 0.08 + 0.25 * t2, w11 = 0.03 + 0.12 * t2;
        if (u < w37) return 37;
        if (u < w37 + w25) return 25;
        if (u < w37 + w25 + w11) return 11;
        return 2;
    }
}

/* One read starting at reference position s; appended to w. */
static void gen_read(const genome_t *g, win_t *w, rng_t *r, int64_t s)
{
    const crumble_synth_cfg *c = g->c;
    int L = c->read_len, hap = (int)(rnd(r) & 1), rev = (int)(rnd(r) & 1);
    int clip_l = 0, clip_r = 0, skip_at = -1, skip_len = 0;
    int q = 0, nops = 0, i, mrun = 0;
    int64_t p = s;
    uint32_t ops[4096];
    uint8_t *bs, *qs;
    uint16_t flag;
    uint8_t mapq;

    if (c->read_len_sigma > 0) {
        /* sum of uniforms ~ normal; log-normal length */
        double z = (rndu(r) + rndu(r) + rndu(r) + rndu(r) - 2.0) * 1.7320508;
        double f = exp(z * c->read_len_sigma / 100.0);
        if (f < 0.5) f = 0.5;
        if (f > 2.0) f = 2.0;
        L = (int)(L * f);
    }
    if (L < 30) L = 30;
    if (s + 2 * (int64_t)L + 1024 >= g->glen) return;
    if (rndu(r) < c->softclip_frac) {
        int cl = 5 + rndi(r, 16);
        if (rnd(r) & 1) clip_l = cl; else clip_r = cl;
    }
    if (rndu(r) < c->refskip_frac && L >= 60) {
        skip_at = 20 + rndi(r, L - 40);
        skip_len = 50 + rndi(r, 451);
    }
    win_reserve(w, 1, 4096, L + 16);
    bs = w->bases + w->n_bases; qs = w->quals + w->n_bases;

    if (rndu(r) < c->unmapped_frac) {          /* unmapped mate: no cigar */
        for (i = 0; i < L; i++) { bs[i] = NT16[rndi(r, 4)]; qs[i] = gen_qual(c, r, (double)i / L); }
        w->pos[w->n_reads] = (int32_t)s; w->mapq[w->n_reads] = 0;
        w->flag[w->n_reads] = 0x1 | 0x4 | (rev ? 0x10 : 0x20) | 0x80;
        w->lq[w->n_reads] = L; w->ncig[w->n_reads] = 0;
        w->n_reads++; w->n_bases += L;
        return;
    }

    for (i = 0; i < clip_l; i++) { bs[q] = NT16[rndi(r, 4)]; q++; }
    push_op(ops, &nops, 4, clip_l);
    while (q < L - clip_r && nops < 4000) {
        int vt = g->vtype[p], applies, safe;
        uint8_t base;
        applies = vt && (!(vt & 1) || hap);          /* hom (even code) or het on hap 1 */
        safe = (mrun >= 3 && q < L - clip_r - 12);
        if (q == skip_at && safe) {                  /* reference skip (N) */
            push_op(ops, &nops, 3, skip_len); p += skip_len; mrun = 0; skip_at = -1;
            continue;
        }
        if (applies && (vt == 3 || vt == 4) && safe) {       /* deletion variant */
            push_op(ops, &nops, 2, g->vlen[p]); p += g->vlen[p]; mrun = 0;
            continue;
        }
        if (safe && rndu(r) < c->err_indel) {               /* sequencing indel */
            if (rnd(r) & 1) { push_op(ops, &nops, 2, 1); p += 1; mrun = 0; continue; }
            bs[q] = (c->qual_model == 2 && q) ? bs[q - 1] : NT16[rndi(r, 4)];
            q++; push_op(ops, &nops, 1, 1); mrun = 0;
            continue;
        }
        base = g->ref[p];
        if (applies && (vt == 1 || vt == 2)) base = (base + g->vlen[p]) & 3;
        bs[q++] = NT16[base]; p++; mrun++;
        push_op(ops, &nops, 0, 1);
        if (applies && (vt == 5 || vt == 6) && safe) {       /* insertion variant */
            int k, il = g->vlen[p - 1];
            for (k = 0; k < il && q < L - clip_r - 4; k++) {
                uint8_t ib = g->instr[p - 1] ? g->ref[p + k] : (uint8_t)(h3(c->seed, 4, (p - 1) * 8 + k) & 3);
                bs[q++] = NT16[ib];
                push_op(ops, &nops, 1, 1);
            }
            mrun = 0;
        }
    }
    for (i = 0; i < clip_r; i++) { bs[q] = NT16[rndi(r, 4)]; q++; }
    push_op(ops, &nops, 4, clip_r);
    L = q;

    for (i = 0; i < L; i++) {                         /* qualities + substitution errors */
        double t = rev ? (double)(L - 1 - i) / L : (double)i / L;
        uint8_t ql = gen_qual(c, r, t);
        uint64_t h = rnd(r);
        qs[i] = ql;
        if (u01(h) < pe_tab[ql]) {
            int k = 0; while (k < 3 && NT16[k] != bs[i]) k++;
            bs[i] = NT16[(k + 1 + (int)((h >> 3) % 3)) & 3];
        }
        if ((h & 0x1fff) == 0x1abc && (h >> 60) < 2) bs[i] = 15;   /* the odd N (~1.5e-5) */
    }
    {
        uint64_t hb = h3(c->seed, 5, (uint64_t)(s / 50000));
        double u = rndu(r);
        if (u01(hb) < c->mapq0_frac) mapq = 0;
        else if (u < c->mapq_mid_frac) mapq = (uint8_t)(1 + rndi(r, 59));
        else mapq = 60;
    }
    flag = 0x1 | 0x2 | (rev ? 0x10 : 0x20) | ((rnd(r) & 1) ? 0x40 : 0x80);
    memcpy(w->cigar + w->n_cig, ops, nops * 4);
    w->pos[w->n_reads] = (int32_t)s; w->mapq[w->n_reads] = mapq; w->flag[w->n_reads] = flag;
    w->lq[w->n_reads] = L; w->ncig[w->n_reads] = nops;
    w->n_reads++; w->n_cig += nops; w->n_bases += L;
}

static int poisson(rng_t *r, double lam)
{
    if (lam <= 0) return 0;
    if (lam < 30) {
        double l = exp(-lam), p = 1.0; int k = 0;
        do { k++; p *= rndu(r); } while (p > l);
        return k - 1;
    }
    {   /* normal approximation for the ultra-deep spike */
        double z = (rndu(r) + rndu(r) + rndu(r) + rndu(r) - 2.0) * 1.7320508;
        int k = (int)(lam + z * sqrt(lam) + 0.5);
        return k < 0 ? 0 : k;
    }
}

int crumble_synth_config(crumble_synth_cfg *c, const char *name, double scale)
{
    memset(c, 0, sizeof(*c));
    c->spike_pos = -1;
    c->het_snp = 1e-3; c->hom_snp = 5e-4; c->indel_var = 1e-4; c->err_indel = 1e-4;
    c->str_every = 2000; c->mapq0_frac = 0.05; c->mapq_mid_frac = 0.03;
    c->softclip_frac = 0.01; c->unmapped_frac = 0.002; c->refskip_frac = 0.0;
    c->depth_sigma = 0.15;
    if (scale <= 0) scale = 1.0;
    if (!strcmp(name, "chr20")) {          /* config 2: NA12878 chr20 30x Illumina shape */
        c->seed = 20; c->contig_len = (int32_t)(63025520 * scale); c->read_len = 150;
        c->depth = 30; c->qual_model = 0;
    } else if (!strcmp(name, "ce")) {      /* config 1: htslib test/ce.bam stand-in */
        c->seed = 1; c->contig_len = (int32_t)(100000 * scale); c->read_len = 100;
        c->depth = 1.0; c->qual_model = 1; c->softclip_frac = 0.05; c->err_indel = 2e-4;
        c->mapq0_frac = 0.1; c->mapq_mid_frac = 0.2; c->refskip_frac = 0.01;
    } else if (!strcmp(name, "panel")) {   /* config 3: 300x exome panel */
        c->seed = 300; c->contig_len = (int32_t)(20000000 * scale); c->read_len = 150;
        c->depth = 300; c->depth_sigma = 0.2; c->qual_model = 1;
        c->target_len = 1000; c->target_every = 10000;
        c->spike_pos = 5500; c->spike_depth = 6000;
    } else if (!strcmp(name, "hifi")) {    /* config 4: PacBio HiFi 20x */
        c->seed = 4; c->contig_len = (int32_t)(248956422 * scale); c->read_len = 15000;
        c->read_len_sigma = 25; c->depth = 20; c->qual_model = 2; c->err_indel = 5e-4;
        c->softclip_frac = 0.002; c->unmapped_frac = 0.0;
    } else return CRUMBLE_EINVAL;
    if (c->contig_len < 4 * c->read_len + 4096) c->contig_len = 4 * c->read_len + 4096;
    return CRUMBLE_OK;
}

crumble_synth *crumble_synth_generate(const crumble_synth_cfg *c, int n_threads)
{
    genome_t g;
    crumble_synth *s = calloc(1, sizeof(*s));
    int64_t nwin, wi, tr = 0, tc = 0, tb = 0;
    win_t *wins;
    int64_t *ro, *co, *bo;

    if (!s) return NULL;
    if (n_threads > 0) omp_set_num_threads(n_threads);
    g.c = c; g.glen = (int64_t)c->contig_len + 4 * (int64_t)c->read_len + 8192;
    g.ref = malloc(g.glen); g.vtype = malloc(g.glen); g.vlen = malloc(g.glen); g.instr = malloc(g.glen);
    if (!g.ref || !g.vtype || !g.vlen || !g.instr) return NULL;
    for (int q = 0; q < 128; q++) pe_tab[q] = pow(10.0, -q / 10.0);
    build_genome(&g);

    nwin = (c->contig_len + WIN - 1) / WIN;
    wins = calloc(nwin, sizeof(*wins));
    #pragma omp parallel for schedule(dynamic, 4)
    for (wi = 0; wi < nwin; wi++) {
        rng_t r = { h3(c->seed, 6, wi) };
        int64_t p0 = wi * WIN, p1 = p0 + WIN, p;
        if (p1 > c->contig_len) p1 = c->contig_len;
        for (p = p0; p < p1; p++) {
            double lam = c->depth / c->read_len, f;
            int k;
            uint64_t hb = h3(c->seed, 7, (uint64_t)(p / 10000));
            f = 1.0 + c->depth_sigma * (u01(hb) + u01(hb * 31 + 7) - 1.0) * 2.0;
            lam *= f;
            if (c->target_len > 0) {
                int64_t t0 = (p / c->target_every) * c->target_every + c->target_every / 2;
                int64_t mid = p + c->read_len / 2;
                if (!(mid >= t0 && mid < t0 + c->target_len)) lam = 0;
            }
            if (c->spike_pos >= 0 && p > c->spike_pos - c->read_len + 8 && p <= c->spike_pos - 8)
                lam += (double)c->spike_depth / (c->read_len - 16);
            k = poisson(&r, lam);
            while (k-- > 0) gen_read(&g, &wins[wi], &r, p);
        }
    }
    ro = malloc((nwin + 1) * 8); co = malloc((nwin + 1) * 8); bo = malloc((nwin + 1) * 8);
    for (wi = 0; wi < nwin; wi++) {
        ro[wi] = tr; co[wi] = tc; bo[wi] = tb;
        tr += wins[wi].n_reads; tc += wins[wi].n_cig; tb += wins[wi].n_bases;
    }
    s->pos = malloc((tr + 1) * 4); s->mapq = malloc(tr + 1); s->flag = malloc((tr + 1) * 2);
    s->seq_off = malloc((tr + 1) * 8); s->cig_off = malloc((tr + 1) * 8);
    s->cigar = malloc((tc + 1) * 4); s->bases = malloc(tb + 1); s->quals = malloc(tb + 1);
    #pragma omp parallel for schedule(dynamic, 4)
    for (wi = 0; wi < nwin; wi++) {
        win_t *w = &wins[wi];
        int64_t i, so = bo[wi], cg = co[wi];
        memcpy(s->pos + ro[wi], w->pos, w->n_reads * 4);
        memcpy(s->mapq + ro[wi], w->mapq, w->n_reads);
        memcpy(s->flag + ro[wi], w->flag, w->n_reads * 2);
        memcpy(s->cigar + co[wi], w->cigar, w->n_cig * 4);
        memcpy(s->bases + bo[wi], w->bases, w->n_bases);
        memcpy(s->quals + bo[wi], w->quals, w->n_bases);
        for (i = 0; i < w->n_reads; i++) {
            s->seq_off[ro[wi] + i] = so; s->cig_off[ro[wi] + i] = cg;
            so += w->lq[i]; cg += w->ncig[i];
        }
        free(w->pos); free(w->mapq); free(w->flag); free(w->lq); free(w->ncig);
        free(w->cigar); free(w->bases); free(w->quals);
    }
    s->seq_off[tr] = tb; s->cig_off[tr] = tc;
    free(wins); free(ro); free(co); free(bo);
    free(g.ref); free(g.vtype); free(g.vlen); free(g.instr);

    s->b.tid = c->tid; s->b.n_reads = tr;
    s->b.pos = s->pos; s->b.mapq = s->mapq; s->b.flag = s->flag;
    s->b.seq_off = s->seq_off; s->b.cig_off = s->cig_off; s->b.cigar = s->cigar;
    s->b.bases = s->bases; s->b.quals = s->quals;
    s->b.n_bed = 0; s->b.bed_start = NULL; s->b.bed_end = NULL;
    return s;
}

const crumble_batch *crumble_synth_batch(const crumble_synth *s) { return &s->b; }
int64_t crumble_synth_n_bases(const crumble_synth *s) { return s->b.n_reads ? s->seq_off[s->b.n_reads] : 0; }

void crumble_synth_free(crumble_synth *s)
{
    if (!s) return;
    free(s->pos); free(s->mapq); free(s->flag); free(s->seq_off); free(s->cig_off);
    free(s->cigar); free(s->bases); free(s->quals); free(s);
}
