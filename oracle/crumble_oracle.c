/*
 * crumble_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * A sequential, plain-C restatement of reference crumble's column engine
 * (/root/reference/snp_score.c, str_finder.c, bed.c @ v0.9.1), written from
 * the behaviour of those files and citing file:line for every function it
 * follows.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may link or call this file; the shipped engine (crumble_amd/csrc) never
 * does, and has no CPU fallback.
 *
 * Pinning (see DESIGN.md "Oracle"):
 *   - find_STR/add_rep are checked against the REAL reference str_finder.c,
 *     compiled unmodified into oracle/_ref/ (it needs no external header).
 *   - the whole transcode() path end to end -- trigger, mask_LC_regions, the STR window and its growth, back-fill,
 *     qhigh / qlow, and pblock -- against the reference's own published input -> output pair: the mpileup before and
 *     after `crumble -9p8 -l5 -u40 -i1.0,2` of the man page (crumble.1 EXAMPLES), all 355 printed cells
 *     (tests/golden/crumble1_example.json, tests/test_manpage_example.py).
 *   - calculate_consensus_pileup, pblock, ref2query_pos, bam_qpos2rpos and
 *     mask_LC_regions are checked against the known answers that the survey
 *     obtained from the compiled reference and against
 *     the documented examples in str_finder.h:31-35.
 *   - snp_score.c itself cannot be built here (it includes htslib headers that
 *     are not in the image and not under /root/reference), and the pileup
 *     column construction lives in htslib (bam_plp_auto), an un-vendored,
 *     version-unpinned dependency: for that part PARITY IS UNPINNED; it is
 *     restated from htslib's published behaviour and
 *     exercised with hand-derived CIGAR cases.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <limits.h>
#include <math.h>
#include <float.h>

#include "../include/crumble_hip.h"
#include "crumble_oracle.h"

#define OMIN(a,b) ((a)<(b)?(a):(b))
#define OMAX(a,b) ((a)>(b)?(a):(b))
#define OABS(a)   ((a)>=0?(a):-(a))

/* BAM constants (SAM spec; htslib sam.h values) */
#define O_FUNMAP      4
#define O_FSECONDARY  256
#define O_FQCFAIL     512
#define O_FDUP        1024
#define O_CMATCH 0
#define O_CINS   1
#define O_CDEL   2
#define O_CREF_SKIP 3
#define O_CSOFT_CLIP 4
#define O_CHARD_CLIP 5
#define O_CPAD   6
#define O_CEQUAL 7
#define O_CDIFF  8
/* bit0: consumes query, bit1: consumes reference (BAM_CIGAR_TYPE 0x3C1A7) */
static inline int cig_type(int op) { return (0x3C1A7 >> (op << 1)) & 3; }
#define cig_op(c)  ((int)((c) & 0xf))
#define cig_len(c) ((int)((c) >> 4))

#define CONS_DISCREP 4      /* snp:252 */
#define CONS_MQUAL   16     /* snp:255 */
#define MAX_DEPTH    20000  /* snp:92  */
#define MASK_WIN     250    /* snp:1229 */

/* ------------------------------------------------------------------------
 * Tables: consensus_init (snp:378-489) for tech row STECH_SOLEXA only (the
 * only row read, snp:600; its tech_undercall is 1.0 so the *= are identities),
 * plus the statics of calculate_consensus_pileup (snp:560-575).
 */
static double t_pMM[101], t_p__[101], t_p_M[101];
static double t_lprior15[15];
static double t_e_tab_a[1002], t_e_tab2_a[1002];
static double *t_e_tab = &t_e_tab_a[500], *t_e_tab2 = &t_e_tab2_a[500];
static double t_q2p[101], t_mqual_pow[256];
static int t_init_done = 0;

static void oracle_tables_init(void)
{
    int i;
    double prior[25];
    const double p_het = 1e-6;                       /* snp:283 */
    if (t_init_done) return;

    for (i = -500; i <= 500; i++) t_e_tab[i]  = exp(i);        /* snp:381 */
    for (i = -500; i <= 500; i++) t_e_tab2[i] = exp(i / 10.);  /* snp:383 */

    for (i = 0; i < 25; i++) prior[i] = p_het / 20;            /* snp:389 */
    prior[0] = prior[6] = prior[12] = prior[18] = prior[24] = (1 - p_het) / 5;
    {   /* snp:393-407: 15 unordered genotypes over {A,C,G,T,*} */
        static const int idx[15]  = {0,1,2,3,4, 6,7,8,9, 12,13,14, 18,19, 24};
        static const int dbl[15]  = {0,1,1,1,1, 0,1,1,1,  0, 1, 1,  0, 1,  0};
        for (i = 0; i < 15; i++)
            t_lprior15[i] = dbl[i] ? log(prior[idx[i]] * 2) : log(prior[idx[i]]);
    }
    for (i = 1; i < 101; i++) {                                 /* snp:412 */
        double prob = 1 - pow(10, -i / 10.0);
        t_pMM[i] = log(prob / 5);                               /* snp:464 */
        t_p__[i] = log((1 - prob) / 20);                        /* snp:465 */
        t_p_M[i] = log((exp(t_pMM[i]) + exp(t_p__[i])) / 2);    /* snp:466 */
    }
    t_pMM[0] = t_pMM[1]; t_p__[0] = t_p__[1]; t_p_M[0] = t_p_M[1]; /* snp:478 */

    for (i = 0; i <= 100; i++) t_q2p[i] = pow(10, -i / 10.0);   /* snp:564 */
    for (i = 0; i < 255; i++)                                   /* snp:568 */
        t_mqual_pow[i] = 1 - pow(10, -(i / 2 + .05) / 10.0);    /* int i/2 */
    t_mqual_pow[255] = t_mqual_pow[10];                         /* snp:574 */
    t_init_done = 1;
}

/* snp:491-503 */
static inline double o_fast_exp(double y)
{
    if (y >= -50 && y <= 50) return t_e_tab2[(int)(y * 10)];
    if (y < -500) y = -500;
    if (y > 500) y = 500;
    return t_e_tab[(int)y];
}

/* snp:506-518: exponent extraction (bias 1024) + cubic on the mantissa with
 * float-typed literals promoted to double; four separate double operations. */
static inline double o_fast_log2(double val)
{
    int64_t x;
    int log_2;
    memcpy(&x, &val, 8);
    log_2 = (int)((x >> 52) & 2047) - 1024;
    x &= ~(2047LL << 52);
    x += 1023LL << 52;
    memcpy(&val, &x, 8);
    val = ((-1.0f / 3) * val + 2) * val - 2.0f / 3;
    return val + log_2;
}
#define O_TENLOG2OVERLOG10 3.0103                   /* snp:287 */
#define o_ph_log(x) (-O_TENLOG2OVERLOG10 * o_fast_log2((x)))   /* snp:527 */

/* Adjusted quality of CONS_MQUAL mode (snp:632-638). */
static inline uint8_t o_mqual_adjust(uint8_t qual, uint8_t mapq)
{
    double _p = t_mqual_pow[qual];
    double _m = t_mqual_pow[mapq];
    return (uint8_t)(o_ph_log(1 - (_m * _p + (1 - _m) / 4)));
}

void crumble_oracle_adjust_lut(uint8_t *lut /* [256*256] mapq-major */)
{
    int m, q;
    oracle_tables_init();
    for (m = 0; m < 256; m++)
        for (q = 0; q < 256; q++)
            lut[m * 256 + q] = o_mqual_adjust((uint8_t)q, (uint8_t)m);
}

/* ------------------------------------------------------------------------
 * One pileup entry as the column engine sees it (the bam_pileup1_t fields
 * crumble reads: snp:589-623, 1662-1760, 1824-1920).
 */
typedef struct {
    int64_t r;          /* read index in the batch */
    int qpos, indel;
    int is_del, is_head, is_tail, is_refskip;
} plp_t;

typedef struct {
    int call, het_call, het_phred, phred, depth;
    float discrep;
    int call_preserve;
} cons_t;   /* consensus_t snp:257-281 */

/* The context the column functions need about reads. */
typedef struct {
    const crumble_batch *b;
    const crumble_params *p;
    uint8_t *qc;        /* pileup copy: capped qualities (cap_quality snp:1325) */
    uint8_t *q2;        /* output copy (uncapped at start)           */
    uint8_t bin2[256];
} octx_t;

static inline int o_lqseq(const crumble_batch *b, int64_t r)
{ return (int)(b->seq_off[r + 1] - b->seq_off[r]); }

/* nt16 code of query index i of read r as bam_seqi() would return it,
 * including the one-past-the-end nibble that mask_LC_regions reads (Q7). */
static inline int o_seqi(const octx_t *c, int64_t r, int i)
{
    int l = o_lqseq(c->b, r);
    if (i < l) return c->b->bases[c->b->seq_off[r] + i] & 0xf;
    if (i == l) {
        if (l & 1) return 0;                        /* pad nibble            */
        if (l == 0) return 0;                       /* fenced (no qual byte) */
        return c->qc[c->b->seq_off[r]] >> 4;        /* high nibble of qual[0]*/
    }
    return 0;
}

/* calculate_consensus_pileup (snp:533-797). */
static void o_consensus(const octx_t *c, int flags, const plp_t *p, int np, cons_t *cons)
{
    static const int map_sing[15] = {0,5,5,5,5, 1,5,5,5, 2,5,5, 3,5, 4};
    static const int map_het[15]  = {0,1,2,3,4, 6,7,8,9, 12,13,14, 18,19, 24};
    static const int L[16] = {5,0,1,5, 2,5,5,5, 3,5,5,5, 5,5,5,5};   /* snp:603 */
    const double min_e_exp = DBL_MIN_EXP * log(2) + 1;               /* snp:540 */
    const uint8_t *pq = c->p->preserve_qual;
    double S[15] = {0}, sumsC[6] = {0};
    int counts[6] = {0};
    int depth = 0, n, j;

    oracle_tables_init();
    cons->call_preserve = 0;
    for (n = 0; n < np; n++) {
        int64_t r = p[n].r;
        int lq, base;
        uint8_t qual;
        const uint8_t *qptr;
        double MM, __, _M;

        if (p[n].is_refskip) continue;                     /* snp:589 */
        lq = o_lqseq(c->b, r);
        if (!lq) continue;                                 /* snp:594 */

        base = L[o_seqi(c, r, p[n].qpos)];                 /* snp:597,608 */
        qptr = &c->qc[c->b->seq_off[r] + p[n].qpos];
        qual = (p[n].qpos < lq) ? *qptr : 0;               /* fence: D at read end */
        if (p[n].is_del) base = 4;                         /* snp:609 */

        if (pq[qual])      cons->call_preserve |= 1 << base;          /* snp:611 */
        if (pq[qual] > 1)  cons->call_preserve |= (1 << base) << 8;   /* snp:613 */
        if (p[n].indel > 0) {                              /* snp:616-623 */
            int ins;
            for (ins = 0; ins < p[n].indel; ins++) {
                int qi = p[n].qpos + 1 + ins;
                if (qi < lq && pq[c->qc[c->b->seq_off[r] + qi]])
                    cons->call_preserve |= 1 << 4;
            }
        }

        if (flags & CONS_MQUAL)                            /* snp:632-638 */
            qual = o_mqual_adjust(qual, c->b->mapq[r]);
        if (qual < 1) qual = 1;                            /* snp:641 */
        if (qual > 100) qual = 100;                        /* fence (Q5)  */

        __ = t_p__[qual];                                  /* snp:644-646 */
        MM = t_pMM[qual] - __;
        _M = t_p_M[qual] - __;

        if (flags & CONS_DISCREP)                          /* snp:648-652 */
            sumsC[base] += 1 - t_q2p[qual];
        counts[base]++;

        switch (base) {                                    /* snp:656-683 */
        case 0: S[0]+=MM; S[1]+=_M; S[2]+=_M; S[3]+=_M; S[4]+=_M; break;
        case 1: S[1]+=_M; S[5]+=MM; S[6]+=_M; S[7]+=_M; S[8]+=_M; break;
        case 2: S[2]+=_M; S[6]+=_M; S[9]+=MM; S[10]+=_M; S[11]+=_M; break;
        case 3: S[3]+=_M; S[7]+=_M; S[10]+=_M; S[12]+=MM; S[13]+=_M; break;
        case 4: S[4]+=_M; S[8]+=_M; S[11]+=_M; S[13]+=_M; S[14]+=MM; break;
        case 5:
            S[0]+=MM; S[1]+=MM; S[2]+=MM; S[3]+=MM; S[4]+=_M;
                      S[5]+=MM; S[6]+=MM; S[7]+=MM; S[8]+=_M;
                                S[9]+=MM; S[10]+=MM; S[11]+=_M;
                                          S[12]+=MM; S[13]+=_M;
            break;
        }
        depth++;
    }

    {
        double shift = -DBL_MAX, max = -DBL_MAX, max_het = -DBL_MAX, norm[15];
        double tot1, tot2;
        int call = 0, het_call = 0, ph;

        for (j = 0; j < 15; j++) {                         /* snp:707-725 */
            S[j] += t_lprior15[j];
            if (shift < S[j]) shift = S[j];
            if (j != 0 && j != 5 && j != 9 && j != 12 && j != 14) {
                if (max_het < S[j]) { max_het = S[j]; het_call = j; }
                continue;
            }
            if (max < S[j]) { max = S[j]; call = j; }
        }
        for (j = 0; j < 15; j++) {                         /* snp:736-741 */
            double e;
            S[j] -= shift;
            e = o_fast_exp(S[j]);
            S[j] = (S[j] > min_e_exp) ? e : DBL_MIN;
            norm[j] = 0;
        }
        tot1 = tot2 = 0;                                   /* snp:743-749 */
        for (j = 0; j < 15; j++) {
            norm[j]      += tot1;
            norm[14 - j] += tot2;
            tot1 += S[j];
            tot2 += S[14 - j];
        }

        if (depth && depth != counts[5]) {                 /* snp:752 */
            double m, cc;
            cons->depth = depth;
            cons->call = map_sing[call];
            if (norm[call] == 0) norm[call] = DBL_MIN;
            ph = o_ph_log(norm[call]) + .5;                /* snp:759 */
            cons->phred = ph > 255 ? 255 : (ph < 0 ? 0 : ph);
            cons->het_call = map_het[het_call];
            if (norm[het_call] == 0) norm[het_call] = DBL_MIN;
            ph = O_TENLOG2OVERLOG10 *
                 (o_fast_log2(S[het_call]) - o_fast_log2(norm[het_call])) + .5;
            cons->het_phred = ph;                          /* snp:766-768 */
            cons->discrep = 0;
            if (flags & CONS_DISCREP) {                    /* snp:773-785 */
                m = sumsC[0] + sumsC[1] + sumsC[2] + sumsC[3] + sumsC[4];
                if (cons->het_phred > 0)
                    cc = sumsC[cons->het_call % 5] + sumsC[cons->het_call / 5];
                else
                    cc = sumsC[cons->call];
                cons->discrep = (m - cc) / sqrt(m);
            }
        } else {                                           /* snp:786-793 */
            cons->call = 5; cons->het_call = 0; cons->het_phred = 0;
            cons->phred = 0; cons->depth = 0; cons->discrep = 0;
        }
    }
}

/* pblock (snp:803-834) on a raw quality string; qual[len] is read/written
 * when a preserve_qual run reaches the end (Q12): we model that byte as
 * `sentinel` (idempotent in the reference, no visible effect). */
void crumble_oracle_pblock(uint8_t *qual, int len, int level, int qcap,
                           const uint8_t *preserve_qual)
{
    int i, j, qmin = INT_MAX, qmax = INT_MIN;
    int last_qmin = 0, last_qmax = 0, mid;
    level *= 2;
    for (i = j = 0; i < len; i++) {
        if (qmin > qual[i]) qmin = qual[i];
        if (qmax < qual[i]) qmax = qual[i];
        if (qmax - qmin > level || preserve_qual[qual[i]]) {
            mid = (last_qmin + last_qmax) / 2;
            if (mid > qcap) mid = qcap;
            memset(qual + j, mid, i - j);
            while (i < len && preserve_qual[qual[i]]) i++;
            if (i < len) qmin = qmax = qual[i];
            else         qmin = qmax = 0;      /* byte past the end: no effect */
            j = i;
        }
        last_qmin = qmin;
        last_qmax = qmax;
    }
    mid = (last_qmin + last_qmax) / 2;
    if (j < len) memset(qual + j, mid, OMIN(i, len) - j);
}

/* ref2query_pos (snp:1156-1179). */
static int o_ref2query_pos(const crumble_batch *b, int64_t r, int pos)
{
    const uint32_t *cig = b->cigar + b->cig_off[r];
    int n = (int)(b->cig_off[r + 1] - b->cig_off[r]);
    int p = b->pos[r], q = 0, i;
    for (i = 0; i < n; i++) {
        int op = cig_op(cig[i]), op_len = cig_len(cig[i]);
        if (p + ((cig_type(op) & 2) ? op_len : 0) < pos) {
            if (cig_type(op) & 1) q += op_len;
            if (cig_type(op) & 2) p += op_len;
            continue;
        }
        if (cig_type(op) & 1) q += (pos - p);
        return q >= 0 ? q : 0;
    }
    return q;
}

/* bam_qpos2rpos (snp:1205-1219). */
static int o_qpos2rpos(const crumble_batch *b, int64_t r, int qpos)
{
    const uint32_t *cig = b->cigar + b->cig_off[r];
    int n = (int)(b->cig_off[r + 1] - b->cig_off[r]);
    int k, rpos = b->pos[r], aqpos = 0;
    for (k = 0; k < n && aqpos < qpos; k++) {
        int op = cig_op(cig[k]), l = cig_len(cig[k]);
        if (cig_type(op) & 2) {
            if (l <= qpos - aqpos) rpos += l;
            else                   rpos += qpos - aqpos;
        }
        if (cig_type(op) & 1) aqpos += l;
    }
    return rpos;
}

/* ------------------------------------------------------------------------
 * find_STR / add_rep (str_finder.c:34-189) with the utlist doubly linked list
 * replaced by an ordered array.  Input never contains '*' here (mask_LC_regions
 * builds it from seq_nt16_str), so the pad-skipping branches are inert; they
 * are kept so the function is also exact on padded consensus strings.
 */
typedef struct { int start, end; } orep_t;
typedef struct { orep_t *e; int n, cap; } oreplist_t;

static const int STR_L[256] = {     /* str_finder.c:15-32: C,c=1 G,g=2 T,t,U,u=3 */
    ['C'] = 1, ['c'] = 1, ['G'] = 2, ['g'] = 2,
    ['T'] = 3, ['t'] = 3, ['U'] = 3, ['u'] = 3,
};

static void o_list_delete(oreplist_t *l, int i)
{
    memmove(&l->e[i], &l->e[i + 1], (l->n - i - 1) * sizeof(orep_t));
    l->n--;
}

static void o_add_rep(oreplist_t *list, const char *cons, int clen, int pos, int rlen)
{
    const char *cp1, *cp2, *cp_end;
    orep_t el;
    int i;

    if (list->n) {                                        /* str_finder.c:41-45 */
        const orep_t *t = &list->e[list->n - 1];
        if (t->start <= pos - rlen * 2 + 1 && t->end >= pos) return;
    }
    cp2 = &cons[pos + 1];
    for (cp1 = &cons[pos], i = 1; i < rlen; cp1--)        /* str_finder.c:52-58 */
        if (*cp1 == '*') continue; else i++;
    while (*cp1 == '*') cp1--;

    cp_end = &cons[clen];                                 /* str_finder.c:62-75 */
    while (cp2 < cp_end) {
        while (*cp1 == '*') cp1++;
        while (*cp2 == '*') cp2++;
        if (STR_L[(unsigned char)*cp1] != STR_L[(unsigned char)*cp2]) break;
        cp1++; cp2++;
    }
    el.end = pos + (int)(cp2 - &cons[pos + 1]);           /* str_finder.c:80 */
    pos++;
    {
        int k = rlen;
        while (k--) {                                     /* str_finder.c:82-85 */
            while (cons[--pos] == '*');
            while (cons[--pos] == '*');
        }
    }
    while (pos > 1 && cons[pos - 1] == '*') pos--;
    el.start = pos;

    if (list->n) {                                        /* str_finder.c:106-122 */
        int cur = list->n - 1;
        for (;;) {
            int prev = (cur == 0) ? list->n - 1 : cur - 1;   /* head->prev == tail */
            int deleted = 0, was_head = (cur == 0);
            if (list->e[cur].end < el.start) break;
            if (list->e[cur].start >= el.start) { o_list_delete(list, cur); deleted = 1; }
            if (was_head && !deleted) break;
            if (!list->n) break;
            if (was_head) prev = list->n - 1;     /* old tail, after the shift */
            cur = prev;
        }
    }
    if (list->n == list->cap) {
        list->cap = list->cap ? list->cap * 2 : 64;
        list->e = realloc(list->e, list->cap * sizeof(orep_t));
    }
    list->e[list->n++] = el;
}

static void o_find_STR(oreplist_t *reps, const char *cons, int len)
{
    int i, j;
    uint32_t w = 0;
    reps->n = 0;
    for (i = j = 0; i < len && j < 15; i++) {             /* str_finder.c:140-162 */
        if (cons[i] == '*') continue;
        w <<= 2; w |= STR_L[(unsigned char)cons[i]];
        if (j >=  1 && (w&0x0003) == ((w>> 2)&0x0003)) o_add_rep(reps, cons, len, i, 1);
        if (j >=  3 && (w&0x000f) == ((w>> 4)&0x000f)) o_add_rep(reps, cons, len, i, 2);
        if (j >=  5 && (w&0x003f) == ((w>> 6)&0x003f)) o_add_rep(reps, cons, len, i, 3);
        if (j >=  7 && (w&0x00ff) == ((w>> 8)&0x00ff)) o_add_rep(reps, cons, len, i, 4);
        if (j >=  9 && (w&0x03ff) == ((w>>10)&0x03ff)) o_add_rep(reps, cons, len, i, 5);
        if (j >= 11 && (w&0x0fff) == ((w>>12)&0x0fff)) o_add_rep(reps, cons, len, i, 6);
        if (j >= 13 && (w&0x3fff) == ((w>>14)&0x3fff)) o_add_rep(reps, cons, len, i, 7);
        j++;
    }
    for (; i < len; i++) {                                /* str_finder.c:164-186 */
        if (cons[i] == '*') continue;
        w <<= 2; w |= STR_L[(unsigned char)cons[i]];
        if      ((w&0xffff) == ((w>>16)&0xffff)) o_add_rep(reps, cons, len, i, 8);
        else if ((w&0x3fff) == ((w>>14)&0x3fff)) o_add_rep(reps, cons, len, i, 7);
        else if ((w&0x0fff) == ((w>>12)&0x0fff)) o_add_rep(reps, cons, len, i, 6);
        else if ((w&0x03ff) == ((w>>10)&0x03ff)) o_add_rep(reps, cons, len, i, 5);
        else if ((w&0x00ff) == ((w>> 8)&0x00ff)) o_add_rep(reps, cons, len, i, 4);
        else if ((w&0x003f) == ((w>> 6)&0x003f)) o_add_rep(reps, cons, len, i, 3);
        else if ((w&0x000f) == ((w>> 4)&0x000f)) o_add_rep(reps, cons, len, i, 2);
        else if ((w&0x0003) == ((w>> 2)&0x0003)) o_add_rep(reps, cons, len, i, 1);
    }
}

int crumble_oracle_find_str(const char *cons, int len, int *starts, int *ends, int max)
{
    oreplist_t l = {0};
    int i, n;
    o_find_STR(&l, cons, len);
    n = l.n;
    for (i = 0; i < n && i < max; i++) { starts[i] = l.e[i].start; ends[i] = l.e[i].end; }
    free(l.e);
    return n;
}

/* mask_LC_regions (snp:1230-1290): widens *min_pos / *max_pos. */
static void o_mask_LC(const octx_t *c, int is_indel, int64_t r, int rpos,
                      int *min_pos, int *max_pos, oreplist_t *scratch)
{
    static const char nt16[] = "=ACMGRSVTWYHKDBN";
    char seq[MASK_WIN * 2 + 2];
    int lq = o_lqseq(c->b, r);
    int start = OMAX(rpos - MASK_WIN, 0);
    int end = OMIN(rpos + MASK_WIN, lq);
    int len = end - start + 1, i;
    int add = is_indel ? c->p->iSTR_add : c->p->sSTR_add;

    for (i = start; i <= end; i++)                         /* snp:1241 inclusive */
        seq[i - start] = nt16[o_seqi(c, r, i)];
    o_find_STR(scratch, seq, len);
    for (i = 0; i < scratch->n; i++) {                     /* snp:1246-1287 */
        int s = scratch->e[i].start, e = scratch->e[i].end;
        if (!(rpos + add >= s + start && rpos - add <= e + start)) continue;
        s = o_qpos2rpos(c->b, r, s + start);
        e = o_qpos2rpos(c->b, r, e + start);
        if (*min_pos > s) *min_pos = s;
        if (*max_pos < e) *max_pos = e;
    }
}

/* ------------------------------------------------------------------------
 * The pileup engine: htslib bam_plp_auto semantics.
 * PARITY UNPINNED for this part (htslib source is not in /root/reference).
 */
typedef struct {
    int64_t r;
    int beg, end;        /* [beg,end) on the reference, raw cigar length */
    int k, x, y;         /* htslib cstate_t: current op, its ref start, query consumed */
} oact_t;

static inline int is_mdnx(int op)
{ return op == O_CMATCH || op == O_CDEL || op == O_CREF_SKIP || op == O_CEQUAL || op == O_CDIFF; }
static inline int is_mx(int op)
{ return op == O_CMATCH || op == O_CEQUAL || op == O_CDIFF; }

/* htslib sam.c resolve_cigar2(): fills one entry for column `pos`. */
static void o_resolve_cigar(const crumble_batch *b, oact_t *s, int pos, plp_t *p)
{
    const uint32_t *cigar = b->cigar + b->cig_off[s->r];
    int n_cigar = (int)(b->cig_off[s->r + 1] - b->cig_off[s->r]);
    int k;

    if (s->k == -1) {                     /* first column of this read */
        s->x = b->pos[s->r]; s->y = 0;
        for (k = 0; k < n_cigar; k++) {
            int op = cig_op(cigar[k]), l = cig_len(cigar[k]);
            if (is_mdnx(op)) break;
            else if (op == O_CINS || op == O_CSOFT_CLIP) s->y += l;
        }
        s->k = k;
    } else {
        int l = cig_len(cigar[s->k]);
        if (pos - s->x >= l) {            /* step to the next M/D/N/=/X op */
            if (is_mx(cig_op(cigar[s->k]))) s->y += l;
            s->x += l;
            for (k = s->k + 1; k < n_cigar; k++) {
                int op = cig_op(cigar[k]), l2 = cig_len(cigar[k]);
                if (is_mdnx(op)) break;
                else if (op == O_CINS || op == O_CSOFT_CLIP) s->y += l2;
            }
            s->k = k;
        }
    }
    {
        int op = cig_op(cigar[s->k]), l = cig_len(cigar[s->k]);
        p->is_del = p->indel = p->is_refskip = 0;
        if (s->x + l - 1 == pos && s->k + 1 < n_cigar) {   /* peek next op */
            int op2 = cig_op(cigar[s->k + 1]), l2 = cig_len(cigar[s->k + 1]);
            if (op2 == O_CDEL && op != O_CDEL) {
                p->indel = -l2;
                for (k = s->k + 2; k < n_cigar; k++) {
                    op2 = cig_op(cigar[k]); l2 = cig_len(cigar[k]);
                    if (op2 == O_CDEL) p->indel -= l2; else break;
                }
            } else if (op2 == O_CINS) {
                p->indel = l2;
                for (k = s->k + 2; k < n_cigar; k++) {
                    op2 = cig_op(cigar[k]); l2 = cig_len(cigar[k]);
                    if (op2 == O_CINS) p->indel += l2;
                    else if (op2 != O_CPAD) break;
                }
            } else if (op2 == O_CPAD && s->k + 2 < n_cigar) {
                int l3 = 0;
                for (k = s->k + 2; k < n_cigar; k++) {
                    op2 = cig_op(cigar[k]); l2 = cig_len(cigar[k]);
                    if (op2 == O_CINS) l3 += l2;
                    else if (op2 == O_CDEL || op2 == O_CMATCH || op2 == O_CREF_SKIP ||
                             op2 == O_CEQUAL || op2 == O_CDIFF) break;
                }
                if (l3 > 0) p->indel = l3;
            }
        }
        if (is_mx(op)) {
            p->qpos = s->y + (pos - s->x);
        } else {                                   /* D or N */
            p->is_del = 1; p->qpos = s->y;
            p->is_refskip = (op == O_CREF_SKIP);
        }
        p->is_head = (pos == b->pos[s->r]);
        p->is_tail = (pos == s->end - 1);
    }
    p->r = s->r;
}

/* pileup_callback's routing (snp:1130-1149) + htslib's default flag mask. */
int crumble_oracle_in_pileup(const crumble_batch *b, int64_t r)
{
    const uint32_t *cig = b->cigar + b->cig_off[r];
    int n = (int)(b->cig_off[r + 1] - b->cig_off[r]), i;
    if (b->tid < 0) return 0;
    if (b->flag[r] & (O_FUNMAP | O_FSECONDARY | O_FQCFAIL | O_FDUP)) return 0;
    for (i = 0; i < n; i++)
        if (cig_type(cig_op(cig[i])) & 2) return 1;
    return 0;
}

static int o_reflen(const crumble_batch *b, int64_t r)
{
    const uint32_t *cig = b->cigar + b->cig_off[r];
    int n = (int)(b->cig_off[r + 1] - b->cig_off[r]), i, l = 0;
    for (i = 0; i < n; i++)
        if (cig_type(cig_op(cig[i])) & 2) l += cig_len(cig[i]);
    return l;
}

/* ------------------------------------------------------------------------
 * transcode() restated (snp:1336-2029) for one contig held in a batch.
 */
int crumble_oracle_contig(const crumble_params *p, const crumble_batch *b,
                          uint8_t *qual_out, uint8_t *read_flags_out,
                          int64_t *stats, crumble_column *trace, int64_t trace_max,
                          int64_t *n_trace)
{
    static const int bit2call[32] = {      /* snp:1384-1417 */
        99,0,1,99,2,99,99,99,3,99,99,99,99,99,99,99,
        4,99,99,99,99,99,99,99,99,99,99,99,99,99,99,99 };
    octx_t c;
    int64_t n = b->n_reads, nb = n ? b->seq_off[n] : 0, r, i64;
    int64_t st[CRUMBLE_ST_N] = {0};
    uint8_t *keep = NULL, *bit15 = NULL, *inplp = NULL;
    oact_t *act = NULL; int n_act = 0, cap_act = 0;
    plp_t *plp = NULL; int cap_plp = 0;
    oreplist_t scratch = {0};
    int str_snp = (p->sSTR_add || p->sSTR_mul);           /* snp:1345 */
    int bed_idx = 0;
    int min_pos = INT_MAX, max_pos = 0, min_pos2 = INT_MAX, max_pos2 = 0;
    int64_t total_depth = 0, total_col = 0, ntr = 0;
    int64_t next_read = 0;
    int pos = 0, rc = CRUMBLE_OK;

    oracle_tables_init();
    memset(&c, 0, sizeof(c));
    c.b = b; c.p = p;
    c.qc = malloc(nb + 1); c.q2 = malloc(nb + 1);
    keep = calloc(n + 1, 1); bit15 = calloc(n + 1, 1); inplp = calloc(n + 1, 1);
    if (!c.qc || !c.q2 || !keep || !bit15 || !inplp) { rc = CRUMBLE_ENOMEM; goto done; }
    {   /* init_bins (snp:234-247) */
        int i;
        for (i = 0; i < p->qcutoff && i < 256; i++) c.bin2[i] = p->qlow;
        for (; i < 256; i++) c.bin2[i] = p->qhigh;
        for (i = 0; i < 256; i++) if (p->preserve_qual[i] > 1) c.bin2[i] = i;
    }
    memcpy(c.q2, b->quals, nb);                /* bam_dup1 before the cap (Q9) */
    for (i64 = 0; i64 < nb; i64++) {           /* cap_quality snp:1325-1332    */
        uint8_t q = b->quals[i64];
        c.qc[i64] = (q > p->qcap && !p->preserve_qual[q]) ? p->qcap : q;
    }
    {   /* only reads that enter the pileup must be sorted (snp:1074-1077) */
        int last = INT_MIN;
        for (r = 0; r < n; r++) {
            inplp[r] = crumble_oracle_in_pileup(b, r);
            if (!inplp[r]) continue;
            if (b->pos[r] < last) { rc = CRUMBLE_EUNSORTED; goto done; }
            last = b->pos[r];
        }
    }

    /* ---- column loop: bam_plp_auto (snp:1437) ---- */
    for (;;) {
        int n_plp = 0, i, a;
        int preserve = 0, indel = 0, perfect = 1;
        int m;
        cons_t consA, consB;
        int call1 = 0, call2 = 0;
        int hA = 0, sA = 0, hB = 0, sB = 0;
        int had_indel = 0, had_indel_Q = 0, low_mq_count = 0, keep_qual = 0, indel_cnt = 0;
        int indel_sz = 0, indel_depth[101], clipped = 0, n_overlap = 0;
        int skipped = 0, why = 0;

        /* skip reads that never enter the pileup */
        while (next_read < n && !inplp[next_read]) next_read++;
        if (n_act == 0) {
            if (next_read >= n) break;
            if (pos < b->pos[next_read]) pos = b->pos[next_read];
        }
        /* push every read starting at or before this column */
        while (next_read < n && (!inplp[next_read] || b->pos[next_read] <= pos)) {
            if (inplp[next_read]) {
                if (n_act == cap_act) {
                    cap_act = cap_act ? cap_act * 2 : 256;
                    act = realloc(act, cap_act * sizeof(*act));
                }
                act[n_act].r = next_read;
                act[n_act].beg = b->pos[next_read];
                act[n_act].end = b->pos[next_read] + o_reflen(b, next_read);
                act[n_act].k = -1; act[n_act].x = act[n_act].y = 0;
                n_act++;
            }
            next_read++;
        }
        /* drop finished reads, keep order */
        for (a = i = 0; i < n_act; i++)
            if (act[i].end > pos) act[a++] = act[i];
        n_act = a;
        if (n_act == 0) continue;          /* jump handled at loop top */

        if (n_act > cap_plp) {
            cap_plp = n_act * 2;
            plp = realloc(plp, cap_plp * sizeof(*plp));
        }
        for (i = 0; i < n_act; i++)
            if (act[i].beg <= pos) o_resolve_cigar(b, &act[i], pos, &plp[n_plp++]);
        if (n_plp == 0) { pos++; continue; }

        /* ---- body of the while loop, snp:1438-1983 ---- */
        if (b->n_bed) {                                    /* snp:1443-1463 */
            while (bed_idx < b->n_bed && b->bed_end[bed_idx] < pos) bed_idx++;
            if (bed_idx < b->n_bed && b->bed_start[bed_idx] <= pos && b->bed_end[bed_idx] > pos)
                preserve = 2;
        }
        for (m = i = 0; i < n_plp; i++) if (plp[i].is_refskip) m++;   /* snp:1466 */
        if (m == n_plp) { skipped = 1; goto trace_col; }   /* `continue`: no retirement */

        st[CRUMBLE_ST_COLUMNS]++;
        total_depth += n_plp; total_col++;                 /* snp:1490 */
        if (n_plp > MAX_DEPTH) { skipped = 2; st[CRUMBLE_ST_TOO_DEEP]++; goto too_deep; }

        if (pos > max_pos2) {                              /* snp:1508-1511 */
            min_pos2 = min_pos = INT_MAX;
            max_pos2 = max_pos = 0;
        }

        if (p->min_qual_A != 0) {                          /* snp:1523-1532 */
            o_consensus(&c, CONS_DISCREP, plp, n_plp, &consA);
            if (consA.het_phred > 0) { call1 = 1 << (consA.het_call / 5); call2 = 1 << (consA.het_call % 5); }
            else call1 = call2 = 1 << consA.call;
        }
        if (p->min_qual_B != 0) {                          /* snp:1534-1543 */
            o_consensus(&c, CONS_DISCREP | CONS_MQUAL, plp, n_plp, &consB);
            if (consB.het_phred > 0) { call1 = 1 << (consB.het_call / 5); call2 = 1 << (consB.het_call % 5); }
            else call1 = call2 = 1 << consB.call;
        }
        if (p->min_qual_A) {                               /* snp:1576-1583 */
            hA = consA.het_phred > 0 ? consA.het_call : consA.call * 5 + consA.call;
            sA = consA.het_phred > 0 ? consA.het_phred : consA.phred;
        }
        if (p->min_qual_B) {                               /* snp:1584-1591 */
            hB = consB.het_phred > 0 ? consB.het_call : consB.call * 5 + consB.call;
            sB = consB.het_phred > 0 ? consB.het_phred : consB.phred;
        }
        if (p->min_qual_A && p->min_qual_B && hA != hB) st[CRUMBLE_ST_DIFF]++;
        if (p->min_qual_A) {                               /* snp:1594-1609 */
            if (consA.het_phred > 0) { st[CRUMBLE_ST_HET_A]++; if (sA < p->min_qual_A) st[CRUMBLE_ST_HET_QUAL_A]++; }
            else                     { st[CRUMBLE_ST_HOM_A]++; if (sA < p->min_qual_A) st[CRUMBLE_ST_HOM_QUAL_A]++; }
            if (consA.discrep >= p->min_discrep_A) st[CRUMBLE_ST_DISCREP_A]++;
            if (consA.call_preserve != 1 << consA.call) perfect = 0;
        }
        if (p->min_qual_B) {                               /* snp:1610-1622 */
            if (consB.het_phred > 0) { st[CRUMBLE_ST_HET_B]++; if (sB < p->min_qual_B) st[CRUMBLE_ST_HET_QUAL_B]++; }
            else                     { st[CRUMBLE_ST_HOM_B]++; if (sB < p->min_qual_B) st[CRUMBLE_ST_HOM_QUAL_B]++; }
            if (consB.discrep >= p->min_discrep_B) st[CRUMBLE_ST_DISCREP_B]++;
        }
        preserve |= ((p->min_qual_A && p->min_qual_B && hA != hB) ||      /* snp:1624 */
                     (p->min_qual_A && sA < p->min_qual_A) ||
                     (p->min_qual_B && sB < p->min_qual_B));
        preserve |= ((p->min_qual_A && consA.discrep >= p->min_discrep_A) ||
                     (p->min_qual_B && consB.discrep >= p->min_discrep_B));
        if (p->min_qual_A && !preserve) {                  /* snp:1633-1638 */
            if ((consA.het_phred <= 0 && bit2call[consA.call_preserve & 31] != consA.call) ||
                (consA.call_preserve >> 8)) perfect = 0;
        }
        if (p->min_qual_B && !preserve) {                  /* snp:1640-1645 */
            if ((consB.het_phred <= 0 && bit2call[consB.call_preserve & 31] != consB.call) ||
                (consB.call_preserve >> 8)) perfect = 0;
        }
        if (p->perfect_col && !perfect) preserve = 1;      /* snp:1648 */

        for (i = 0; i < n_plp; i++) {                      /* snp:1662-1666 */
            low_mq_count += (b->mapq[plp[i].r] <= p->min_mqual);
            if (plp[i].indel || plp[i].is_del) { had_indel = 1; indel_cnt++; }
        }
        keep_qual = low_mq_count > p->low_mqual_perc * (n_plp + .01);   /* snp:1668 */
        st[CRUMBLE_ST_LOW_MQUAL_PERC] += keep_qual;
        if (n_plp * (total_col + 1) > p->over_depth * (total_depth + 1)) {  /* snp:1673 */
            keep_qual = 1; st[CRUMBLE_ST_OVER_DEPTH]++; why |= CRUMBLE_WHY_DEEP;        /* -b DEEP snp:1676 */
        }
        if (total_col > 1024 * 1024) { total_col >>= 1; total_depth >>= 1; }  /* snp:1684 */

        indel_depth[0] = 0;
        for (i = 0; i < n_plp; i++) {                      /* snp:1695-1760 */
            int is_indel;
            int64_t rr = plp[i].r;
            if (plp[i].is_refskip) continue;
            is_indel = (plp[i].indel || plp[i].is_del);
            if ((plp[i].is_head && plp[i].qpos > 0) ||
                (plp[i].is_tail && plp[i].qpos + 1 < o_lqseq(b, rr))) clipped++;
            if (!plp[i].is_tail && !plp[i].is_head) n_overlap++;
            if (!plp[i].is_head && !plp[i].is_tail && (plp[i].indel > 0 || had_indel)) {
                while (indel_sz < plp[i].indel && indel_sz < 100) indel_depth[++indel_sz] = 0;
                if (plp[i].indel >= 0) indel_depth[OMIN(plp[i].indel, 99)]++;
            }
            if ((is_indel || (str_snp && preserve)) &&
                ((p->min_qual_A && sA < p->min_indel_A) ||
                 (p->min_qual_B && sB < p->min_indel_B))) {
                if (is_indel) had_indel_Q++;
                if (is_indel) {
                    if (indel < OABS(plp[i].indel) + plp[i].is_del)
                        indel = OABS(plp[i].indel) + plp[i].is_del;
                } else indel = 1;
                st[CRUMBLE_ST_TRIGGERS]++;
                if (indel_cnt >= n_plp * p->indel_fract) { /* snp:1732-1740 */
                    /* called twice in the reference with an unused `apos`: idempotent */
                    o_mask_LC(&c, is_indel, rr, plp[i].qpos + 1, &min_pos, &max_pos, &scratch);
                }
                if (min_pos > pos) min_pos = pos;
                if (max_pos < pos) max_pos = pos;
                if (is_indel) {                            /* snp:1745-1755 */
                    min_pos2 = OMIN(min_pos2, pos - (pos - min_pos) * p->iSTR_mul - p->iSTR_add);
                    max_pos2 = OMAX(max_pos2, pos + (max_pos - pos) * p->iSTR_mul + p->iSTR_add);
                } else {
                    min_pos2 = OMIN(min_pos2, pos - (pos - min_pos) * p->sSTR_mul - p->sSTR_add);
                    max_pos2 = OMAX(max_pos2, pos + (max_pos - pos) * p->sSTR_mul + p->sSTR_add);
                }
            }
        }
        if (had_indel) st[CRUMBLE_ST_INDEL]++;
        if (had_indel_Q) st[CRUMBLE_ST_INDEL_QUAL]++;

        if ((clipped - 1.0) >= p->clip_perc * n_overlap) { /* snp:1764 */
            keep_qual = 1; st[CRUMBLE_ST_CLIP_PERC]++; why |= CRUMBLE_WHY_CLIP;         /* -b CLIP snp:1768 */
        }
        if (indel_sz) {                                    /* snp:1777-1819 */
            int qd1 = 0, qd2 = 0, indel_overlap = 0;
            for (i = 0; i <= indel_sz && i < 100; i++) {
                if (!indel_depth[i]) continue;
                indel_overlap += indel_depth[i];
                if (qd1 < indel_depth[i]) { qd2 = qd1; qd1 = indel_depth[i]; }
                else if (qd2 < indel_depth[i]) qd2 = indel_depth[i];
            }
            if ((indel_overlap - qd1 - qd2) > p->ins_len_perc * (indel_overlap + .1)) {
                keep_qual = 1; st[CRUMBLE_ST_INS_LEN_PERC]++; why |= CRUMBLE_WHY_INDEL_LEN;         /* snp:1802 */
            }
            if ((double)indel_overlap < p->indel_ov_perc * n_plp) {
                keep_qual = 1; st[CRUMBLE_ST_INDEL_OV_PERC]++; why |= CRUMBLE_WHY_INDEL_COVERAGE;   /* snp:1810 */
            }
        }

        for (i = 0; i < n_plp; i++) {                      /* snp:1824-1920 */
            int64_t rr = plp[i].r, so = b->seq_off[rr];
            int lq = o_lqseq(b, rr), x, base;
            uint8_t *qual;
            if (keep_qual) keep[rr] = 1;                   /* snp:1847 */
            if (plp[i].is_head && b->mapq[rr] <= p->min_mqual)   /* snp:1852-1859 */
                for (x = 0; x < lq; x++) c.q2[so + x] |= 0x80;
            if (!lq) continue;
            if (plp[i].qpos >= lq) continue;               /* fence: read ends in D/N */
            qual = &c.q2[so + plp[i].qpos];
            base = b->bases[so + plp[i].qpos] & 0xf;
            if (indel) {                                   /* snp:1870-1879 */
                for (x = o_ref2query_pos(b, rr, min_pos2); x <= plp[i].qpos; x++)
                    c.q2[so + x] = c.qc[so + x] | 0x80;
            }
            if (min_pos != INT_MAX)                        /* snp:1880-1883 */
                *qual = c.qc[so + plp[i].qpos] | 0x80;
            if (preserve || p->preserve_qual[*qual & 0x7f] >= 1 + perfect)   /* snp:1885 */
                *qual |= 0x80;
            if (preserve > 1) bit15[rr] = 1;               /* snp:1890 */
            if (!keep_qual && p->softclip) {               /* snp:1894-1904 */
                if (plp[i].is_head) {
                    for (x = 1; x <= plp[i].qpos; x++) qual[-x] = c.bin2[qual[-x]];
                } else if (plp[i].is_tail) {
                    for (x = lq - plp[i].qpos - 1; x > 0; x--) qual[x] = c.bin2[qual[x]];
                }
            }
            if (!(*qual & 0x80)) {                         /* snp:1906-1919 */
                if (base == call1 || base == call2) *qual = p->qhigh;
                else if (p->reduce_qual) *qual = p->binary_qual ? c.bin2[*qual] : p->qlow;
            }
        }

    too_deep:
        for (i = 0; i < n_plp; i++) {                      /* snp:1931-1975 */
            int64_t rr = plp[i].r;
            if (!plp[i].is_tail) continue;
            if (keep[rr])
                memcpy(&c.q2[b->seq_off[rr]], &c.qc[b->seq_off[rr]], o_lqseq(b, rr));
        }

    trace_col:
        if (trace && ntr < trace_max) {
            crumble_column *t = &trace[ntr];
            memset(t, 0, sizeof(*t));
            t->pos = pos; t->n_plp = n_plp; t->skipped = skipped;
            if (!skipped) {
                t->skipped = (uint8_t)why;
                int a1 = 0, a2 = 0;
                while ((1 << a1) != call1 && a1 < 6) a1++;
                while ((1 << a2) != call2 && a2 < 6) a2++;
                t->score = p->min_qual_B ? sB : sA;
                t->call1 = a1; t->call2 = a2;
                t->preserve = preserve; t->perfect = perfect; t->keep_qual = keep_qual;
                t->trigger = indel != 0; t->window = (min_pos != INT_MAX);
                t->min_pos2 = indel ? min_pos2 : 0;
            }
        }
        ntr++;
        pos++;
    }

    /* flush_bam_list epilogue (snp:1090-1100) for every read, passthrough ones
     * included (b_hist riders and trailing unmapped reads: snp:2002-2005). */
    for (r = 0; r < n; r++) {
        int64_t so = b->seq_off[r];
        int lq = o_lqseq(b, r), x;
        for (x = 0; x < lq; x++) c.q2[so + x] &= ~0x80;
        if (p->pblock && !bit15[r])
            crumble_oracle_pblock(&c.q2[so], lq, p->pblock, p->qcap, p->preserve_qual);
        if (read_flags_out)
            read_flags_out[r] = (keep[r] ? CRUMBLE_RF_KEEP : 0) |
                                (bit15[r] ? CRUMBLE_RF_NOPBLOCK : 0) |
                                (inplp[r] ? 0 : CRUMBLE_RF_PASSTHRU);
        st[CRUMBLE_ST_READS]++;
        st[CRUMBLE_ST_BASES] += lq;
        for (x = 0; x < lq; x++) {
            st[CRUMBLE_ST_HIST0 + c.q2[so + x]]++;
            if (c.q2[so + x] != b->quals[so + x]) st[CRUMBLE_ST_BASES_CHANGED]++;
        }
    }
    if (qual_out) memcpy(qual_out, c.q2, nb);
    if (stats) { int k; for (k = 0; k < CRUMBLE_ST_N; k++) stats[k] += st[k]; }
    if (n_trace) *n_trace = ntr;

done:
    free(c.qc); free(c.q2); free(keep); free(bit15); free(inplp);
    free(act); free(plp); free(scratch.e);
    return rc;
}

/* ---- thin unit-level entry points for the known-answer tests ---- */

/* One column made of single-base reads: bases are nt16 codes. */
int crumble_oracle_consensus(const crumble_params *p, int flags, int np,
                             const uint8_t *base16, const uint8_t *qual,
                             const uint8_t *mapq, const uint8_t *is_del,
                             const uint8_t *is_refskip,
                             int *call, int *het_call, int *het_phred,
                             int *phred, float *discrep, int *call_preserve)
{
    crumble_batch b; octx_t c; cons_t cons;
    int64_t *so = malloc((np + 1) * sizeof(*so)), *co = malloc((np + 1) * sizeof(*co));
    uint32_t *cig = malloc((np + 1) * sizeof(*cig));
    int32_t *pos = calloc(np + 1, sizeof(*pos));
    uint16_t *flag = calloc(np + 1, sizeof(*flag));
    plp_t *plp = calloc(np + 1, sizeof(*plp));
    int i;
    memset(&b, 0, sizeof(b)); memset(&c, 0, sizeof(c));
    for (i = 0; i <= np; i++) { so[i] = i; co[i] = i; cig[i] = (1 << 4) | O_CMATCH; }
    b.n_reads = np; b.pos = pos; b.mapq = mapq; b.flag = flag; b.seq_off = so;
    b.cig_off = co; b.cigar = cig; b.bases = base16; b.quals = qual;
    c.b = &b; c.p = p; c.qc = (uint8_t *)qual; c.q2 = NULL;
    for (i = 0; i < np; i++) {
        plp[i].r = i; plp[i].qpos = 0;
        plp[i].is_del = is_del ? is_del[i] : 0;
        plp[i].is_refskip = is_refskip ? is_refskip[i] : 0;
    }
    o_consensus(&c, flags, plp, np, &cons);
    *call = cons.call; *het_call = cons.het_call; *het_phred = cons.het_phred;
    *phred = cons.phred; *discrep = cons.discrep; *call_preserve = cons.call_preserve;
    free(so); free(co); free(cig); free(pos); free(flag); free(plp);
    return 0;
}

int crumble_oracle_ref2query_pos(const crumble_batch *b, int64_t r, int pos)
{ return o_ref2query_pos(b, r, pos); }
int crumble_oracle_qpos2rpos(const crumble_batch *b, int64_t r, int qpos)
{ return o_qpos2rpos(b, r, qpos); }

/* mask_LC_regions on read r of a batch; quals are capped with p->qcap first. */
int crumble_oracle_mask_lc(const crumble_params *p, const crumble_batch *b,
                           int64_t r, int is_indel, int rpos, int *min_pos, int *max_pos)
{
    octx_t c; oreplist_t s = {0};
    int64_t nb = b->seq_off[b->n_reads], i;
    memset(&c, 0, sizeof(c));
    c.b = b; c.p = p; c.qc = malloc(nb + 1);
    for (i = 0; i < nb; i++) {
        uint8_t q = b->quals[i];
        c.qc[i] = (q > p->qcap && !p->preserve_qual[q]) ? p->qcap : q;
    }
    o_mask_LC(&c, is_indel, r, rpos, min_pos, max_pos, &s);
    free(c.qc); free(s.e);
    return 0;
}

/* Pileup columns of a batch, flattened, for the CIGAR hand cases (App. C). */
int64_t crumble_oracle_pileup(const crumble_batch *b, int64_t max,
                              int32_t *col_pos, int64_t *read, int32_t *qpos,
                              int32_t *indel, uint8_t *bits /* 1 del,2 head,4 tail,8 refskip */)
{
    int64_t n = b->n_reads, next = 0, out = 0;
    oact_t *act = calloc(n + 1, sizeof(*act));
    int n_act = 0, pos = 0, i, a;
    for (;;) {
        while (next < n && !crumble_oracle_in_pileup(b, next)) next++;
        if (n_act == 0) { if (next >= n) break; if (pos < b->pos[next]) pos = b->pos[next]; }
        while (next < n && (!crumble_oracle_in_pileup(b, next) || b->pos[next] <= pos)) {
            if (crumble_oracle_in_pileup(b, next)) {
                act[n_act].r = next; act[n_act].beg = b->pos[next];
                act[n_act].end = b->pos[next] + o_reflen(b, next);
                act[n_act].k = -1; act[n_act].x = act[n_act].y = 0; n_act++;
            }
            next++;
        }
        for (a = i = 0; i < n_act; i++) if (act[i].end > pos) act[a++] = act[i];
        n_act = a;
        if (!n_act) continue;
        for (i = 0; i < n_act; i++) {
            plp_t p;
            if (act[i].beg > pos) continue;
            o_resolve_cigar(b, &act[i], pos, &p);
            if (out < max) {
                col_pos[out] = pos; read[out] = p.r; qpos[out] = p.qpos; indel[out] = p.indel;
                bits[out] = p.is_del | (p.is_head << 1) | (p.is_tail << 2) | (p.is_refskip << 3);
            }
            out++;
        }
        pos++;
    }
    free(act);
    return out;
}

/* indel_only.c:83-122: keep qualities within +-D query bases of any pileup entry that has
 * indel != 0 or is_del, set everything else to Q.  (The tool writes reads in tail order and
 * silently drops reads that never reach a column; here those are returned unchanged.) */
int crumble_oracle_indel_only(const crumble_batch *b, int D, int Q, uint8_t *qual_out)
{
    int64_t n = b->n_reads, nb = n ? b->seq_off[n] : 0, next = 0, r;
    oact_t *act = calloc(n + 1, sizeof(*act));
    uint8_t *mark = calloc(nb + 1, 1);
    int n_act = 0, pos = 0, i, a, x;
    if (!act || !mark) { free(act); free(mark); return CRUMBLE_ENOMEM; }
    memcpy(qual_out, b->quals, nb);
    for (;;) {
        while (next < n && !crumble_oracle_in_pileup(b, next)) next++;
        if (n_act == 0) { if (next >= n) break; if (pos < b->pos[next]) pos = b->pos[next]; }
        while (next < n && (!crumble_oracle_in_pileup(b, next) || b->pos[next] <= pos)) {
            if (crumble_oracle_in_pileup(b, next)) {
                act[n_act].r = next; act[n_act].beg = b->pos[next];
                act[n_act].end = b->pos[next] + o_reflen(b, next);
                act[n_act].k = -1; act[n_act].x = act[n_act].y = 0; n_act++;
            }
            next++;
        }
        for (a = i = 0; i < n_act; i++) if (act[i].end > pos) act[a++] = act[i];
        n_act = a;
        if (!n_act) continue;
        for (i = 0; i < n_act; i++) {
            plp_t p;
            int lq;
            if (act[i].beg > pos) continue;
            o_resolve_cigar(b, &act[i], pos, &p);
            lq = o_lqseq(b, p.r);
            if (p.indel || p.is_del) {                     /* indel_only.c:95-103 */
                int x_s = p.qpos + 1 - D, x_e = p.qpos + 1 + D;
                if (x_s < 0) x_s = 0;
                if (x_e >= lq) x_e = lq - 1;
                for (x = x_s; x <= x_e; x++) mark[b->seq_off[p.r] + x] = 1;
            }
        }
        pos++;
    }
    for (r = 0; r < n; r++) {                              /* indel_only.c:106-121 */
        if (!crumble_oracle_in_pileup(b, r)) continue;
        for (x = 0; x < o_lqseq(b, r); x++)
            if (!mark[b->seq_off[r] + x]) qual_out[b->seq_off[r] + x] = (uint8_t)Q;
    }
    free(act); free(mark);
    return CRUMBLE_OK;
}
