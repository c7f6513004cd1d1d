/*
 * kernels.h -- the HIP kernels of the column engine (gfx950 / CDNA4).
 *
 * Re-decomposition of transcode()'s sequential loop body
 * (/root/reference/snp_score.c:1437-1983) into data-parallel passes that give
 * bit-identical results (DESIGN.md "Kernels"):
 *
 *   k_columns      one lane per pileup column: consensus (snp:533-797) with the
 *                  per-genotype FP64 sums taken in file order, column decisions
 *                  (snp:1575-1649) and the O(depth) heuristics (snp:1658-1773)
 *   scan_*         prefix sums over positions (running depth, work-list offsets)
 *   k_epochs/k_overdepth   running-average over-depth test (snp:1490,1673-1687)
 *   k_inshist      indel-length histogram test (snp:1708-1713,1777-1819)
 *   k_emit         ordered list of STR triggers per column (snp:1718-1730)
 *   k_str          find_STR/mask_LC_regions per trigger (str_finder.c:34-189,
 *                  snp:1230-1290)
 *   k_fold_*       min_pos/max_pos window state across columns (snp:1508-1511,
 *                  1741-1755), parallel over provably independent chains
 *   k_replay       per-read replay of the per-base loop (snp:1822-1920) and the
 *                  retirement/flush epilogue (snp:1939, 1093-1096)
 *   k_pblock       P-block (snp:803-834)
 *   k_stats        output-quality histogram
 *
 * All floating point here must round exactly like the reference's separate C
 * double operations: the translation unit is compiled with -ffp-contract=off
 * and without fast-math.
 */
#ifndef CRUMBLE_KERNELS_H
#define CRUMBLE_KERNELS_H

#include "dev_common.h"
#include <float.h>
#include <math.h>

#ifndef CRUMBLE_ABL
#define CRUMBLE_ABL 0
#endif
#ifdef CRUMBLE_EMU
#define UNIFORM(x) (x)
#else
#define UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#endif

typedef unsigned long long u64;

/* stats indices (mirror of the enum in include/crumble_hip.h) */
#define ST_COLUMNS 0
#define ST_HET_A 1
#define ST_HET_QUAL_A 2
#define ST_HOM_A 3
#define ST_HOM_QUAL_A 4
#define ST_DISCREP_A 5
#define ST_HET_B 6
#define ST_HET_QUAL_B 7
#define ST_HOM_B 8
#define ST_HOM_QUAL_B 9
#define ST_DISCREP_B 10
#define ST_DIFF 11
#define ST_INDEL 12
#define ST_INDEL_QUAL 13
#define ST_LOW_MQUAL_PERC 14
#define ST_CLIP_PERC 15
#define ST_INS_LEN_PERC 16
#define ST_INDEL_OV_PERC 17
#define ST_OVER_DEPTH 18
#define ST_READS 19
#define ST_BASES 20
#define ST_BASES_CHANGED 21
#define ST_TRIGGERS 22
#define ST_TOO_DEEP 23
#define ST_HIST0 32

__device__ __forceinline__ void stat_inc(u64 *stats, int which, bool cond)
{
    /* per-lane +1 under a predicate: hipcc folds these into one wave-level add */
    if (cond) atomicAdd(&stats[which], (u64)1);
}

/* Same, into a workgroup's LDS counters: one hot global word saturates at ~90
 * atomics/us (MI355X_MICROARCH.md "dequeue"), which a counter bumped once per
 * 64-column tile would hit; LDS counters are flushed once per workgroup. */
__device__ __forceinline__ void lds_inc(unsigned *cnt, int which, bool cond)
{
    if (cond) atomicAdd(&cnt[which], 1u);
}

/* ------------------------------------------------------------------------ */
/* fast_log2 / fast_exp (snp:491-518) */
__device__ __forceinline__ double dev_fast_log2(double val)
{
    int64_t x;
    memcpy(&x, &val, 8);
    const int log_2 = (int)((x >> 52) & 2047) - 1024;
    x &= ~(2047LL << 52);
    x += 1023LL << 52;
    memcpy(&val, &x, 8);
    val = ((-1.0f / 3) * val + 2) * val - 2.0f / 3;
    return val + log_2;
}
#define DEV_TENLOG2OVERLOG10 3.0103
__device__ __forceinline__ double dev_ph_log(double x) { return -DEV_TENLOG2OVERLOG10 * dev_fast_log2(x); }

__device__ __forceinline__ double dev_fast_exp(const DTables *T, double y)
{
    if (y >= -50 && y <= 50) return T->e_tab2[(int)(y * 10) + 500];
    if (y < -500) y = -500;
    if (y > 500) y = 500;
    return T->e_tab[(int)y + 500];
}

/* ------------------------------------------------------------------------ */
/* One pileup entry (the bam_pileup1_t fields crumble reads). */
struct Entry { int qpos, indel, is_del, is_refskip, is_head, is_tail; };

__device__ __forceinline__ int seg_find(const Seg *s, int n, int off)
{
    if (n <= 1) return 0;
    if (n <= 6) {
        int i = 0;
        while (i < n - 1 && off >= s[i].roff + seg_len(s[i])) i++;
        return i;
    }
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (off >= s[mid].roff + seg_len(s[mid])) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ void entry_from_seg(const Seg &sg, int off, int reflen, Entry &e)
{
    const int len = seg_len(sg);
    const unsigned t = seg_type(sg);
    const int o = off - sg.roff;
    e.is_del = (t != SEG_M);
    e.is_refskip = (t == SEG_N);
    e.qpos = (t == SEG_M) ? sg.qstart + o : sg.qstart;
    e.indel = (o == len - 1) ? sg.indel_after : 0;
    e.is_head = (off == 0);
    e.is_tail = (off == reflen - 1);
}

__device__ __forceinline__ void resolve_entry(const DBatch &B, int r, int apos, Entry &e)
{
    const int rp = B.pos[r];
    const int64_t sgo = B.seg_off[r];
    const int ns = (int)(B.seg_off[r + 1] - sgo);
    const int off = apos - rp;
    const int si = seg_find(B.segs + sgo, ns, off);
    entry_from_seg(B.segs[sgo + si], off, B.end[r] - rp, e);
}

__device__ __forceinline__ uint8_t cap_q(const DParams *P, uint8_t q)
{   /* cap_quality snp:1325-1332 */
    return (q > P->qcap && !P->preserve_qual[q]) ? (uint8_t)P->qcap : q;
}

/* keep-BED test (snp:1443-1463): the monotone cursor stops at the first region
 * (in bed_collapse order) whose end >= pos; that is the first index whose
 * prefix-max of `end` is >= pos. */
__device__ __forceinline__ int bed_preserve(const DBatch &B, int apos)
{
    if (!B.n_bed) return 0;
    int lo = 0, hi = B.n_bed;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (B.bed_pmax[mid] < apos) lo = mid + 1; else hi = mid;
    }
    if (lo < B.n_bed && B.bed_start[lo] <= apos && B.bed_end[lo] > apos) return 2;
    return 0;
}
/* The same for the 64 columns of a tile at once (k_columns2): the per-column search is ~13 dependent loads with
 * thousands of regions, and nearly every tile lies between two regions.  B.tile_bed[tile] (k_p0_tilebed, once per
 * batch) is the first region whose prefix-max of `end` reaches the tile's first column -- every column's own cursor
 * position is at or behind it; bed_preserve_from() walks on from there for one column. */
__device__ __forceinline__ int bed_preserve_from(const DBatch &B, int apos, int lo)
{
    while (lo < B.n_bed && B.bed_pmax[lo] < apos) lo++;
    if (lo < B.n_bed && B.bed_start[lo] <= apos && B.bed_end[lo] > apos) return 2;
    return 0;
}

/* ------------------------------------------------------------------------ */
struct ConsOut { int call, het_call, het_phred, phred; float discrep;
                 int disc_ge; };      /* cons_finalize_fast: discrep >= the active min_discrep, decided (0 / 1) without the value */

/* v_max_f64 / v_min_f64 as they are: fmax() / fmin() canonicalise both operands first (an extra v_max_f64 x, x each) --
 * the sums are never NaN */
#ifdef CRUMBLE_EMU
__device__ __forceinline__ double dev_max(double a, double b) { return a > b ? a : b; }
__device__ __forceinline__ double dev_min(double a, double b) { return a < b ? a : b; }
__device__ __forceinline__ double dev_rsq(double a) { return 1.0 / sqrt(a); }
#else
__device__ __forceinline__ double dev_max(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double dev_min(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double dev_rsq(double a) { return __builtin_amdgcn_rsq(a); }
#endif

/* The "speculate" half of calculate_consensus_pileup (snp:689-794).
 * Written without dynamically indexed arrays so that the 15 genotype sums stay
 * in registers: the reference's norm[j] (sum of all S[i], i != j, built from a
 * prefix and a suffix pass, snp:743-749) is prefix[j] + suffix[j] exactly,
 * because each norm[j] receives exactly those two addends starting from 0. */
/* sE: optional LDS copy of the halves of the exp tables that a non-positive argument can reach
 * (sE[0..500] = e_tab2[0..500] = exp(-50..0 step .1), sE[501..951] = e_tab[0..450] = exp(-500..-50)); the 15
 * arguments are S[j] - max(S) <= 0, so fast_exp (snp:491-503) is one table read each and all 15 go out together
 * (through dev_fast_exp they were 15 dependent global round trips per tile) */
#define CONS_NEXP 952
__device__ __forceinline__ void cons_finalize(const DTables *T, double *S, const double *sC,
                                              int depth, int nN, ConsOut &o, const double *sE = nullptr)
{
    const double min_e_exp = T->min_e_exp;            /* DBL_MIN_EXP*log(2)+1 (snp:540), host libm */
    double shift = -DBL_MAX, max = -DBL_MAX, max_het = -DBL_MAX;
    int call = 0, het_call = 0, ph;

    #pragma unroll
    for (int j = 0; j < 15; j++) {                     /* snp:707-725 */
        S[j] += T->lprior15[j];
        if (shift < S[j]) shift = S[j];
        if (j != 0 && j != 5 && j != 9 && j != 12 && j != 14) {
            if (max_het < S[j]) { max_het = S[j]; het_call = j; }
        } else {
            if (max < S[j]) { max = S[j]; call = j; }
        }
    }
    if (sE) {
        int ix[15];
        unsigned live = 0;                             /* bit j: S[j] > min_e_exp (else DBL_MIN, snp:739) */
        #pragma unroll
        for (int j = 0; j < 15; j++) {                 /* snp:736-741 with fast_exp's index arithmetic (snp:496-502) */
            const double y = S[j] - shift;             /* <= 0 */
            const double yc = y < -500.0 ? -500.0 : y;
            ix[j] = (y >= -50.0) ? (int)(y * 10) + 500 : 501 + (int)yc + 500;
            live |= (y > min_e_exp) ? (1u << j) : 0u;
        }
        #pragma unroll
        for (int j = 0; j < 15; j++) S[j] = sE[ix[j]];                       /* the 15 loads go out together */
        #pragma unroll
        for (int j = 0; j < 15; j++) S[j] = ((live >> j) & 1u) ? S[j] : DBL_MIN;
    } else {
        #pragma unroll
        for (int j = 0; j < 15; j++) {                 /* snp:736-741 */
            S[j] -= shift;
            const double e = dev_fast_exp(T, S[j]);
            S[j] = (S[j] > min_e_exp) ? e : DBL_MIN;
        }
    }
    /* norm[j] = (0 + sum of S[i], i < j, left to right) + (sum of S[i], i > j, right to left) (snp:743-749), needed
     * for j = call and j = het_call only: the running sums are captured as they pass those two indices (no
     * pre[] / suf[] arrays: 60 fewer live registers).  call is one of 0 5 9 12 14, het_call one of the others. */
    double tot1 = 0, tot2 = 0, pre_c = 0, pre_h = 0, suf_c = 0, suf_h = 0, s_het = 0;
    #pragma unroll
    for (int j = 0; j < 15; j++) {
        const bool hom = (j == 0 || j == 5 || j == 9 || j == 12 || j == 14);
        if (hom) { if (j == call) pre_c = tot1; }
        else     { if (j == het_call) pre_h = tot1; }
        tot1 += S[j];
    }
    #pragma unroll
    for (int j = 14; j >= 0; j--) {
        const bool hom = (j == 0 || j == 5 || j == 9 || j == 12 || j == 14);
        if (hom) { if (j == call) suf_c = tot2; }
        else     { if (j == het_call) { suf_h = tot2; s_het = S[j]; } }
        tot2 += S[j];
    }
    double nc = (0.0 + pre_c) + suf_c, nh = (0.0 + pre_h) + suf_h;
    if (depth && depth != nN) {
        /* map_sing / map_het (snp:547-558) */
        o.call = (call == 0) ? 0 : (call == 5) ? 1 : (call == 9) ? 2 : (call == 12) ? 3 : 4;
        if (nc == 0) nc = DBL_MIN;
        ph = (int)(dev_ph_log(nc) + .5);
        o.phred = ph > 255 ? 255 : (ph < 0 ? 0 : ph);
        o.het_call = (het_call < 5) ? het_call : (het_call < 9) ? het_call + 1 :
                     (het_call < 12) ? het_call + 3 : (het_call < 14) ? het_call + 6 : 24;
        if (nh == 0) nh = DBL_MIN;
        ph = (int)(DEV_TENLOG2OVERLOG10 * (dev_fast_log2(s_het) - dev_fast_log2(nh)) + .5);
        o.het_phred = ph;
        {
            const double m = sC[0] + sC[1] + sC[2] + sC[3] + sC[4];
            double c;
            if (o.het_phred > 0) {
                const int x = o.het_call % 5, y = o.het_call / 5;
                const double cx = (x == 0) ? sC[0] : (x == 1) ? sC[1] : (x == 2) ? sC[2] : (x == 3) ? sC[3] : sC[4];
                const double cy = (y == 0) ? sC[0] : (y == 1) ? sC[1] : (y == 2) ? sC[2] : (y == 3) ? sC[3] : sC[4];
                c = cx + cy;
            } else {
                const int x = o.call;
                c = (x == 0) ? sC[0] : (x == 1) ? sC[1] : (x == 2) ? sC[2] : (x == 3) ? sC[3] : sC[4];
            }
            o.discrep = (float)((m - c) / sqrt(m));
        }
    } else {
        o.call = 5; o.het_call = 0; o.het_phred = 0; o.phred = 0; o.discrep = 0;
    }
}

#define COL_CSWAP(a, b, c) { const double ta_ = (c) ? (b) : (a), tb_ = (c) ? (a) : (b); (a) = ta_; (b) = tb_; }
/* The same "speculate" half for the common column, straight from k_columns2's key-relative sums (kcol2.h header:
 * class 0 = the column's key base).  v0..v4 = S00 S01 S02 S03 S04 -- the five genotypes that contain the key base --
 * WITHOUT the priors; dmax = the largest of the ten other sums.  When every one of those ten is so far below the five
 * that its exp() term cannot move a bit of any partial sum (cons_fast_ok), the reference's result follows from the five
 * alone:
 *   - shift, call and het_call: a smaller value never wins a strict "<" scan (snp:707-725), so call = the key base's
 *     homozygote and het_call = the first maximum among the four key/x heterozygotes in the reference's order j;
 *   - norm[] (snp:743-749): a left-to-right FP64 sum of positive terms is unchanged by terms whose total stays below
 *     2^-55 of every other term: before the first large term a the small ones add up to T < ulp(a)/2, so T + a == a,
 *     and after it every P + d == P.  The ten terms are each <= exp(y + 1) (fast_exp truncates its index towards
 *     zero, snp:491-503; below -500 the clamp or DBL_MIN, snp:736-741, is smaller still), so
 *         max(y_other, -500) + 1 + ln 10 + 55 ln 2  <=  y_min_of_the_five          (41.5; the test uses 45)
 *     is sufficient; y_min >= -455 keeps the five clear of the clamp and of min_e_exp.
 * The five in the reference's order j are {a, key} for a = A C G T, then {key, *}: v[a ^ key] -- the homozygote lands
 * at a == key.  All priors of a kind are one value (snp:389-407: lprior15[0] for the homozygotes, [1] for the rest).
 * Everything else is the reference's arithmetic operation for operation. */
__device__ __forceinline__ bool cons_fast_ok(const DTables *T, double v0, double v1, double v2, double v3, double v4,
                                             double dmax, int depth, int nN)
{
    const double lph = T->lprior15[0], lpt = T->lprior15[1];
    const double a0 = v0 + lph, a1 = v1 + lpt, a2 = v2 + lpt, a3 = v3 + lpt, a4 = v4 + lpt;
    const double lmin = dev_min(dev_min(dev_min(a0, a1), dev_min(a2, a3)), a4), lmax = dev_max(dev_max(dev_max(a0, a1), dev_max(a2, a3)), a4);
    return depth > 0 && depth != nN && dmax + lph + 45.0 <= lmin && lmin - lmax >= -455.0;
}

#define CONS_SEL2(k1_, k2_, x0, x1, x2, x3) ((k2_) ? ((k1_) ? (x3) : (x2)) : ((k1_) ? (x1) : (x0)))
__device__ __forceinline__ void cons_finalize_fast(const DTables *T, double v0, double v1, double v2, double v3, double v4,
                                                   double c0, double c1, double c2, double c3, double c4,
                                                   unsigned key, ConsOut &o, const double *sE, double min_discrep, bool live)
{   /* called by every lane of the wave (the ballot below); live = the lane's column reaches the consensus */
    const bool k1 = (key & 1u) != 0u, k2 = (key & 2u) != 0u;
    const double lph = T->lprior15[0], lpt = T->lprior15[1];
    v0 += lph; v1 += lpt; v2 += lpt; v3 += lpt; v4 += lpt;                          /* snp:708 */
    /* into the reference's order: L[a] = v[a ^ key] (and the discrepancy sums with them) */
    COL_CSWAP(v0, v1, k1); COL_CSWAP(v2, v3, k1); COL_CSWAP(v0, v2, k2); COL_CSWAP(v1, v3, k2);
    COL_CSWAP(c0, c1, k1); COL_CSWAP(c2, c3, k1); COL_CSWAP(c0, c2, k2); COL_CSWAP(c1, c3, k2);
    const double shift = dev_max(dev_max(dev_max(v0, v1), dev_max(v2, v3)), v4);
    /* het_call: first maximum among the heterozygotes in order j (strict <, snp:715-718) */
    double max_het = -DBL_MAX; int h = 0;
    if (key != 0u && max_het < v0) { max_het = v0; h = 0; }
    if (key != 1u && max_het < v1) { max_het = v1; h = 1; }
    if (key != 2u && max_het < v2) { max_het = v2; h = 2; }
    if (key != 3u && max_het < v3) { max_het = v3; h = 3; }
    if (max_het < v4) { max_het = v4; h = 4; }
    /* fast_exp of the five (snp:491-503, 736-741): arguments in [-455, 0], so no clamp and never DBL_MIN */
    double e0, e1, e2, e3, e4;
#define CONS_FEXP(e_, v_) { const double y_ = (v_) - shift; const int i1_ = (int)(y_ * 10) + 500, i2_ = (int)y_ + 1001; \
                            e_ = sE[(y_ >= -50.0) ? i1_ : i2_]; }
    CONS_FEXP(e0, v0) CONS_FEXP(e1, v1) CONS_FEXP(e2, v2) CONS_FEXP(e3, v3) CONS_FEXP(e4, v4)
#undef CONS_FEXP
    /* norm[j] = (0 + sum of the terms before j, left to right) + (sum of the terms behind j, right to left) */
    const double p1 = 0.0 + e0, p2 = p1 + e1, p3 = p2 + e2, p4 = p3 + e3;
    const double u3 = 0.0 + e4, u2 = u3 + e3, u1 = u2 + e2, u0 = u1 + e1;
    const double pre_c = CONS_SEL2(k1, k2, 0.0, p1, p2, p3), suf_c = CONS_SEL2(k1, k2, u0, u1, u2, u3);
    const double pre_h = (h == 0) ? 0.0 : (h == 1) ? p1 : (h == 2) ? p2 : (h == 3) ? p3 : p4;
    const double suf_h = (h == 0) ? u0 : (h == 1) ? u1 : (h == 2) ? u2 : (h == 3) ? u3 : 0.0;
    const double s_het = (h == 0) ? e0 : (h == 1) ? e1 : (h == 2) ? e2 : (h == 3) ? e3 : e4;
    const double nc = (0.0 + pre_c) + suf_c, nh = (0.0 + pre_h) + suf_h;             /* > 0: no DBL_MIN substitution */
    int ph = (int)(dev_ph_log(nc) + .5);
    o.call = (int)key;
    o.phred = ph > 255 ? 255 : (ph < 0 ? 0 : ph);
    {   /* map_het (snp:551-558): genotype {x < y} -> 5 x + y */
        const int a = h, b = (int)key;
        o.het_call = a < b ? 5 * a + b : 5 * b + a;
    }
    ph = (int)(DEV_TENLOG2OVERLOG10 * (dev_fast_log2(s_het) - dev_fast_log2(nh)) + .5);
    o.het_phred = ph;
    {
        const double m = c0 + c1 + c2 + c3 + c4;                                    /* snp:776, in the reference's order */
        const double ck = CONS_SEL2(k1, k2, c0, c1, c2, c3);                        /* sumsC[call] */
        double c;
        if (ph > 0) {                                                               /* sumsC[het_call % 5] + sumsC[het_call / 5] */
            const double ch = (h == 0) ? c0 : (h == 1) ? c1 : (h == 2) ? c2 : (h == 3) ? c3 : c4;
            c = ck + ch;                                                            /* (two addends: the order is immaterial) */
        } else c = ck;
        /* discrep = (float)((m - c) / sqrt(m)) (snp:776-784) is only ever compared with min_discrep (snp:1604,1620,
         * 1624-1628).  (m - c) * rsq(m) is within 2^-22 of the quotient (v_rsq_f64: 2^29 ulp) and the reference's two
         * roundings and the float conversion move it by less than 2^-23, so outside a band of 1e-4 around the
         * threshold the comparison is decided; a wave with a lane inside the band divides */
        const double t = m - c, qa = t * dev_rsq(m);
        const bool sure_ge = qa >= min_discrep * (1.0 + 1e-4) && qa >= min_discrep * (1.0 - 1e-4);
        const bool sure_lt = qa < min_discrep * (1.0 - 1e-4) && qa < min_discrep * (1.0 + 1e-4);
        o.disc_ge = sure_ge ? 1 : 0;
        o.discrep = 0;
        if (__ballot(live && !(sure_ge || sure_lt))) {
            o.discrep = (float)(t / sqrt(m));
            o.disc_ge = o.discrep >= min_discrep ? 1 : 0;
        }
    }
}

/* ------------------------------------------------------------------------ */
/* k_columns: one lane per reference position; a wave owns 64 consecutive
 * positions (a tile) and walks the candidate reads of its tile in file order,
 * so each lane adds its column's contributions in exactly the reference's
 * order.  MQ selects CONS_MQUAL (mode B, snp:1535) vs plain (mode A, snp:1524).
 *
 * Memory behaviour: the per-read records of up to 64 candidate reads are
 * fetched with one coalesced load (lane i <- read i) and broadcast with
 * v_readlane; reads whose alignment is one match op ("simple", ~97 %) are
 * processed four at a time so their base/quality byte loads are in flight
 * together; the (mapq, qual) -> adjusted-quality table and the log-probability
 * tables live in LDS; workgroups are persistent (grid-stride over tiles) so the
 * 34 KB of LDS tables are staged once per workgroup. */
#ifdef CRUMBLE_EMU
#define READLANE(v, l) __shfl((int)(v), (int)(l))
#else
#define READLANE(v, l) __builtin_amdgcn_readlane((int)(v), (int)(l))
#endif

struct ColState {
    double S[15], sC[5];        /* the active consensus (mode B when enabled, else mode A)      */
    double S2[15], sC2[5];      /* mode A sums when both modes run (MODE 2); unused otherwise   */
    int depth, nN, n_plp, nref, low_mq, indel_cnt, clipped, n_overlap, n_isindel, has_ins;
    unsigned cp;
};

/* column decisions: snp:1438-1773 minus the passes that need cross-column state */
/* MODE 0: consensus A only (snp:1523, plain qualities); 1: B only (snp:1534, mapq-adjusted);
 * 2: both.  No preset enables A; -q does. */
/* PRE: the consensus result comes from the caller (cons_finalize_fast), st.S / st.sC are not read */
template <int MODE, bool PRE = false>
__device__ __forceinline__ void col_decide(ColState &st, const DBatch &B, const DParams *__restrict__ P,
                                           const DTables *__restrict__ T, int c, int apos, bool use_cons,
                                           uint32_t *__restrict__ colrec, int32_t *__restrict__ nplp_out,
                                           int32_t *__restrict__ score_out, int32_t *__restrict__ trigcnt, unsigned *stats,
                                           const double *sE = nullptr, const bool cnt = true, const ConsOut *pre = nullptr,
                                           const int bed_lo = -1)
{   /* bed_lo >= 0: the tile's keep-region cursor (B.tile_bed) */
    const int n_plp = st.n_plp;
    if (n_plp == 0) { colrec[c] = 0; nplp_out[c] = 0; score_out[c] = 0; trigcnt[c] = 0; return; }

    uint32_t rec = CR_EXISTS;
    int preserve = bed_lo >= 0 ? bed_preserve_from(B, apos, bed_lo) : bed_preserve(B, apos);   /* snp:1443-1463 */
    int perfect = 1, score = 0, tcnt = 0;
    nplp_out[c] = n_plp;
    if (st.nref == n_plp) {                            /* snp:1466-1472 */
        colrec[c] = rec | CR_ALLREFSKIP; score_out[c] = 0; trigcnt[c] = 0;
        return;
    }
    rec |= CR_COUNTED;
    lds_inc(stats, ST_COLUMNS, cnt && (true));
    if (n_plp > MAX_DEPTH_COL) {                       /* snp:1493-1500 */
        lds_inc(stats, ST_TOO_DEEP, cnt && (true));
        colrec[c] = rec | CR_TOODEEP; score_out[c] = 0; trigcnt[c] = 0;
        return;
    }

    int a1 = 6, a2 = 6;                                /* call1 = call2 = 0 */
    bool cond_score = false;
    if (use_cons) {
        const unsigned cp = st.cp;
        const unsigned lowb = cp & 31;                 /* bit2call (snp:1384): one-hot ACGT* -> 0..4, else 99 */
        const int b2c = (lowb == 1) ? 0 : (lowb == 2) ? 1 : (lowb == 4) ? 2 : (lowb == 8) ? 3 : (lowb == 16) ? 4 : 99;
        ConsOut coA, coB;
        int hA = 0, sA = 0, hB = 0, sB = 0;
        bool dgeA_ = false, dgeB_ = false;               /* discrep >= min_discrep (snp:1604,1620) */
        const bool useA = MODE != 1, useB = MODE != 0;
        if (useA) {                                    /* snp:1523-1532, 1576-1583, 1594-1609 */
            if (PRE) coA = *pre;
            else cons_finalize(T, MODE == 2 ? st.S2 : st.S, MODE == 2 ? st.sC2 : st.sC, st.depth, st.nN, coA, sE);
            const bool het = coA.het_phred > 0;
            if (het) { a1 = coA.het_call / 5; a2 = coA.het_call % 5; } else { a1 = a2 = coA.call; }
            hA = het ? coA.het_call : coA.call * 5 + coA.call;
            sA = het ? coA.het_phred : coA.phred;
            lds_inc(stats, ST_HET_A, cnt && (het)); lds_inc(stats, ST_HET_QUAL_A, cnt && (het && sA < P->min_qual_A));
            lds_inc(stats, ST_HOM_A, cnt && (!het)); lds_inc(stats, ST_HOM_QUAL_A, cnt && (!het && sA < P->min_qual_A));
            dgeA_ = PRE ? coA.disc_ge != 0 : coA.discrep >= P->min_discrep_A;
            lds_inc(stats, ST_DISCREP_A, cnt && dgeA_);
            if ((int)cp != (1 << coA.call)) perfect = 0;                 /* snp:1606 */
        }
        if (useB) {                                    /* snp:1534-1543, 1584-1591, 1610-1622 */
            if (PRE) coB = *pre;
            else cons_finalize(T, st.S, st.sC, st.depth, st.nN, coB, sE);
            const bool het = coB.het_phred > 0;
            if (het) { a1 = coB.het_call / 5; a2 = coB.het_call % 5; } else { a1 = a2 = coB.call; }
            hB = het ? coB.het_call : coB.call * 5 + coB.call;
            sB = het ? coB.het_phred : coB.phred;
            lds_inc(stats, ST_HET_B, cnt && (het)); lds_inc(stats, ST_HET_QUAL_B, cnt && (het && sB < P->min_qual_B));
            lds_inc(stats, ST_HOM_B, cnt && (!het)); lds_inc(stats, ST_HOM_QUAL_B, cnt && (!het && sB < P->min_qual_B));
            dgeB_ = PRE ? coB.disc_ge != 0 : coB.discrep >= P->min_discrep_B;
            lds_inc(stats, ST_DISCREP_B, cnt && dgeB_);
        }
        if (MODE == 2) lds_inc(stats, ST_DIFF, cnt && (hA != hB));                /* snp:1593 */
        preserve |= ((MODE == 2 && hA != hB) || (useA && sA < P->min_qual_A) || (useB && sB < P->min_qual_B)) ? 1 : 0;   /* snp:1624 */
        preserve |= ((useA && dgeA_) || (useB && dgeB_)) ? 1 : 0;   /* float vs double */
        if (useA && !preserve) {                                         /* snp:1633-1638 */
            if ((coA.het_phred <= 0 && b2c != coA.call) || (cp >> 8)) perfect = 0;
        }
        if (useB && !preserve) {                                         /* snp:1640-1645 */
            if ((coB.het_phred <= 0 && b2c != coB.call) || (cp >> 8)) perfect = 0;
        }
        cond_score = (useA && sA < P->min_indel_A) || (useB && sB < P->min_indel_B);   /* snp:1719-1720 */
        score = useB ? sB : sA;
    }
    if (P->perfect_col && !perfect) preserve = 1;                     /* snp:1648 */

    bool why_clip = false;
    bool keep = (double)st.low_mq > P->low_mqual_perc * (n_plp + .01);   /* snp:1668 */
    lds_inc(stats, ST_LOW_MQUAL_PERC, cnt && (keep));
    if ((st.clipped - 1.0) >= P->clip_perc * st.n_overlap) {          /* snp:1764 */
        keep = true;
        why_clip = true;
        lds_inc(stats, ST_CLIP_PERC, cnt && (true));
    }
    if (cond_score)                                                   /* snp:1718 */
        tcnt = (P->str_snp && preserve) ? (n_plp - st.nref) : st.n_isindel;
    lds_inc(stats, ST_INDEL, cnt && (st.indel_cnt > 0));                      /* snp:1761 */
    lds_inc(stats, ST_INDEL_QUAL, cnt && (cond_score && st.n_isindel > 0));   /* snp:1762 */
    if (tcnt && cnt) atomicAdd(&stats[ST_TRIGGERS], (unsigned)tcnt);

    rec |= (uint32_t)a1 | ((uint32_t)a2 << 3) | ((uint32_t)preserve << 6);
    if (perfect) rec |= CR_PERFECT;
    if (keep) rec |= CR_KEEP;
    if (why_clip) rec |= CR_R_CLIP;
    if (tcnt) rec |= CR_TRIG | CR_WIN;
    if (st.has_ins) rec |= CR_HASINS;
    if ((double)st.indel_cnt >= n_plp * P->indel_fract) rec |= CR_SCAN;  /* snp:1732 */
    colrec[c] = rec; score_out[c] = score; trigcnt[c] = tcnt;
}

/* is_head / is_tail bookkeeping (snp:1701-1706) is two columns per read, not a per-(column,
 * read) job: one thread per read adds them into a packed per-position word
 * (low 16 bits: entries that are head or tail; high 16 bits: "clipped" entries), which
 * col_decide() reads back.  Counts only matter on columns of depth <= MAX_DEPTH. */
__global__ void k_headtail(DBatch B, uint32_t *__restrict__ ht)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= B.n_reads) return;
    const ReadRec rr = B.rr[r];
    if (rr.end <= rr.pos) return;
    int q_head = rr.q0, q_tail = rr.q0 + (rr.end - rr.pos) - 1;
    bool head_act = true, tail_act = true;                 /* a ref-skip entry is not counted (snp:1696) */
    if (!(rr.info & RR_SIMPLE)) {
        const Seg *segs = B.segs + B.seg_off[r];
        const int ns = (int)(B.seg_off[r + 1] - B.seg_off[r]);
        const Seg s0 = segs[0], sl = segs[ns - 1];
        q_head = s0.qstart; head_act = seg_type(s0) != SEG_N;
        q_tail = (seg_type(sl) == SEG_M) ? sl.qstart + seg_len(sl) - 1 : sl.qstart; tail_act = seg_type(sl) != SEG_N;
    }
    const bool hc = q_head > 0, tc = q_tail + 1 < rr.lq;
    if (rr.end - rr.pos == 1) {                            /* head and tail on the same column */
        if (head_act) atomicAdd(&ht[rr.pos - B.p0], 1u + ((hc || tc) ? 0x10000u : 0u));
    } else {
        if (head_act) atomicAdd(&ht[rr.pos - B.p0], 1u + (hc ? 0x10000u : 0u));
        if (tail_act) atomicAdd(&ht[rr.end - 1 - B.p0], 1u + (tc ? 0x10000u : 0u));
    }
}

/* index 0..3 of a one-hot nt16 code (A=1 C=2 G=4 T=8) */
#define NT_IDX(code) (((int)(code) >> 1) - ((int)(code) >> 3))
/* k_columns numbers the base classes of a column relative to its first called base (class = idx ^ k).
 * XOR by k maps genotype {i,j} to {i^k, j^k}: bit 0 of k swaps A<->C and G<->T, bit 1 swaps A<->G and
 * C<->T ('*' stays).  The same involution brings the 15 sums and the 5 discrepancy sums back.
 * Genotype order: AA AC AG AT A* CC CG CT C* GG GT G* TT T* **  (snp:  the reference's 15 pairs). */
__device__ __forceinline__ void col_unpermute(double *S, double *sC, int k)
{
    const bool k1 = (k & 1) != 0, k2 = (k & 2) != 0;
    COL_CSWAP(S[0], S[5], k1); COL_CSWAP(S[2], S[7], k1); COL_CSWAP(S[3], S[6], k1); COL_CSWAP(S[9], S[12], k1);
    COL_CSWAP(S[4], S[8], k1); COL_CSWAP(S[11], S[13], k1); COL_CSWAP(sC[0], sC[1], k1); COL_CSWAP(sC[2], sC[3], k1);
    COL_CSWAP(S[0], S[9], k2); COL_CSWAP(S[5], S[12], k2); COL_CSWAP(S[1], S[10], k2); COL_CSWAP(S[3], S[6], k2);
    COL_CSWAP(S[4], S[11], k2); COL_CSWAP(S[8], S[13], k2); COL_CSWAP(sC[0], sC[2], k2); COL_CSWAP(sC[1], sC[3], k2);
}

/* the consensus adds of one pileup entry by base class (snp:656-683): genotype sums S[15] (AA AC AG AT A* CC
 * CG CT C* GG GT G* TT T* **), discrepancy sums sC[5]; MODE 2 keeps the mode-A sums beside them */
#define KC_CLS0 { sC[0] += om; S[0] += MM; S[1] += _M; S[2] += _M; S[3] += _M; S[4] += _M; \
                  if (MODE == 2) { sC2[0] += om2; S2[0] += MM2; S2[1] += _M2; S2[2] += _M2; S2[3] += _M2; S2[4] += _M2; } }
#define KC_CLS1 { sC[1] += om; S[1] += _M; S[5] += MM; S[6] += _M; S[7] += _M; S[8] += _M; \
                  if (MODE == 2) { sC2[1] += om2; S2[1] += _M2; S2[5] += MM2; S2[6] += _M2; S2[7] += _M2; S2[8] += _M2; } }
#define KC_CLS2 { sC[2] += om; S[2] += _M; S[6] += _M; S[9] += MM; S[10] += _M; S[11] += _M; \
                  if (MODE == 2) { sC2[2] += om2; S2[2] += _M2; S2[6] += _M2; S2[9] += MM2; S2[10] += _M2; S2[11] += _M2; } }
#define KC_CLS3 { sC[3] += om; S[3] += _M; S[7] += _M; S[10] += _M; S[12] += MM; S[13] += _M; \
                  if (MODE == 2) { sC2[3] += om2; S2[3] += _M2; S2[7] += _M2; S2[10] += _M2; S2[12] += MM2; S2[13] += _M2; } }
#define KC_CLSN { st.nN++;                                             /* not one of A C G T: snp:677-682 */ \
                  S[0] += MM; S[1] += MM; S[2] += MM; S[3] += MM; S[4] += _M; \
                  S[5] += MM; S[6] += MM; S[7] += MM; S[8] += _M; \
                  S[9] += MM; S[10] += MM; S[11] += _M; \
                  S[12] += MM; S[13] += _M; \
                  if (MODE == 2) { \
                      S2[0] += MM2; S2[1] += MM2; S2[2] += MM2; S2[3] += MM2; S2[4] += _M2; \
                      S2[5] += MM2; S2[6] += MM2; S2[7] += MM2; S2[8] += _M2; \
                      S2[9] += MM2; S2[10] += MM2; S2[11] += _M2; \
                      S2[12] += MM2; S2[13] += _M2; } }
#ifndef KCOL_NREC
#define KCOL_NREC 4               /* candidate records per loop iteration (their loads are in flight together) */
#endif
#ifndef KCOL_THREADS
#define KCOL_THREADS 256
#define KCOL_WAVES 4
#endif
/* (an `asm volatile` anywhere in this kernel makes the compiler treat memory as clobbered and turns the
 * scalar s_load_dwordx8 record loads into vector loads + readfirstlane: do not add one) */
/* MULTI: the batch has many multi-op reads (long reads): skip a read's other ops cheaply */
/* PERM: base classes numbered per column (see the accumulate block): fewer instructions when few base
 * calls disagree with their column, more when many do -- the host picks by the batch's qualities */
template <int MODE, bool MULTI, bool PERM>
__global__ void __launch_bounds__(KCOL_THREADS, MODE == 2 ? 3 : KCOL_WAVES)
k_columns(DBatch B, const DParams *__restrict__ P, const DTables *__restrict__ T,
          uint32_t *__restrict__ colrec, int32_t *__restrict__ nplp_out,
          int32_t *__restrict__ score_out, int32_t *__restrict__ trigcnt, const uint32_t *__restrict__ ht,
          u64 *stats, unsigned *__restrict__ tile_ctr)
{
    __shared__ double sTab[101 * 4];                       /* [aq] -> {MM, _M, 1 - q2p, -} : one 32-byte LDS row */
    constexpr bool MQ = MODE != 0;                          /* the active consensus uses CONS_MQUAL */
    __shared__ uint8_t sAdj[MQ ? 256 * 128 : 16];
    __shared__ unsigned sStat[32];
    /* composed row table of the batch's dominant mapq: sHot[q] = sTab[adj[hot_mq][q]], so that most
     * records need one dependent LDS round trip instead of two */
    __shared__ double sHot[MQ ? 128 * 4 : 4];
    const int hot_mq = MQ ? B.hot_mq : -1;
    if (threadIdx.x < 32) sStat[threadIdx.x] = 0;
    for (int i = threadIdx.x; i < 101; i += blockDim.x) {
        sTab[i * 4] = T->MMd[i]; sTab[i * 4 + 1] = T->_Md[i]; sTab[i * 4 + 2] = T->omq[i]; sTab[i * 4 + 3] = 0;
    }
    if (MQ) {
        /* adj[mapq][qual] for qual < 128 (input and capped qualities are < 128) */
        for (int i = threadIdx.x; i < 256 * 128 / 4; i += blockDim.x) {
            const int m = i >> 5, q4 = (i & 31) * 4;
            uint32_t w = *(const uint32_t *)&T->adj[m * 256 + q4], o = 0;
            #pragma unroll
            for (int k = 0; k < 4; k++) {
                uint32_t v = (w >> (8 * k)) & 255;
                v = v < 1 ? 1 : (v > 100 ? 100 : v);
                o |= v << (8 * k);
            }
            *(uint32_t *)&sAdj[m * 128 + q4] = o;
        }
        for (int q = threadIdx.x; q < 128; q += blockDim.x) {
            int a = T->adj[hot_mq * 256 + q];
            a = a < 1 ? 1 : (a > 100 ? 100 : a);
            sHot[q * 4] = T->MMd[a]; sHot[q * 4 + 1] = T->_Md[a]; sHot[q * 4 + 2] = T->omq[a]; sHot[q * 4 + 3] = 0;
        }
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave0 = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    const int nwaves = (int)(gridDim.x * (blockDim.x >> 6));
    const bool use_cons = MODE == 0 ? (P->min_qual_A != 0) : (P->min_qual_B != 0);
    const int qcap = P->qcap, min_mqual = P->min_mqual, any_pres = P->any_preserve;

    /* Work split.  A wave takes the next unit (B.unit_tiles consecutive tiles) from one counter.  Dynamic, because waves
     * do not run at equal speed (clock, memory channel, neighbours on the CU) -- a fixed split, however well balanced
     * on paper, left 15 % -- and in order, so that concurrently running waves share reads through L2. */
    (void)wave0; (void)nwaves;
    for (;;) {
    int t_begin, t_end;
    {
        int g = 0;
        if (lane == 0) g = (int)atomicAdd(tile_ctr, 1u);
        g = READLANE(g, 0);
        if (g >= B.n_units) break;
        t_begin = g * B.unit_tiles; t_end = t_begin + B.unit_tiles < B.n_tiles ? t_begin + B.unit_tiles : B.n_tiles;
    }
    for (int tile = t_begin; tile < t_end; tile++) {
        const int c = tile * 64 + lane;
        const int apos = B.p0 + c;
        const bool inr = c < B.L;
        const int lo = UNIFORM(B.stile_lo[tile]), hi = UNIFORM(B.stile_hi[tile]);

        ColState st;
        #pragma unroll
        for (int j = 0; j < 15; j++) st.S[j] = 0;
        #pragma unroll
        for (int j = 0; j < 5; j++) st.sC[j] = 0;
        if (MODE == 2) {
            #pragma unroll
            for (int j = 0; j < 15; j++) st.S2[j] = 0;
            #pragma unroll
            for (int j = 0; j < 5; j++) st.sC2[j] = 0;
        }
        st.depth = st.nN = st.n_plp = st.nref = st.low_mq = st.indel_cnt = 0;
        st.clipped = st.n_overlap = st.n_isindel = st.has_ins = 0;
        st.cp = 0;
        int kcode = 0;                                   /* nt16 code of the column's class-0 base; 0 = none yet */
        bool firstit = true;

        const int c0 = B.p0 + tile * 64, c1 = c0 + 63;
        int n_lq0 = 0;                                   /* covered entries of reads without SEQ */
        /* Record feed.  Short-read batches: the candidate range [lo, hi) is walked four records at a
         * time.  MULTI (long reads, many ops per read): the range holds every op of every read that
         * touches the tile, most of which lie elsewhere -- 64 candidates at a time are tested by the
         * lanes (one 8-byte load each), and only the ballot's set bits are walked, still in file order. */
        unsigned long long cand = 0;                     /* MULTI: untested-yet overlapping records of the chunk */
        int cbase = lo, jm = lo;
        for (int j = lo; MULTI || j < hi; j += MULTI ? 0 : KCOL_NREC) {
            int id[KCOL_NREC]; bool valid[KCOL_NREC];
            if (!MULTI) {
                #pragma unroll
                for (int u = 0; u < KCOL_NREC; u++) { valid[u] = true; id[u] = (j + u < hi) ? j + u : j; }
            } else {
                while (cand == 0 && jm < hi) {
                    const int idx = jm + lane;
                    bool ov = false;
                    if (idx < hi) { const int2 pe = *(const int2 *)&B.sr[idx]; ov = pe.y > c0 && pe.x <= c1; }
                    cand = __ballot(ov);
                    cbase = jm; jm += 64;
                }
                if (cand == 0) break;
                #pragma unroll
                for (int u = 0; u < KCOL_NREC; u++) {
                    valid[u] = cand != 0;
                    if (valid[u]) { id[u] = cbase + __builtin_ctzll(cand); cand &= cand - 1; }
                    else id[u] = id[0];
                }
            }
            /* four candidate records (one reference-consuming CIGAR op of one read each);
             * the index is wave-uniform, so these are scalar loads (s_load_dwordx8) */
            SegRec R[KCOL_NREC];
            #pragma unroll
            for (int u = 0; u < KCOL_NREC; u++) R[u] = B.sr[MULTI ? UNIFORM(id[u]) : id[u]];
            /* phase A: the byte loads of the four records go out together
             * (lanes that need no byte read byte 0: no branch between the loads) */
            uint8_t cc[KCOL_NREC], qq[KCOL_NREC];
            bool cvr[KCOL_NREC];                                  /* lane's column lies inside the record's span */
            #pragma unroll
            for (int u = 0; u < KCOL_NREC; u++) {
                /* branch-free on purpose (bitwise &, selects): control flow here costs more than it saves */
                const unsigned isMm = (SR_TYPE(R[u].info) == SEG_M) ? 0xffffffffu : 0u;   /* uniform */
                const int qp = R[u].qoff + (int)((unsigned)apos & isMm);
                cvr[u] = inr & ((unsigned)(apos - R[u].pos) < (unsigned)(R[u].end - R[u].pos));
                const bool cv = cvr[u] & (qp < R[u].lq);
                const uint8_t *bp = B.bases + R[u].so, *qp8 = B.quals + R[u].so;   /* wave-uniform bases */
                const unsigned a = cv ? (unsigned)qp : 0u;                          /* 32-bit lane offset */
                cc[u] = bp[a]; qq[u] = qp8[a];
            }
            /* class 0 of a column: the majority of its base calls among the first records of the tile
             * (ANY choice gives the same sums; a minority choice just sends the column's later records
             * through the rare block).  Columns not covered yet take their first call, below. */
            if (PERM && firstit) {                                             /* uniform */
                firstit = false;
                int a[KCOL_NREC], best = 0, bs = 0;
                #pragma unroll
                for (int u = 0; u < KCOL_NREC; u++) {
                    const int cd = cc[u] & 15;
                    a[u] = (cvr[u] && SR_TYPE(R[u].info) == SEG_M && ((0x0116u >> cd) & 1u)) ? cd : 0;
                }
                #pragma unroll
                for (int u = 0; u < KCOL_NREC; u++) {
                    int sc = 0;
                    #pragma unroll
                    for (int v = 0; v < KCOL_NREC; v++) sc += (a[v] == a[u]) ? 1 : 0;
                    sc = a[u] ? sc : 0;
                    if (sc > bs) { bs = sc; best = a[u]; }
                }
                kcode = best;
            }
            /* phase B: accumulate in file order.  Everything that depends only on the record is
             * wave-uniform (scalar), so the rare cases cost a scalar branch, not vector work; the
             * is_head/is_tail bookkeeping is not here at all (k_headtail). */
            #pragma unroll
            for (int u = 0; u < KCOL_NREC; u++) {
                if (MULTI ? !valid[u] : (j + u >= hi)) break;
                const int rp = R[u].pos, re = R[u].end, lq = R[u].lq;
                const unsigned info = R[u].info, ty = SR_TYPE(info);
                const int mq = (int)(info & 255);
                const bool cov = cvr[u];
                const bool isM = ty == SEG_M, isN = ty == SEG_N;               /* uniform */
                st.n_plp += cov ? 1 : 0;
                if (mq <= min_mqual) st.low_mq += cov ? 1 : 0;                 /* snp:1663 */
                if (!isM || R[u].indel_after != 0) {                           /* indel bookkeeping: rare ops */
                    const int e_indel = (apos == re - 1) ? R[u].indel_after : 0;
                    const bool isind = (e_indel != 0) || !isM;
                    const bool head = (info & SR_FIRST) && apos == rp, tail = (info & SR_LAST) && apos == re - 1;
                    st.indel_cnt += (cov && isind) ? 1 : 0;                    /* snp:1664 */
                    st.nref += (cov && isN) ? 1 : 0;                           /* snp:1468 */
                    st.n_isindel += (cov && !isN && isind) ? 1 : 0;
                    st.has_ins |= (cov && !isN && !head && !tail && e_indel > 0) ? 1 : 0;   /* snp:1708 */
                }
                if (lq == 0) n_lq0 += cov ? 1 : 0;
                if (isN || lq == 0 || !use_cons) continue;                     /* snp:589,594 (uniform) */
                /* consensus accumulation (snp:594-685).  Each base class is an exec-masked block of
                 * in-place FP64 adds (only the lanes that the record covers with that class take part;
                 * the block is skipped when there is none), so every genotype sum sees exactly its own
                 * addends, in file order, and the 20 running sums never leave their registers. */
                const int e_qpos = isM ? R[u].qoff + apos : R[u].qoff;
                int q = (int)qq[u], code = cc[u] & 15;
                if (!isM && e_qpos >= lq) { q = 0; code = 15; }                /* fence: read ends in D/N (uniform) */
                if (any_pres) q = (q > qcap && !P->preserve_qual[q]) ? qcap : q;   /* cap_quality snp:1325 */
                else          q = q > qcap ? qcap : q;
                if (any_pres && cov) {                                         /* snp:611-623 */
                    const int b = !isM ? 4 : (code == 1) ? 0 : (code == 2) ? 1 : (code == 4) ? 2 : (code == 8) ? 3 : 5;
                    const int e_indel = (apos == re - 1) ? R[u].indel_after : 0;
                    const uint8_t pv = P->preserve_qual[q];
                    if (pv) st.cp |= 1u << b;
                    if (pv > 1) st.cp |= (1u << b) << 8;
                    if (e_indel > 0)
                        for (int ins = 0; ins < e_indel; ins++) {
                            const int qi = e_qpos + 1 + ins;
                            if (qi < lq && P->preserve_qual[cap_q(P, B.quals[R[u].so + qi])]) st.cp |= 1u << 4;
                        }
                }
                /* snp:632-638; sAdj is stored already clamped to 1..100 (snp:641; fence Q5) */
                double MM, _M, om;
                if (MQ && mq == hot_mq) {                                      /* uniform: one lookup */
                    const double *row = &sHot[(q & 127) * 4];
                    MM = row[0]; _M = row[1]; om = row[2];
                } else {
                    const int aq = MQ ? (int)sAdj[mq * 128 + (q & 127)] : (q < 1 ? 1 : (q > 100 ? 100 : q));
                    MM = sTab[aq * 4]; _M = sTab[aq * 4 + 1]; om = sTab[aq * 4 + 2];
                }
                /* mode A next to mode B (MODE 2): the same entry with the plain (capped) quality */
                const int aq2 = q < 1 ? 1 : (q > 100 ? 100 : q);
                const double MM2 = MODE == 2 ? sTab[aq2 * 4] : 0.0, _M2 = MODE == 2 ? sTab[aq2 * 4 + 1] : 0.0, om2 = MODE == 2 ? sTab[aq2 * 4 + 2] : 0.0;
                const bool cm = cov && isM;                                    /* a base call (snp:603) */
                const bool cS = cov && !isM;                                   /* deletion: '*' (snp:609) */
                double *S = st.S, *sC = st.sC, *S2 = st.S2, *sC2 = st.sC2;
                /* !PERM: one exec-masked block per base class A C G T (+ N).  PERM: base classes are numbered
                 * PER COLUMN: class 0 is the column's key base (kcode), class pb = idx(base) ^ idx(kcode).
                 * An XOR is a permutation of {A,C,G,T}, the genotype table is symmetric under it, and
                 * col_unpermute() undoes it before the sums are read -- so every sum still receives exactly
                 * its own addends in file order, but a record whose covered base calls all agree with their
                 * columns is ONE block of adds for the whole wave instead of four (the lanes of a wave sit
                 * on different reference bases); disagreeing calls take the rare block.  Measured: -16 % on
                 * the 300x panel (0.2 % of the calls disagree), +2 % on the chr20 set (2.9 %). */
                if (!PERM) {
                    if (cm && code == 1) KC_CLS0
                    if (cm && code == 2) KC_CLS1
                    if (cm && code == 4) KC_CLS2
                    if (cm && code == 8) KC_CLS3
                    if (cm && !((0x0116u >> code) & 1u)) KC_CLSN
                } else {
                    if (cm && kcode == 0) kcode = ((0x0116u >> code) & 1u) ? code : 1;
                    if (cm && code == kcode) KC_CLS0
                    if (cm && code != kcode) {                                 /* mismatch or N: the rare block */
                        const int pb = ((0x0116u >> code) & 1u) ? (NT_IDX(code) ^ NT_IDX(kcode)) : 0;
                        const unsigned pbm = 1u << pb;                         /* (bit tests: no switch, no branch tree) */
                        if (pbm & 2u) KC_CLS1
                        if (pbm & 4u) KC_CLS2
                        if (pbm & 8u) KC_CLS3
                        if (pbm & 1u) KC_CLSN
                    }
                }
                if (cS) { sC[4] += om; S[4] += _M; S[8] += _M; S[11] += _M; S[13] += _M; S[14] += MM;
                          if (MODE == 2) { sC2[4] += om2; S2[4] += _M2; S2[8] += _M2; S2[11] += _M2; S2[13] += _M2; S2[14] += MM2; } }
            }
        }
        if (inr) {
            const uint32_t h = ht[c];
            st.depth = st.n_plp - st.nref - n_lq0;       /* entries that reached the sums (snp:685) */
            st.clipped = (int)(h >> 16);
            st.n_overlap = st.n_plp - st.nref - (int)(h & 0xffff);
        }
        if (PERM && kcode > 1) {                    /* back to the reference's A C G T numbering */
            const int k = NT_IDX(kcode);
            col_unpermute(st.S, st.sC, k);
            if (MODE == 2) col_unpermute(st.S2, st.sC2, k);
        }
        if (inr) {
            if (c >= B.col_lo && c < B.col_hi)
                col_decide<MODE>(st, B, P, T, c, apos, use_cons, colrec, nplp_out, score_out, trigcnt, sStat, nullptr, c < B.stat_hi);
            else { colrec[c] = 0; nplp_out[c] = 0; score_out[c] = 0; trigcnt[c] = 0; }
        }
    }
    }
    __syncthreads();
    if (threadIdx.x < 32 && sStat[threadIdx.x]) atomicAdd(&stats[threadIdx.x], (u64)sStat[threadIdx.x]);
}

#include "kcol2.h"

/* Work-list kernels take their length either from the host (n, `dev` null) or from device memory (the scan's totals,
 * clamped to the capacity n the lists were reserved with) and stride over it with whatever grid they were given: the
 * run's launches then need no size read-back in the middle (crumble_hip_run's fast path, engine.hip).  `abort_`: the
 * scan found the speculative capacities too small (or a too-deep column): the lists are incomplete, do nothing. */
#define LIST_N(n_, dev_) ((dev_) ? ((*(dev_) < (n_)) ? *(dev_) : (n_)) : (n_))
#define LIST_ABORTED(abort_) ((abort_) && *(abort_))

/* ------------------------------------------------------------------------ */
/* Chunked device-wide exclusive scan of ScanVal over positions.            */
#ifndef SCAN_CHUNK
#define SCAN_CHUNK 8192               /* (2048: scan stage 0.69 ms on the chr20 set, 4096: 0.63, 8192: 0.59) */
#define SCAN_CHUNK_SMALL 1024         /* small contigs: one workgroup walking 8192 columns in eight dependent rounds was a tenth of a 5 kb contig's run */
#define SCAN_CHUNK_OF(L_) ((L_) <= (1 << 18) ? SCAN_CHUNK_SMALL : SCAN_CHUNK)
#endif

__device__ __forceinline__ void scanval_of(uint32_t rec, int32_t nplp, int32_t tc, ScanVal &v)
{
    const bool counted = (rec & CR_COUNTED) != 0;
    v.depth = counted ? nplp : 0;
    v.cols = counted ? 1 : 0;
    v.trigcols = (rec & CR_TRIG) ? 1 : 0;
    v.items = tc;
    v.ins = (rec & CR_HASINS) ? 1 : 0;
    v.toodeep = (rec & CR_TOODEEP) ? 1 : 0;
    v.pad = 0;
}
__device__ __forceinline__ void scanval_add(ScanVal &a, const ScanVal &b)
{ a.depth += b.depth; a.cols += b.cols; a.trigcols += b.trigcols; a.items += b.items; a.ins += b.ins; a.toodeep += b.toodeep; }

/* wave-level inclusive scan of a ScanVal, then block-level via LDS.  Round 5: the six fields go through the DPP scans of
 * kcol2.h (VALU only, six dependent steps of a few cycles) instead of __shfl_up ladders -- 36 ds_bpermute round trips per
 * scan, six of them dependent -- in kernels that do one such scan per 1024 columns between two barriers */
#ifdef CRUMBLE_EMU
__device__ __forceinline__ long long kc2_scan_add64(long long v)
{ const int ln = threadIdx.x & 63; for (int d = 1; d < 64; d <<= 1) { const long long o = __shfl_up(v, d); if (ln >= d) v += o; } return v; }
#else
__device__ __forceinline__ long long kc2_scan_add64(long long v)
{
#define KC2_STEP64(ctrl_, rows_) { const int lo_ = KC2_DPP(0, (int)v, ctrl_, rows_), hi_ = KC2_DPP(0, (int)(v >> 32), ctrl_, rows_); \
                                   v += (long long)(((unsigned long long)(unsigned)hi_ << 32) | (unsigned)lo_); }
    KC2_STEP64(0x111, 0xf) KC2_STEP64(0x112, 0xf) KC2_STEP64(0x114, 0xf) KC2_STEP64(0x118, 0xf) KC2_STEP64(0x142, 0xa) KC2_STEP64(0x143, 0xc)
#undef KC2_STEP64
    return v;
}
#endif
__device__ __forceinline__ ScanVal scanval_wave_incl(ScanVal v, int lane)
{
    (void)lane;
    v.depth = kc2_scan_add64(v.depth);
    v.cols = kc2_scan_add(v.cols); v.trigcols = kc2_scan_add(v.trigcols); v.items = kc2_scan_add(v.items);
    v.ins = kc2_scan_add(v.ins); v.toodeep = kc2_scan_add(v.toodeep);
    return v;
}
/* the same for per-position values, whose four flag fields are 0/1: their prefix is a popcount of
 * the ballot below the lane; only depth and items need the shuffle ladder */
__device__ __forceinline__ ScanVal scanval_wave_incl01(ScanVal v, int lane)
{
    const unsigned long long le = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
    v.cols = __popcll(__ballot(v.cols != 0) & le);
    v.trigcols = __popcll(__ballot(v.trigcols != 0) & le);
    v.ins = __popcll(__ballot(v.ins != 0) & le);
    v.toodeep = __popcll(__ballot(v.toodeep != 0) & le);
    v.depth = kc2_scan_add64(v.depth); v.items = kc2_scan_add(v.items);
    return v;
}

/* one workgroup reduces one chunk of SCAN_CHUNK consecutive positions (coalesced) */
__global__ void __launch_bounds__(256)
k_scan_reduce(int L, const uint32_t *colrec, const int32_t *nplp,
              const int32_t *trigcnt, ScanVal *partial, int nchunks, int chunk)
{
    __shared__ ScanVal ws[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int base = blockIdx.x * chunk;
    ScanVal acc = {0, 0, 0, 0, 0, 0, 0}, v;
    for (int k = 0; k < chunk / 1024; k++) {                /* four consecutive columns per thread: 16-byte loads */
        const int i = base + k * 1024 + 4 * (int)threadIdx.x;
        if (i + 3 < L) {
            const uint4 r4 = *(const uint4 *)(colrec + i);
            const int4 n4 = *(const int4 *)(nplp + i), t4 = *(const int4 *)(trigcnt + i);
            scanval_of(r4.x, n4.x, t4.x, v); scanval_add(acc, v);
            scanval_of(r4.y, n4.y, t4.y, v); scanval_add(acc, v);
            scanval_of(r4.z, n4.z, t4.z, v); scanval_add(acc, v);
            scanval_of(r4.w, n4.w, t4.w, v); scanval_add(acc, v);
        } else
            for (int u = 0; u < 4; u++) if (i + u < L) { scanval_of(colrec[i + u], nplp[i + u], trigcnt[i + u], v); scanval_add(acc, v); }
    }
    acc = scanval_wave_incl(acc, lane);
    if (lane == 63) ws[w] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        ScanVal t = ws[0];
        for (int j = 1; j < (int)(blockDim.x >> 6); j++) scanval_add(t, ws[j]);
        partial[blockIdx.x] = t;
    }
}

/* one workgroup: exclusive scan of the chunk partials; totals -> partial[nchunks].  The partials are taken
 * in tiles of 1024 (thread t <- entry base + t: coalesced), each tile scanned by the workgroup (wave shuffles
 * + the 16 wave totals in LDS), the carry running from tile to tile. */
__global__ void __launch_bounds__(1024)
k_scan_partials(ScanVal *partial, int nchunks, int cap_trig, int cap_items, int cap_ins, int *spec_abort, ScanVal *totals_copy)
{   /* the fast path (null otherwise): *spec_abort <- 1 when a total exceeds its capacity or a column is too deep;
     * *totals_copy <- the totals, next to the run's flags for the host's one read-back at the end of the run */
    __shared__ ScanVal ws[16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (int)(blockDim.x >> 6);
    ScanVal carry = {0, 0, 0, 0, 0, 0, 0};
    for (int base = 0; base < nchunks; base += (int)blockDim.x) {
        const int i = base + (int)threadIdx.x;
        ScanVal v = {0, 0, 0, 0, 0, 0, 0};
        if (i < nchunks) v = partial[i];
        const ScanVal inc = scanval_wave_incl(v, lane);
        if (lane == 63) ws[w] = inc;
        __syncthreads();
        ScanVal run = carry;
        for (int j = 0; j < w; j++) scanval_add(run, ws[j]);
        /* exclusive prefix of this entry = run + inc - v */
        run.depth += inc.depth - v.depth; run.cols += inc.cols - v.cols; run.trigcols += inc.trigcols - v.trigcols;
        run.items += inc.items - v.items; run.ins += inc.ins - v.ins; run.toodeep += inc.toodeep - v.toodeep;
        if (i < nchunks) partial[i] = run;
        for (int j = 0; j < nw; j++) scanval_add(carry, ws[j]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partial[nchunks] = carry;                             /* the grand total */
        if (spec_abort) {
            if (carry.trigcols > cap_trig || carry.items > cap_items || carry.ins > cap_ins || carry.toodeep != 0) *spec_abort = 1;
            *totals_copy = carry;
        }
    }
}

__global__ void __launch_bounds__(256)
k_scan_apply(int L, const uint32_t *colrec, const int32_t *nplp,
             const int32_t *trigcnt, const ScanVal *partial, int nchunks,
             int64_t *dpre, int32_t *col_ord, int32_t *trig_ord,
             int32_t *item_off, int32_t *ins_ord, int chunk)
{
    __shared__ ScanVal ws[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int base = blockIdx.x * chunk;
    ScanVal carry = partial[blockIdx.x];
    for (int k = 0; k < chunk / 256; k++) {
        const int i = base + k * 256 + (int)threadIdx.x;
        ScanVal v = {0, 0, 0, 0, 0, 0, 0};
        if (i < L) scanval_of(colrec[i], nplp[i], trigcnt[i], v);
        ScanVal inc = scanval_wave_incl01(v, lane);
        __syncthreads();                               /* ws reuse */
        if (lane == 63) ws[w] = inc;
        __syncthreads();
        ScanVal run = carry;
        for (int j = 0; j < w; j++) scanval_add(run, ws[j]);
        if (i < L) {
            dpre[i] = run.depth + inc.depth - v.depth; col_ord[i] = run.cols + inc.cols - v.cols;
            /* only trigger / insertion columns are ever looked up in these three (k_overdepth) */
            if (v.trigcols) { trig_ord[i] = run.trigcols + inc.trigcols - v.trigcols; item_off[i] = run.items + inc.items - v.items; }
            if (v.ins) ins_ord[i] = run.ins + inc.ins - v.ins;
        }
        for (int j = 0; j < (int)(blockDim.x >> 6); j++) scanval_add(carry, ws[j]);
    }
}

/* ------------------------------------------------------------------------ */
/* Running depth average (snp:1490-1491, 1684-1687): total_col grows by one per
 * counted column and both sums are halved after the test of the first column
 * (not a too-deep one) that finds total_col > 2^20.  One thread walks the few
 * hundred halving points; every column then evaluates its own test. */
__device__ __forceinline__ int ord_to_pos(const int32_t *col_ord, const uint32_t *colrec, int L, int64_t ord)
{   /* position of the counted column with ordinal `ord` */
    int lo = 0, hi = L;                 /* first i with col_ord[i+1-ish] > ord */
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        /* inclusive count at mid */
        int64_t inc = col_ord[mid] + ((colrec[mid] & CR_COUNTED) ? 1 : 0);
        if (inc <= ord) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ void k_epochs(int L, const uint32_t *colrec, const int32_t *nplp, const int32_t *col_ord,
                         const int64_t *dpre, const ScanVal *totals, Epoch *ep, int *n_ep, int64_t tc0, int64_t td0)
{
    if (blockIdx.x || threadIdx.x) return;
    const int64_t ncols = totals->cols;
    int64_t tc = tc0, td = td0, ord = 0;            /* (tc0, td0): the state a window of a contig starts from */
    int n = 0;
    while (ord < ncols && n < MAX_EPOCHS - 1) {
        int p0 = ord_to_pos(col_ord, colrec, L, ord);
        ep[n].ord_start = ord; ep[n].tc_start = tc; ep[n].td_start = td; ep[n].dpre_start = dpre[p0];
        n++;
        int64_t j = ord + ((tc + 1 > 1048576) ? 0 : (1048576 - tc));   /* first ordinal with tc' > 2^20 */
        if (j >= ncols) break;
        int pj = ord_to_pos(col_ord, colrec, L, j);
        while (j < ncols && (colrec[pj] & CR_TOODEEP)) {               /* too deep: no halving (goto) */
            j++;
            if (j < ncols) pj = ord_to_pos(col_ord, colrec, L, j);
        }
        if (j >= ncols) break;
        const int64_t tcj = tc + (j - ord + 1);
        const int64_t tdj = td + (dpre[pj] + nplp[pj] - ep[n - 1].dpre_start);
        tc = tcj >> 1; td = tdj >> 1; ord = j + 1;
    }
    *n_ep = n;
}

/* Same result when no column is deeper than MAX_DEPTH (the usual case): the halving
 * points are then at fixed ordinals (2^20, then every 2^19+1 columns), so each epoch's
 * prefix lookup is independent; only the tiny td recurrence is sequential. */
__global__ void __launch_bounds__(256)
k_epochs_par(int L, const uint32_t *colrec, const int32_t *col_ord, const int64_t *dpre,
             const ScanVal *totals, Epoch *ep, int *n_ep, int64_t tc0, int64_t td0)
{
    /* (tc0 <= 2^20, td0): the state a window of a contig starts from; the first halving column is the one that
     * brings total_col to 2^20 + 1 */
    const int64_t ncols = totals->cols, first = 1048577 - tc0;
    int n = 0;
    if (ncols > 0) n = 1 + (int)((ncols > first) ? (ncols - first + 524288) / 524289 : 0);
    if (n > MAX_EPOCHS - 1) n = MAX_EPOCHS - 1;
    for (int k = threadIdx.x; k < n; k += blockDim.x) {
        const int64_t ord = (k == 0) ? 0 : first + (int64_t)(k - 1) * 524289;
        const int p = ord_to_pos(col_ord, colrec, L, ord);
        ep[k].ord_start = ord; ep[k].tc_start = (k == 0) ? tc0 : 524288; ep[k].dpre_start = dpre[p]; ep[k].td_start = td0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t td = td0;
        for (int k = 0; k + 1 < n; k++) {
            td = (td + ep[k + 1].dpre_start - ep[k].dpre_start) >> 1;
            ep[k + 1].td_start = td;
        }
        *n_ep = n;
    }
}

/* (total_col, total_depth) before the column at index `cut` (snp:1490-1491, 1684-1687): what the next window of the
 * contig starts from.  The counted column before the cut is looked up in its epoch, then the halving rule is applied
 * to it as the reference does after its test. */
__global__ void k_depth_state_at(int L, int cut, const uint32_t *colrec, const int32_t *nplp, const int32_t *col_ord,
                                 const int64_t *dpre, const ScanVal *totals, const Epoch *ep, const int *n_ep,
                                 int64_t tc0, int64_t td0, int64_t *out)
{
    if (blockIdx.x || threadIdx.x) return;
    const int64_t o = cut < L ? (int64_t)col_ord[cut] : totals->cols;    /* counted columns before the cut */
    int64_t tc = tc0, td = td0;
    if (o > 0) {
        const int p = ord_to_pos(col_ord, colrec, L, o - 1);
        const int nep = *n_ep;
        int e = 0;
        while (e + 1 < nep && ep[e + 1].ord_start <= o - 1) e++;
        tc = ep[e].tc_start + (o - 1 - ep[e].ord_start + 1);
        td = ep[e].td_start + (dpre[p] + nplp[p] - ep[e].dpre_start);
        if (tc > 1048576 && !(colrec[p] & CR_TOODEEP)) { tc >>= 1; td >>= 1; }
    }
    out[0] = tc; out[1] = td;
}

/* over-depth test per column (snp:1673-1681) + compaction of the trigger and
 * histogram columns. */
/* the test of one counted, not too deep column: sets CR_KEEP | CR_R_DEEP in its record when it fires */
__device__ __forceinline__ void overdepth_test(int i, uint32_t rec, int64_t ord, int n, int64_t dp, uint32_t *colrec,
                                               const Epoch *ep, int nep, double over_depth, u64 *stats, int stat_hi)
{
    /* last epoch with ord_start <= ord.  Epochs are ~2^20 columns first and ~2^19 afterwards, so an
     * arithmetic guess is right or off by one: two probes instead of a dependent binary search */
    int lo = ord < (1 << 20) ? 0 : 1 + (int)((ord - (1 << 20)) >> 19);
    if (lo > nep - 1) lo = nep - 1;
    while (lo + 1 < nep && ep[lo + 1].ord_start <= ord) lo++;
    while (lo > 0 && ep[lo].ord_start > ord) lo--;
    const Epoch e = ep[lo];
    const int64_t total_col = e.tc_start + (ord - e.ord_start + 1);
    const int64_t total_depth = e.td_start + (dp + n - e.dpre_start);
    if ((double)((int64_t)n * (total_col + 1)) > over_depth * (double)(total_depth + 1)) {
        colrec[i] = rec | CR_KEEP | CR_R_DEEP;
        stat_inc(stats, ST_OVER_DEPTH, i < stat_hi);
    }
}
__device__ __forceinline__ void overdepth_col(int i, uint32_t rec, int64_t ord, int n, int64_t dp, uint32_t *colrec,
                                              const int32_t *trig_ord, const int32_t *item_off, const int32_t *ins_ord,
                                              const Epoch *ep, int nep, double over_depth, int32_t *trigcol_pos,
                                              int32_t *trigcol_off, int32_t *inscol_pos, u64 *stats, int stat_hi)
{
    if (!(rec & CR_COUNTED) || (rec & CR_TOODEEP)) return;
    overdepth_test(i, rec, ord, n, dp, colrec, ep, nep, over_depth, stats, stat_hi);
    if (rec & CR_TRIG) { trigcol_pos[trig_ord[i]] = i; trigcol_off[trig_ord[i]] = item_off[i]; }
    if (rec & CR_HASINS) inscol_pos[ins_ord[i]] = i;
}
/* four consecutive columns per thread (16-byte loads of the per-position arrays) */
__global__ void __launch_bounds__(256)
k_overdepth(int L, uint32_t *colrec, const int32_t *nplp, const int32_t *col_ord,
            const int64_t *dpre, const int32_t *trig_ord, const int32_t *item_off,
            const int32_t *ins_ord, const Epoch *ep, const int *n_ep,
            const DParams *P, int32_t *trigcol_pos, int32_t *trigcol_off,
            int32_t *inscol_pos, u64 *stats, int stat_hi)
{
    const int i0 = (int)(blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i0 >= L) return;
    const int nep = *n_ep;
    const double od = P->over_depth;
    if (i0 + 3 < L) {
        const uint4 r4 = *(const uint4 *)(colrec + i0);
        if (!((r4.x | r4.y | r4.z | r4.w) & CR_COUNTED)) return;
        const int4 o4 = *(const int4 *)(col_ord + i0), n4 = *(const int4 *)(nplp + i0);
        const longlong2 d01 = *(const longlong2 *)(dpre + i0), d23 = *(const longlong2 *)(dpre + i0 + 2);
        overdepth_col(i0,     r4.x, o4.x, n4.x, d01.x, colrec, trig_ord, item_off, ins_ord, ep, nep, od, trigcol_pos, trigcol_off, inscol_pos, stats, stat_hi);
        overdepth_col(i0 + 1, r4.y, o4.y, n4.y, d01.y, colrec, trig_ord, item_off, ins_ord, ep, nep, od, trigcol_pos, trigcol_off, inscol_pos, stats, stat_hi);
        overdepth_col(i0 + 2, r4.z, o4.z, n4.z, d23.x, colrec, trig_ord, item_off, ins_ord, ep, nep, od, trigcol_pos, trigcol_off, inscol_pos, stats, stat_hi);
        overdepth_col(i0 + 3, r4.w, o4.w, n4.w, d23.y, colrec, trig_ord, item_off, ins_ord, ep, nep, od, trigcol_pos, trigcol_off, inscol_pos, stats, stat_hi);
    } else {
        for (int i = i0; i < L; i++)
            overdepth_col(i, colrec[i], col_ord[i], nplp[i], dpre[i], colrec, trig_ord, item_off, ins_ord, ep, nep, od,
                          trigcol_pos, trigcol_off, inscol_pos, stats, stat_hi);
    }
}

/* ---- the same two steps without the per-column prefix arrays (whole contigs without a too-deep column: the usual
 * case).  k_scan_apply wrote 12 B per column (depth prefix, ordinal) that only k_epochs_par and k_overdepth read
 * back.  Here the epochs come from the chunk prefixes plus a walk inside one chunk each, and the pass that knows
 * every column's prefixes applies the over-depth test and fills the trigger / insertion column lists itself. */

/* one wave per epoch k >= 1: the counted column with ordinal ord_k and the depth prefix in front of it */
/* number of epochs of a whole contig with `ncols` counted columns that starts from total_col tc0 (no too-deep column) */
__device__ __forceinline__ int epochs_count(int64_t ncols, int64_t tc0)
{
    const int64_t first = 1048577 - tc0;
    int n = 0;
    if (ncols > 0) n = 1 + (int)((ncols > first) ? (ncols - first + 524288) / 524289 : 0);
    return n > MAX_EPOCHS - 1 ? MAX_EPOCHS - 1 : n;
}
__global__ void __launch_bounds__(64)
k_epochs_find(int L, const uint32_t *__restrict__ colrec, const int32_t *__restrict__ nplp,
              const ScanVal *__restrict__ partial, int nchunks, Epoch *__restrict__ ep, int64_t tc0, int chunk)
{
    const int k = blockIdx.x + 1, lane = threadIdx.x;
    if (k >= epochs_count(partial[nchunks].cols, tc0)) return;   /* (the grid may be sized for the most epochs L columns can have) */
    const int64_t ord = (1048577 - tc0) + (int64_t)(k - 1) * 524289;
    int lo = 0, hi = nchunks;                              /* last chunk whose exclusive column count is <= ord */
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((int64_t)partial[mid].cols <= ord) lo = mid; else hi = mid; }
    const int base = lo * chunk, per = chunk / 64;
    /* lane l takes `per` consecutive columns: its sums, the wave's exclusive prefix, then the lane that holds the
     * ordinal walks its slice */
    int cnt = 0;
    int64_t dep = 0;
    for (int j = 0; j < per; j++) {
        const int i = base + lane * per + j;
        if (i < L && (colrec[i] & CR_COUNTED)) { cnt++; dep += nplp[i]; }
    }
    int pc = cnt;
    int64_t pd = dep;
    #pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int oc = __shfl_up(pc, d); const int64_t od = __shfl_up(pd, d);
        if (lane >= d) { pc += oc; pd += od; }
    }
    const int64_t c0 = (int64_t)partial[lo].cols + (pc - cnt);      /* counted columns before this lane's slice */
    int64_t d0 = (int64_t)partial[lo].depth + (pd - dep);
    if (ord >= c0 && ord < c0 + cnt) {
        int64_t c = c0;
        for (int j = 0; j < per; j++) {
            const int i = base + lane * per + j;
            if (i < L && (colrec[i] & CR_COUNTED)) {
                if (c == ord) { ep[k].ord_start = ord; ep[k].tc_start = 524288; ep[k].dpre_start = d0; ep[k].td_start = 0; break; }
                c++; d0 += nplp[i];
            }
        }
    }
}
__global__ void k_epochs_td(Epoch *ep, int n, int *n_ep, int64_t tc0, int64_t td0, const ScanVal *totals)
{
    if (blockIdx.x || threadIdx.x) return;
    if (totals) n = epochs_count(totals->cols, tc0);           /* (fast path: the count is on the device) */
    ep[0].ord_start = 0; ep[0].tc_start = tc0; ep[0].dpre_start = 0; ep[0].td_start = td0;
    int64_t td = td0;
    for (int k = 0; k + 1 < n; k++) {
        td = (td + ep[k + 1].dpre_start - ep[k].dpre_start) >> 1;
        ep[k + 1].td_start = td;
    }
    *n_ep = n;
}

/* k_scan_apply + k_overdepth in one pass over the columns */
/* (six waves per SIMD: left alone the compiler takes 120 VGPRs -- four waves -- for a kernel that waits on its per-column epoch
 * look-ups; at 79, without a spill, chr20 0.318 -> 0.276 ms, config 4 1.23 -> 1.04) */
#ifndef CRUMBLE_EMU
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 6)))
#else
__global__ void __launch_bounds__(256)
#endif
k_scan_overdepth(int L, uint32_t *colrec, const int32_t *nplp, const int32_t *trigcnt, const ScanVal *partial,
                 const Epoch *ep, const int *n_ep, const DParams *P, int32_t *trigcol_pos, int32_t *trigcol_off,
                 int32_t *inscol_pos, u64 *stats, int stat_hi, const int *spec_abort, int chunk)
{
    __shared__ ScanVal ws[4];
    if (LIST_ABORTED(spec_abort)) return;                  /* (the lists would not hold the columns: the run is repeated with exact sizes) */
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int base = blockIdx.x * chunk;
    const int nep = *n_ep;
    const double od = P->over_depth;
    ScanVal carry = partial[blockIdx.x];
    /* four consecutive columns per thread (16-byte loads, one wave scan and two barriers per 1024 columns); the next
     * round's loads are issued before this round's barriers */
    uint4 r4n = {0, 0, 0, 0};
    int4 n4n = {0, 0, 0, 0}, t4n = {0, 0, 0, 0};
#define KSO_LOAD(i0_) do { \
        const int j0_ = (i0_); \
        r4n.x = r4n.y = r4n.z = r4n.w = 0u; n4n.x = n4n.y = n4n.z = n4n.w = 0; t4n = n4n; \
        if (j0_ + 3 < L) { r4n = *(const uint4 *)(colrec + j0_); n4n = *(const int4 *)(nplp + j0_); t4n = *(const int4 *)(trigcnt + j0_); } \
        else if (j0_ < L) { \
            r4n.x = colrec[j0_]; n4n.x = nplp[j0_]; t4n.x = trigcnt[j0_]; \
            if (j0_ + 1 < L) { r4n.y = colrec[j0_ + 1]; n4n.y = nplp[j0_ + 1]; t4n.y = trigcnt[j0_ + 1]; } \
            if (j0_ + 2 < L) { r4n.z = colrec[j0_ + 2]; n4n.z = nplp[j0_ + 2]; t4n.z = trigcnt[j0_ + 2]; } \
        } } while (0)
    KSO_LOAD(base + 4 * (int)threadIdx.x);
    for (int k = 0; k < chunk / 1024; k++) {
        const int i0 = base + k * 1024 + 4 * (int)threadIdx.x;
        const uint32_t rec[4] = {r4n.x, r4n.y, r4n.z, r4n.w};
        const int n[4] = {n4n.x, n4n.y, n4n.z, n4n.w}, tcv[4] = {t4n.x, t4n.y, t4n.z, t4n.w};
        if (k + 1 < chunk / 1024) KSO_LOAD(i0 + 1024);
        ScanVal v[4], tot = {0, 0, 0, 0, 0, 0, 0};
        #pragma unroll
        for (int u = 0; u < 4; u++) { scanval_of(rec[u], n[u], tcv[u], v[u]); if (i0 + u >= L) { const ScanVal z = {0, 0, 0, 0, 0, 0, 0}; v[u] = z; } scanval_add(tot, v[u]); }
        const ScanVal inc = scanval_wave_incl(tot, lane);
        __syncthreads();                               /* ws reuse */
        if (lane == 63) ws[w] = inc;
        __syncthreads();
        ScanVal run = carry;
        for (int j = 0; j < w; j++) scanval_add(run, ws[j]);
        /* exclusive prefix of the thread's first column: the waves before, the lanes before */
        run.depth += inc.depth - tot.depth; run.cols += inc.cols - tot.cols; run.trigcols += inc.trigcols - tot.trigcols;
        run.items += inc.items - tot.items; run.ins += inc.ins - tot.ins;
        #pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = i0 + u;
            if (i < L && (rec[u] & CR_COUNTED) && !(rec[u] & CR_TOODEEP)) {
                overdepth_test(i, rec[u], run.cols, n[u], run.depth, colrec, ep, nep, od, stats, stat_hi);
                /* the lists are indexed by the exclusive prefixes k_scan_apply would have stored */
                if (rec[u] & CR_TRIG) { const int t = (int)run.trigcols; trigcol_pos[t] = i; trigcol_off[t] = (int)run.items; }
                if (rec[u] & CR_HASINS) inscol_pos[(int)run.ins] = i;
            }
            scanval_add(run, v[u]);
        }
        for (int j = 0; j < (int)(blockDim.x >> 6); j++) scanval_add(carry, ws[j]);
    }
#undef KSO_LOAD
}

/* ------------------------------------------------------------------------ */
/* Indel-length histogram test (snp:1708-1713, 1777-1819), columns with an
 * insertion only.  One thread per such column. */
__global__ void __launch_bounds__(64)
k_inshist(DBatch B, const DParams *P, int n_ins_, const int *n_dev, const int *spec_abort, const int32_t *inscol_pos,
          uint32_t *colrec, const int32_t *nplp, u64 *stats, int gshift, const ScanVal *totals)     /* (statistics: columns < B.stat_hi) */
{
    if (LIST_ABORTED(spec_abort)) return;
    const int n_ins = LIST_N(n_ins_, n_dev);
    if (totals) {                                          /* fast path: the host has not seen the scan's totals */
        const int64_t mean_depth = totals->cols > 0 ? totals->depth / totals->cols : 0;
        gshift = mean_depth < 40 ? 4 : 6;
    }
    /* a group of 1 << gshift lanes per column, lanes over its candidate reads: the whole 64-lane workgroup on a deep
     * pileup; 16 lanes -- four columns per wave -- when a column has a few dozen candidates (the kernel waits on chains
     * of dependent loads: a quarter of the waves, the same chains) */
    __shared__ int hist_all[4 * 100];
    __shared__ int s_sz_all[4];
    const int glanes = 1 << gshift, grp = (int)threadIdx.x >> gshift, gl = (int)threadIdx.x & (glanes - 1);
    int *const hist = hist_all + grp * 100;
    for (int tb = (int)blockIdx.x; tb * (64 >> gshift) < n_ins; tb += (int)gridDim.x) {     /* (uniform over the workgroup) */
    const int t = tb * (64 >> gshift) + grp;
    __syncthreads();                                       /* the previous round's histograms are done with */
    for (int i = gl; i < 100; i += glanes) hist[i] = 0;
    if (gl == 0) s_sz_all[grp] = 0;
    __syncthreads();
    const bool active = t < n_ins;
    const int c = active ? inscol_pos[t] : 0, apos = B.p0 + c, tile = c >> 6;
    if (active) {
        const int lo = B.tile_lo[tile], hi = B.tile_hi[tile];
        for (int r = lo + gl; r < hi; r += glanes) {
            if (!(apos >= B.pos[r] && apos < B.end[r])) continue;
            Entry e;
            resolve_entry(B, r, apos, e);
            if (e.is_refskip || e.is_head || e.is_tail) continue;
            if (e.indel > 0) atomicMax(&s_sz_all[grp], e.indel);
            if (e.indel >= 0) atomicAdd(&hist[e.indel < 99 ? e.indel : 99], 1);
        }
    }
    __syncthreads();
    if (!active || gl != 0) continue;
    int indel_sz = s_sz_all[grp];

    if (indel_sz > 100) indel_sz = 100;
    if (!indel_sz) continue;
    int qd1 = 0, qd2 = 0, ov = 0;
    for (int i = 0; i <= indel_sz && i < 100; i++) {
        if (!hist[i]) continue;
        ov += hist[i];
        if (qd1 < hist[i]) { qd2 = qd1; qd1 = hist[i]; }
        else if (qd2 < hist[i]) qd2 = hist[i];
    }
    uint32_t why = 0;
    if ((double)(ov - qd1 - qd2) > P->ins_len_perc * (ov + .1)) { why |= CR_KEEP | CR_R_INSLEN; stat_inc(stats, ST_INS_LEN_PERC, c < B.stat_hi); }
    if ((double)ov < P->indel_ov_perc * nplp[c]) { why |= CR_KEEP | CR_R_INDELOV; stat_inc(stats, ST_INDEL_OV_PERC, c < B.stat_hi); }
    if (why) colrec[c] |= why;
    }
}

/* ------------------------------------------------------------------------ */
/* Ordered trigger list of each trigger column (snp:1695-1730).  One wave per trigger column: the lanes
 * test 64 candidate reads at a time and the hits are written in file order (ballot + prefix popcount). */
__global__ void __launch_bounds__(256)
k_emit(DBatch B, const DParams *P, int n_trig_, const int *n_dev, const int *spec_abort, const int32_t *trigcol_pos,
       const int32_t *trigcol_off, const uint32_t *colrec, TrigItem *items)
{
    if (LIST_ABORTED(spec_abort)) return;
    const int n_trig = LIST_N(n_trig_, n_dev), lane = threadIdx.x & 63;
    for (int t = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)); t < n_trig; t += (int)(gridDim.x * (blockDim.x >> 6))) {   /* (the whole wave) */
    const int c = trigcol_pos[t], apos = B.p0 + c, tile = c >> 6;
    const int lo = B.tile_lo[tile], hi = B.tile_hi[tile];
    const uint32_t rec = colrec[c];
    const bool snp_kind = P->str_snp && CR_PRESERVE(rec);
    int o = trigcol_off[t];
    for (int base = lo; base < hi; base += 64) {
        const int r = base + lane;
        bool emit = false;
        TrigItem it;
        it.read = r; it.qpos1_kind = 0; it.lo = INT_MAX; it.hi = 0;
        if (r < hi && apos >= B.pos[r] && apos < B.end[r]) {
            Entry e;
            resolve_entry(B, r, apos, e);
            const int is_indel = (e.indel || e.is_del) ? 1 : 0;
            if (!e.is_refskip && (is_indel || snp_kind)) {
                emit = true;
                it.qpos1_kind = (e.qpos + 1) | (is_indel << 30) | ((rec & CR_SCAN) ? 0 : (1 << 29));
            }
        }
        const unsigned long long m = __ballot(emit);
        if (emit) items[o + __popcll(m & ((1ull << lane) - 1ull))] = it;
        o += __popcll(m);
    }
    }
}

/* the same with one lane per trigger column: long-read batches, where a column has only a few candidates */
__global__ void k_emit_lanes(DBatch B, const DParams *P, int n_trig_, const int *n_dev, const int *spec_abort, const int32_t *trigcol_pos,
                             const int32_t *trigcol_off, const uint32_t *colrec, TrigItem *items)
{
    if (LIST_ABORTED(spec_abort)) return;
    const int n_trig = LIST_N(n_trig_, n_dev);
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n_trig; t += (int)(gridDim.x * blockDim.x)) {
    const int c = trigcol_pos[t], apos = B.p0 + c, tile = c >> 6;
    const int lo = B.tile_lo[tile], hi = B.tile_hi[tile];
    const uint32_t rec = colrec[c];
    const bool snp_kind = P->str_snp && CR_PRESERVE(rec);
    int o = trigcol_off[t];
    for (int r = lo; r < hi; r++) {
        if (!(apos >= B.pos[r] && apos < B.end[r])) continue;
        Entry e;
        resolve_entry(B, r, apos, e);
        if (e.is_refskip) continue;
        const int is_indel = (e.indel || e.is_del) ? 1 : 0;
        if (!(is_indel || snp_kind)) continue;
        TrigItem it;
        it.read = r; it.qpos1_kind = (e.qpos + 1) | (is_indel << 30) | ((rec & CR_SCAN) ? 0 : (1 << 29));
        it.lo = INT_MAX; it.hi = 0;
        items[o++] = it;
    }
    }
}

/* ------------------------------------------------------------------------ */
/* find_STR / add_rep (str_finder.c:34-189) on the 2-bit codes of a read
 * window, and mask_LC_regions (snp:1230-1290).  One thread per trigger.      */
struct StrWin {
    const unsigned *w;      /* this lane's window in LDS as two bit planes of the 2-bit codes, 32 positions
                             * per word: low bits at w[k*64], high bits at w[(STR_PW + k)*64] */
};
#define STR_PW    16        /* words per plane: 512 positions >= 2*MASK_WIN+1 */
#define STR_WORDS (2 * STR_PW)
__device__ __forceinline__ int str_code(const StrWin &sw, int i)
{
    const unsigned lo = sw.w[(i >> 5) * 64] >> (i & 31), hi = sw.w[(STR_PW + (i >> 5)) * 64] >> (i & 31);
    return (int)((lo & 1u) | ((hi & 1u) << 1));
}
/* str_finder.c:15-32 via seq_nt16_str: C=1 G=2 T=3, all else (A, N, '=', IUPAC) 0 */
__device__ __forceinline__ unsigned nt16_to_str2(unsigned c16)
{
    return (c16 == 2) ? 1u : (c16 == 4) ? 2u : (c16 == 8) ? 3u : 0u;
}

/* add_rep (str_finder.c:34-127) without pads; the utlist list is an ordered array (entry = start | end << 16) whose
 * tail element is mirrored in registers (ts, te): most calls end at the "already covered by the tail" test or append
 * without touching older elements.  Three parts: the covered test, the extension of the repeat to the right
 * (str_finder.c:62-75; by bytes here, by whole words of the equality masks in dev_find_STR), the list update. */
#define STR_ENT(st_, en_) ((unsigned)(st_) | ((unsigned)(en_) << 16))
#define STR_ENT_ST(e_)    ((int)((e_) & 0xffffu))
#define STR_ENT_EN(e_)    ((int)((e_) >> 16))
__device__ __forceinline__ bool str_rep_covered(int n, int ts, int te, int pos, int rlen)
{
    return n && ts <= pos - rlen * 2 + 1 && te >= pos;                       /* :41-45 */
}
__device__ __forceinline__ int str_rep_insert(unsigned *lst, int n, int &ts, int &te, int el_start, int el_end, int *overflow)
{
    if (n && te >= el_start) {                          /* :106-122 (the walk stops at once otherwise) */
        int cur = n - 1;
        int cs = ts, ce = te;                           /* the tail's entry is in registers */
        for (;;) {
            const bool was_head = (cur == 0);
            int prev = was_head ? n - 1 : cur - 1;      /* head->prev is the tail */
            bool deleted = false;
            if (ce < el_start) break;
            if (cs >= el_start) {
                for (int k = cur; k < n - 1; k++) lst[k] = lst[k + 1];
                n--; deleted = true;
            }
            if (was_head && !deleted) break;
            if (!n) break;
            if (was_head) prev = n - 1;
            cur = prev;
            const unsigned e = lst[cur];
            cs = STR_ENT_ST(e); ce = STR_ENT_EN(e);
        }
    }
    if (n < STR_CAP) { lst[n] = STR_ENT(el_start, el_end); n++; ts = el_start; te = el_end; }
    else *overflow = 1;
    return n;
}
__device__ __forceinline__ int dev_add_rep(const StrWin &w, int len, int pos, int rlen,
                                           unsigned *lst, int n, int &ts, int &te, int *overflow)
{
    if (str_rep_covered(n, ts, te, pos, rlen)) return n;
    int cp1 = pos + 1 - rlen, cp2 = pos + 1;
    while (cp2 < len && str_code(w, cp1) == str_code(w, cp2)) { cp1++; cp2++; }  /* :62-75 */
    return str_rep_insert(lst, n, ts, te, pos + 1 - 2 * rlen, cp2 - 1, overflow);
}

/* find_STR (str_finder.c:135-189).  From position 15 on, the reference asks at every position i for
 * the LARGEST period rl in 1..8 whose last rl codes equal the rl codes before them, i.e. whether
 * D_rl[i-k] = (x[i-k] == x[i-k-rl]) holds for k = 0..rl-1.  That is computed for 32 positions at a
 * time on the two bit planes (funnel shifts across the word boundary, run-ANDs by doubling); the
 * positions with a hit are then visited in order, and only those reach add_rep().  The first 15
 * positions (every period that fits, ascending) keep the reference's rolling word. */
/* pos_limit: positions beyond it are not visited.  mask_LC_regions only keeps the repeats that overlap
 * [rpos - add, rpos + add]; a repeat found at position p starts at p + 1 - 2 * period >= p - 15 and can evict only
 * repeats that start at or after its own start, so what is found more than 15 positions to the right of that zone
 * neither is kept nor evicts anything that is (the extension of a repeat to the right still reads the whole
 * window). */
__device__ __forceinline__ int dev_find_STR(const StrWin &w, int len, unsigned *lst, int *overflow, int pos_limit)
{
    int n = 0, ts = 0, te = -1;
    uint32_t word = 0;
    for (int i = 0; i < len && i < 15 && i <= pos_limit; i++) {
        word = (word << 2) | (uint32_t)str_code(w, i);
        for (int rlen = 1; rlen <= 7; rlen++) {             /* every period that fits */
            if (i < 2 * rlen - 1) break;
            const uint32_t mask = (1u << (2 * rlen)) - 1;
            if ((word & mask) == ((word >> (2 * rlen)) & mask))
                n = dev_add_rep(w, len, i, rlen, lst, n, ts, te, overflow);
        }
    }
    /* D[p-1]: equality with the code p positions back, for the previous word (needed for runs that
     * cross the word boundary) */
    uint32_t Dp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t x0p = 0, x1p = 0;                              /* planes of the previous word */
    const int nwords = (len + 31) >> 5;
    int nw = nwords;
    if (pos_limit < 0) nw = 0;
    else if ((pos_limit >> 5) + 1 < nw) nw = (pos_limit >> 5) + 1;
    for (int wi = 0; wi < nw; wi++) {
        const uint32_t x0 = w.w[wi * 64], x1 = w.w[(STR_PW + wi) * 64];
        uint32_t E[8];
        #pragma unroll
        for (int p = 1; p <= 8; p++) {
            /* codes p positions back, aligned with this word's positions */
            const uint32_t s0 = (x0 << p) | (x0p >> (32 - p)), s1 = (x1 << p) | (x1p >> (32 - p));
            const uint32_t D = ~((x0 ^ s0) | (x1 ^ s1)), Dq = Dp[p - 1];
            /* run of p consecutive ones ending at each position (previous word supplies the carry-in) */
            uint32_t r = D, rq = Dq;                        /* runs of length 1 */
            if (p >= 2) { const uint32_t t = r & ((r << 1) | (rq >> 31)), tq = rq & (rq << 1); r = t; rq = tq; }          /* 2 */
            if (p >= 4) { const uint32_t t = r & ((r << 2) | (rq >> 30)), tq = rq & (rq << 2); r = t; rq = tq; }          /* 4 */
            if (p == 8) { r = r & ((r << 4) | (rq >> 28)); }                                                             /* 8 */
            if (p == 3) { r = r & ((D << 2) | (Dq >> 30)); }                                                             /* 2 + 1 */
            if (p == 5) { r = r & ((D << 4) | (Dq >> 28)); }                                                             /* 4 + 1 */
            if (p == 6) { const uint32_t d2 = D & ((D << 1) | (Dq >> 31)), d2q = Dq & (Dq << 1);
                          r = r & ((d2 << 4) | (d2q >> 28)); }                                                           /* 4 + 2 */
            if (p == 7) { const uint32_t d2 = D & ((D << 1) | (Dq >> 31)), d2q = Dq & (Dq << 1);
                          r = r & ((d2 << 4) | (d2q >> 28)) & ((D << 6) | (Dq >> 26)); }                                 /* 4 + 2 + 1 */
            E[p - 1] = r;
            Dp[p - 1] = D;
        }
        uint32_t any = E[0] | E[1] | E[2] | E[3] | E[4] | E[5] | E[6] | E[7];
        if (wi == 0) any &= ~0x7fffu;                       /* positions 0..14 were handled above */
        if (wi == nwords - 1 && (len & 31)) any &= (1u << (len & 31)) - 1u;
        if (wi == (pos_limit >> 5) && (pos_limit & 31) != 31) any &= (2u << (pos_limit & 31)) - 1u;
        while (any) {
            const int b = __builtin_ctz(any);
            any &= any - 1;
            const uint32_t bit = 1u << b;
            const int rl = (E[7] & bit) ? 8 : (E[6] & bit) ? 7 : (E[5] & bit) ? 6 : (E[4] & bit) ? 5
                         : (E[3] & bit) ? 4 : (E[2] & bit) ? 3 : (E[1] & bit) ? 2 : 1;
            const int pos = wi * 32 + b;
            if (str_rep_covered(n, ts, te, pos, rl)) continue;
            /* the repeat runs on to the right while x[i] == x[i - rl] (str_finder.c:62-75): the ones of this period's
             * equality mask behind bit b, word by word (a byte loop through LDS before: two dependent reads per
             * position).  Codes behind the window's end read as 0: the count is cut at len. */
            int ext;
            {
                const uint32_t s0 = (x0 << rl) | (x0p >> (32 - rl)), s1 = (x1 << rl) | (x1p >> (32 - rl));
                const uint32_t D = ~((x0 ^ s0) | (x1 ^ s1));
                const int room = 31 - b;
                const uint32_t m = room ? (D >> (b + 1)) : 0u;
                const uint32_t z = ~m;
                ext = z ? __builtin_ctz(z) : 32;
                if (ext >= room) {
                    ext = room;
                    uint32_t y0 = x0, y1 = x1;
                    for (int wj = wi + 1; wj < nwords; wj++) {
                        const uint32_t n0 = w.w[wj * 64], n1 = w.w[(STR_PW + wj) * 64];
                        const uint32_t t0 = (n0 << rl) | (y0 >> (32 - rl)), t1 = (n1 << rl) | (y1 >> (32 - rl));
                        const uint32_t zn = (n0 ^ t0) | (n1 ^ t1);                      /* ~D of the next word */
                        const int c = zn ? __builtin_ctz(zn) : 32;
                        ext += c;
                        if (c < 32) break;
                        y0 = n0; y1 = n1;
                    }
                }
                if (ext > len - 1 - pos) ext = len - 1 - pos;
            }
            n = str_rep_insert(lst, n, ts, te, pos + 1 - 2 * rl, pos + ext, overflow);
        }
        x0p = x0; x1p = x1;
    }
    return n;
}

/* bam_qpos2rpos (snp:1205-1219) on the raw CIGAR */
__device__ __forceinline__ int dev_qpos2rpos(const uint32_t *cig, int nc, int rpos0, int qpos)
{
    int rpos = rpos0, aqpos = 0;
    for (int k = 0; k < nc && aqpos < qpos; k++) {
        const int op = cig[k] & 15, l = (int)(cig[k] >> 4);
        const int ty = (0x3C1A7 >> (op << 1)) & 3;
        if (ty & 2) rpos += (l <= qpos - aqpos) ? l : (qpos - aqpos);
        if (ty & 1) aqpos += l;
    }
    return rpos;
}
/* ref2query_pos (snp:1156-1179) */
__device__ __forceinline__ int dev_ref2query_pos(const uint32_t *cig, int nc, int rpos0, int pos)
{
    int p = rpos0, q = 0;
    for (int i = 0; i < nc; i++) {
        const int op = cig[i] & 15, l = (int)(cig[i] >> 4);
        const int ty = (0x3C1A7 >> (op << 1)) & 3;
        if (p + ((ty & 2) ? l : 0) < pos) {
            if (ty & 1) q += l;
            if (ty & 2) p += l;
            continue;
        }
        if (ty & 1) q += (pos - p);
        return q >= 0 ? q : 0;
    }
    return q;
}

/* ref2query_pos (snp:1156-1179) from the op table instead of the CIGAR walk (k_restore on long reads: the walk is a
 * chain of dependent loads, ~25 ops per HiFi read).  The reference stops at the first op -- in CIGAR order, clips
 * and insertions included -- with p + (ref-consuming ? len : 0) >= pos.  For pos <= rpos0 that is the first op and
 * the result is 0.  Otherwise a non-reference op never stops the walk (the reference op before it did not, so
 * p < pos there), and the stopping op is the first reference-consuming op whose end reaches pos: Seg.qstart is the
 * query length consumed before it (M: + the offset inside it; D/N: as it is); past the last one the walk returns the
 * whole query length. */
__device__ __forceinline__ int dev_ref2query_seg(const Seg *__restrict__ segs, int ns, int rpos0, int pos, int lq)
{
    if (pos <= rpos0) return 0;
    const int off = pos - rpos0;
    int lo = 0, hi = ns;                                   /* first op with roff + len >= off */
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (segs[mid].roff + seg_len(segs[mid]) < off) lo = mid + 1; else hi = mid; }
    if (lo >= ns) return lq;
    const Seg sg = segs[lo];
    return seg_type(sg) == SEG_M ? sg.qstart + (off - sg.roff) : sg.qstart;
}

__device__ inline void dev_mask_lc(const DBatch &B, const DParams *P, int r, int rpos, int is_indel,
                                   int *min_pos, int *max_pos, int *overflow, unsigned *lds_lane)
{
    unsigned lst[STR_CAP];                                 /* start | end << 16 */
    const int64_t so = B.seq_off[r];
    const int lq = (int)(B.seq_off[r + 1] - so);
    const uint8_t *bases = B.bases + so;
    const int start = rpos - MASK_WIN_DEV > 0 ? rpos - MASK_WIN_DEV : 0;
    const int end = rpos + MASK_WIN_DEV < lq ? rpos + MASK_WIN_DEV : lq;
    const int len = end - start + 1;                       /* inclusive (snp:1235,1241) */
    /* index l_qseq is read too: pad nibble, or high nibble of the capped qual[0] */
    const unsigned tail_code = ((lq & 1) || !lq) ? 0u : (unsigned)(cap_q(P, B.quals[so]) >> 4);
    /* stage the window in LDS as two bit planes of the 2-bit codes (C=1 G=2 T=3, else 0), 32 positions
     * per word */
    for (int k = 0; k * 32 < len; k++) {
        unsigned p0 = 0, p1 = 0;
        if (k * 32 + 32 <= len && start + k * 32 + 32 <= lq) {
            /* all 32 codes are bases of the read: eight dwords (+1 for the byte offset), four bases at
             * a time with packed byte compares against C, G, T */
            const uint8_t *bp = bases + start + k * 32;
            const unsigned sh = (unsigned)((uintptr_t)bp & 3u), s8 = sh * 8u;
            const uint32_t *bw = (const uint32_t *)(bp - sh);
            uint32_t prev = bw[0];
            #pragma unroll
            for (int d = 0; d < 8; d++) {
                const uint32_t next = (d < 7 || sh) ? bw[d + 1] : 0u;
                const uint32_t c = (sh ? (prev >> s8) | (next << (32u - s8)) : prev) & 0x0f0f0f0fu;
                prev = next;
                #define STR_NZ(x) ((((x) + 0x7f7f7f7fu) & 0x80808080u) >> 7)          /* 1 in bytes that are != 0 */
                const uint32_t nC = STR_NZ(c ^ 0x02020202u), nG = STR_NZ(c ^ 0x04040404u), nT = STR_NZ(c ^ 0x08080808u);
                #undef STR_NZ
                const uint32_t b0 = ~(nC & nT) & 0x01010101u;                        /* C or T: low bit of the code  */
                const uint32_t b1 = ~(nG & nT) & 0x01010101u;                        /* G or T: high bit of the code */
                p0 |= ((b0 * 0x01020408u) >> 24) << (4 * d);
                p1 |= ((b1 * 0x01020408u) >> 24) << (4 * d);
            }
        } else {
            #pragma unroll
            for (int j = 0; j < 32; j++) {
                const int i = k * 32 + j, qi = start + i;
                unsigned c16 = 0;
                if (i < len) c16 = (qi < lq) ? (unsigned)(bases[qi] & 15) : tail_code;
                const unsigned c = nt16_to_str2(c16);
                p0 |= (c & 1u) << j; p1 |= (c >> 1) << j;
            }
        }
        lds_lane[k * 64] = p0; lds_lane[(STR_PW + k) * 64] = p1;
    }
    StrWin w; w.w = lds_lane;
    const int add = is_indel ? P->iSTR_add : P->sSTR_add;
    const int64_t lim = (int64_t)rpos + (add > 0 ? add : 0) - start + 15;
#ifdef KSTR_EXP_NOFIND       /* diagnostic build (wrong results): the window staging alone */
    const int n = (lds_lane[0] == 0x12345u) ? 1 : 0; lst[0] = 0;
#else
    const int n = dev_find_STR(w, len, lst, overflow, lim > len ? len : (int)lim);
#endif
    const uint32_t *cig = B.cigar + B.cig_off[r];
    const int nc = (int)(B.cig_off[r + 1] - B.cig_off[r]);
    for (int i0 = 0; i0 < n; i0 += 4) {                    /* (four entries' loads in flight) */
        unsigned e4[4];
        #pragma unroll
        for (int u = 0; u < 4; u++) e4[u] = i0 + u < n ? lst[i0 + u] : 0xffffu;       /* (start 65535: never kept) */
        #pragma unroll
        for (int u = 0; u < 4; u++) {
            int s = STR_ENT_ST(e4[u]) + start, e = STR_ENT_EN(e4[u]) + start;
            if (i0 + u >= n || !(rpos + add >= s && rpos - add <= e)) continue;
            s = dev_qpos2rpos(cig, nc, B.pos[r], s);
            e = dev_qpos2rpos(cig, nc, B.pos[r], e);
            if (*min_pos > s) *min_pos = s;
            if (*max_pos < e) *max_pos = e;
        }
    }
}

__global__ void __launch_bounds__(64)
k_str(DBatch B, const DParams *P, int n_items_, const int *n_dev, const int *spec_abort, TrigItem *items, int *overflow)
{
    __shared__ unsigned win[STR_WORDS * 64];           /* lane t owns win[k*64 + t] */
    if (LIST_ABORTED(spec_abort)) return;
    const int n_items = LIST_N(n_items_, n_dev);
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n_items; t += (int)(gridDim.x * blockDim.x)) {
        const TrigItem it = items[t];
        if (it.qpos1_kind & (1 << 29)) continue;       /* indel_cnt < n_plp*indel_fract: snp:1732 */
        int lo = INT_MAX, hi = 0;
        dev_mask_lc(B, P, it.read, it.qpos1_kind & 0x1fffffff, (it.qpos1_kind >> 30) & 1, &lo, &hi, overflow,
                    &win[threadIdx.x & 63]);
        items[t].lo = lo; items[t].hi = hi;
    }
}

/* ------------------------------------------------------------------------ */
/* Window state across columns (snp:1508-1511, 1741-1755).
 * A chain is a run of trigger columns not separated by a reset
 * (pos > max_pos2).  ubound[k] is a sound upper bound of max_pos2 after
 * column k under ANY incoming state, so pos[k+1] > ubound[k] proves a reset:
 * those are independent segments, folded sequentially by one thread each. */
__device__ __forceinline__ int fold_bound_of(const DBatch &B, int k, const int32_t *trigcol_pos, const int32_t *trigcol_off,
                                             const int32_t *trigcnt, const TrigItem *items, int in_max_pos)
{
    const int c = trigcol_pos[k], pos = B.p0 + c;
    const int o = trigcol_off[k], n = trigcnt[c];
    int h = pos;
    for (int i = 0; i < n; i++) if (items[o + i].hi > h) h = items[o + i].hi;
    if (k == 0 && in_max_pos > h) h = in_max_pos;            /* the state the window starts from */
    return h;
}
__global__ void k_fold_bounds(DBatch B, const DParams *P, int n_trig_, const int *n_dev, const int *spec_abort, const int32_t *trigcol_pos,
                              const int32_t *trigcol_off, const int32_t *trigcnt,
                              const TrigItem *items, int32_t *colhi, int in_max_pos)
{
    if (LIST_ABORTED(spec_abort)) return;
    const int n_trig = LIST_N(n_trig_, n_dev);
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n_trig; k += (int)(gridDim.x * blockDim.x))
        colhi[k] = fold_bound_of(B, k, trigcol_pos, trigcol_off, trigcnt, items, in_max_pos);
}

/* generic chunked inclusive prefix-max over int32: one 256-thread workgroup per chunk of PMAX_CHUNK
 * values (8 consecutive values per thread), the chunk maxima scanned by one wave */
#define PMAX_CHUNK 2048
__device__ __forceinline__ int pmax_wave_max(int m)
{
    #pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(m, d); m = o > m ? o : m; }
    return m;
}
__global__ void __launch_bounds__(256)
k_pmax_reduce(int n_, const int *n_dev, const int32_t *__restrict__ v, int32_t *__restrict__ partial)
{
    __shared__ int wm[4];
    const int n = LIST_N(n_, n_dev);
    if ((int)blockIdx.x * PMAX_CHUNK >= n) return;
    const int b = blockIdx.x * PMAX_CHUNK + threadIdx.x * 8;
    int m = INT_MIN;
    #pragma unroll
    for (int i = 0; i < 8; i++) if (b + i < n) { const int x = v[b + i]; m = x > m ? x : m; }
    m = pmax_wave_max(m);
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        int r = wm[0];
        #pragma unroll
        for (int w = 1; w < 4; w++) r = wm[w] > r ? wm[w] : r;
        partial[blockIdx.x] = r;
    }
}
/* exclusive prefix-max of the chunk maxima, in place: one wave, 64 chunks per step */
__global__ void __launch_bounds__(64)
k_pmax_partials(int32_t *partial, int nchunks_, int n_, const int *n_dev)
{
    const int lane = threadIdx.x;
    const int nchunks = n_dev ? (LIST_N(n_, n_dev) + PMAX_CHUNK - 1) / PMAX_CHUNK : nchunks_;
    int run = INT_MIN;
    for (int base = 0; base < nchunks; base += 64) {
        const int x = base + lane < nchunks ? partial[base + lane] : INT_MIN;
        int incl = x;
        #pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d && o > incl) incl = o; }
        int excl = __shfl_up(incl, 1);
        if (lane == 0) excl = INT_MIN;
        if (base + lane < nchunks) partial[base + lane] = excl > run ? excl : run;
        const int tot = __shfl(incl, 63);
        run = tot > run ? tot : run;
    }
}
__global__ void __launch_bounds__(256)
k_pmax_apply(int n_, const int *n_dev, int32_t *__restrict__ v, const int32_t *__restrict__ partial)
{
    __shared__ int wm[4];
    const int n = LIST_N(n_, n_dev);
    if ((int)blockIdx.x * PMAX_CHUNK >= n) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int b = blockIdx.x * PMAX_CHUNK + threadIdx.x * 8;
    int x[8], m = INT_MIN;
    #pragma unroll
    for (int i = 0; i < 8; i++) { x[i] = b + i < n ? v[b + i] : INT_MIN; m = x[i] > m ? x[i] : m; x[i] = m; }   /* thread-local inclusive */
    int incl = m;
    #pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d && o > incl) incl = o; }
    if (lane == 63) wm[w] = incl;
    int excl = __shfl_up(incl, 1);
    if (lane == 0) excl = INT_MIN;
    __syncthreads();
    int run = partial[blockIdx.x];
    for (int k = 0; k < w; k++) run = wm[k] > run ? wm[k] : run;
    run = excl > run ? excl : run;
    #pragma unroll
    for (int i = 0; i < 8; i++) if (b + i < n) v[b + i] = x[i] > run ? x[i] : run;
}

/* colhi (already prefix-maxed = bound on max_pos) -> per-column bound on max_pos2 */
__device__ __forceinline__ int fold_ub_of(const DBatch &B, const DParams *P, int k, const int32_t *trigcol_pos, int colhi_k, int in_max_pos2)
{
    const double pos = B.p0 + trigcol_pos[k], h = colhi_k;
    double a = pos + (h - pos) * P->iSTR_mul + P->iSTR_add;
    double b = pos + (h - pos) * P->sSTR_mul + P->sSTR_add;
    double m = a > b ? a : b;
    if (m < pos) m = pos;
    m += 2;                                         /* slack over the truncations */
    if (k == 0 && (double)in_max_pos2 > m) m = in_max_pos2;
    return m > 2147483000.0 ? 2147483000 : (int)m;
}
__global__ void k_fold_ub(DBatch B, const DParams *P, int n_trig_, const int *n_dev, const int *spec_abort, const int32_t *trigcol_pos,
                          const int32_t *colhi, int32_t *ub, int in_max_pos2)
{
    if (LIST_ABORTED(spec_abort)) return;
    const int n_trig = LIST_N(n_trig_, n_dev);
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n_trig; k += (int)(gridDim.x * blockDim.x))
        ub[k] = fold_ub_of(B, P, k, trigcol_pos, colhi[k], in_max_pos2);
}

/* one segment of trigger columns from its start k0 (no-op when k0 is not a segment start) */
__device__ __forceinline__ void fold_chain(const DBatch &B, const DParams *P, int n_trig, int k0, const int32_t *trigcol_pos,
                       const int32_t *trigcol_off, const int32_t *trigcnt, const TrigItem *items,
                       const int32_t *ubmax /* inclusive prefix max of ub */,
                       int32_t *pos_mn2, int32_t *trig_mx2,
                       int in_min_pos, int in_max_pos, int in_min_pos2, int in_max_pos2, int cut_pos, int32_t *carry_out)
{
    /* (in_*: the state the first column starts from -- INT_MAX 0 INT_MAX 0 for a whole contig; carry_out: the
     * state after the last trigger column before cut_pos, for the next window of the contig) */
    if (k0 > 0 && !(B.p0 + trigcol_pos[k0] > ubmax[k0 - 1])) return;   /* not a segment start */
    int min_pos = INT_MAX, max_pos = 0, min_pos2 = INT_MAX, max_pos2 = 0;
    if (k0 == 0) { min_pos = in_min_pos; max_pos = in_max_pos; min_pos2 = in_min_pos2; max_pos2 = in_max_pos2; }
    const double imul = P->iSTR_mul, smul = P->sSTR_mul;
    const int iadd = P->iSTR_add, sadd = P->sSTR_add;
    for (int k = k0; k < n_trig; k++) {
        const int c = trigcol_pos[k], pos = B.p0 + c;
        if (k > k0 && pos > ubmax[k - 1]) break;                         /* next segment */
        if (pos > max_pos2) {                                            /* snp:1508-1511 */
            min_pos2 = min_pos = INT_MAX;
            max_pos2 = max_pos = 0;
        }
        const int o = trigcol_off[k], n = trigcnt[c];
        for (int i = 0; i < n; i++) {
            const TrigItem it = items[o + i];
            if (min_pos > it.lo) min_pos = it.lo;                        /* mask_LC_regions */
            if (max_pos < it.hi) max_pos = it.hi;
            if (min_pos > pos) min_pos = pos;                            /* snp:1741-1742 */
            if (max_pos < pos) max_pos = pos;
            const bool ind = (it.qpos1_kind >> 30) & 1;
            const double mul = ind ? imul : smul;
            const int add = ind ? iadd : sadd;
            const double e1 = pos - (pos - min_pos) * mul - add;         /* snp:1746-1754 */
            const double e2 = pos + (max_pos - pos) * mul + add;
            min_pos2 = ((double)min_pos2 < e1) ? min_pos2 : (int)e1;
            max_pos2 = ((double)max_pos2 > e2) ? max_pos2 : (int)e2;
        }
        pos_mn2[c] = min_pos2;
        trig_mx2[k] = max_pos2;
        if (pos < cut_pos && (k + 1 == n_trig || B.p0 + trigcol_pos[k + 1] >= cut_pos)) {
            carry_out[0] = min_pos; carry_out[1] = max_pos; carry_out[2] = min_pos2; carry_out[3] = max_pos2;
        }
    }
}
__global__ void k_fold(DBatch B, const DParams *P, int n_trig_, const int *n_dev, const int *spec_abort, const int32_t *trigcol_pos,
                       const int32_t *trigcol_off, const int32_t *trigcnt, const TrigItem *items,
                       const int32_t *ubmax /* inclusive prefix max of ub */,
                       uint32_t *colrec, int32_t *pos_mn2, int32_t *trig_mx2,
                       int in_min_pos, int in_max_pos, int in_min_pos2, int in_max_pos2, int cut_pos, int32_t *carry_out)
{
    if (LIST_ABORTED(spec_abort)) return;
    const int n_trig = LIST_N(n_trig_, n_dev);
    for (int k0 = blockIdx.x * blockDim.x + threadIdx.x; k0 < n_trig; k0 += (int)(gridDim.x * blockDim.x))
        fold_chain(B, P, n_trig, k0, trigcol_pos, trigcol_off, trigcnt, items, ubmax, pos_mn2, trig_mx2,
                   in_min_pos, in_max_pos, in_min_pos2, in_max_pos2, cut_pos, carry_out);
}

/* the columns from the window's first one up to the first trigger column are "in window" while pos <= the
 * max_pos2 the window starts from (snp:1880) */
__global__ void k_window_head(DBatch B, int n_trig, const int32_t *trigcol_pos, int in_max_pos2, uint32_t *colrec)
{
    const int next = n_trig > 0 ? trigcol_pos[0] : B.col_hi;
    int64_t last = (int64_t)in_max_pos2 - B.p0;
    if (last > next - 1) last = next - 1;
    for (int64_t i = B.col_lo + (int64_t)(blockIdx.x * blockDim.x + threadIdx.x); i <= last; i += (int64_t)gridDim.x * blockDim.x)
        colrec[i] |= CR_WIN;
}

/* columns after a trigger column stay "in window" while pos <= max_pos2 (snp:1880) */
__device__ __forceinline__ void window_of(const DBatch &B, int n_trig, int k, const int32_t *trigcol_pos, const int32_t *trig_mx2, uint32_t *colrec)
{
    const int c = trigcol_pos[k];
    int64_t last = (int64_t)trig_mx2[k] - B.p0;
    const int next = (k + 1 < n_trig) ? trigcol_pos[k + 1] : B.L;
    if (last > next - 1) last = next - 1;
    for (int64_t i = c + 1; i <= last; i++) colrec[i] |= CR_WIN;
}
__global__ void k_window(DBatch B, int n_trig_, const int *n_dev, const int *spec_abort, const int32_t *trigcol_pos, const int32_t *trig_mx2,
                         uint32_t *colrec)
{
    if (LIST_ABORTED(spec_abort)) return;
    const int n_trig = LIST_N(n_trig_, n_dev);
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n_trig; k += (int)(gridDim.x * blockDim.x))
        window_of(B, n_trig, k, trigcol_pos, trig_mx2, colrec);
}

/* Short trigger lists (a small contig): bounds, both prefix maxima, the fold and the window marks by ONE 1024-thread
 * workgroup in one launch -- ten dependent launches of a few microseconds each were a fifth of a 5 kb contig's run.
 * Whole contigs only (the windows' phases hand state on between the steps). */
__device__ __forceinline__ void block_prefix_max(int32_t *v, int n, int *wm /* [17] */)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (int)(blockDim.x >> 6);
    int carry = INT_MIN;
    for (int base = 0; base < n; base += (int)blockDim.x) {
        const int i = base + (int)threadIdx.x;
        int incl = i < n ? v[i] : INT_MIN;
        #pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d && o > incl) incl = o; }
        if (lane == 63) wm[w] = incl;
        __syncthreads();
        int run = carry;
        for (int j = 0; j < w; j++) run = wm[j] > run ? wm[j] : run;
        if (i < n) v[i] = incl > run ? incl : run;
        for (int j = 0; j < nw; j++) carry = wm[j] > carry ? wm[j] : carry;
        __syncthreads();
    }
}
__global__ void __launch_bounds__(1024)
k_fold_small(DBatch B, const DParams *P, int n_trig_, const int *n_dev, const int *spec_abort, const int32_t *trigcol_pos,
             const int32_t *trigcol_off, const int32_t *trigcnt, const TrigItem *items, int32_t *colhi, int32_t *ub,
             uint32_t *colrec, int32_t *pos_mn2, int32_t *trig_mx2, int32_t *carry_out)
{
    __shared__ int wm[17];
    if (LIST_ABORTED(spec_abort)) return;
    const int n_trig = LIST_N(n_trig_, n_dev);
    for (int k = threadIdx.x; k < n_trig; k += (int)blockDim.x) colhi[k] = fold_bound_of(B, k, trigcol_pos, trigcol_off, trigcnt, items, 0);
    __syncthreads();
    block_prefix_max(colhi, n_trig, wm);
    for (int k = threadIdx.x; k < n_trig; k += (int)blockDim.x) ub[k] = fold_ub_of(B, P, k, trigcol_pos, colhi[k], 0);
    __syncthreads();
    block_prefix_max(ub, n_trig, wm);
    for (int k0 = threadIdx.x; k0 < n_trig; k0 += (int)blockDim.x)
        fold_chain(B, P, n_trig, k0, trigcol_pos, trigcol_off, trigcnt, items, ub, pos_mn2, trig_mx2,
                   INT_MAX, 0, INT_MAX, 0, B.p0 + B.stat_hi, carry_out);
    __syncthreads();
    for (int k = threadIdx.x; k < n_trig; k += (int)blockDim.x) window_of(B, n_trig, k, trigcol_pos, trig_mx2, colrec);
}

/* ------------------------------------------------------------------------ */
/* Replay of the per-base loop (snp:1822-1920), the keep_qual retirement copy
 * (snp:1939) and the 0x80 strip of flush_bam_list (snp:1093-1096).
 *
 * The reference mutates each read's output copy column by column.  The final
 * value of quality byte x of a read only depends on (DESIGN.md "replay"):
 *   (a) the is_head low-mapq mark (snp:1852) and -S binning (snp:1894),
 *   (b) the columns whose pileup entry has qpos == x, in order: the D/N columns
 *       just before x (htslib reports the NEXT base for them) and x's own column,
 *   (c) whether any LATER column of the read is a trigger column whose back-fill
 *       range [ref2query_pos(min_pos2), qpos] reaches down to x (snp:1875):
 *       the byte is then the capped original,
 *   (d) the OR of keep_qual over the read's columns (snp:1847,1939): capped original.
 * (a)+(b) are a pass over the flat base array -- packed 16-byte chunks for short-read
 * batches (k_actions / k_replay_fast / k_replay_reads), 16 bases per lane with general
 * per-base logic otherwise (k_replay_flat), one lane per base with -S (k_replay_general);
 * (c) is a scatter from the few trigger columns (k_restore); (d) and the per-read flags
 * are k_readfix / k_keepfix.  (c) and (d) write the same value, so their order does not matter. */
struct ColOp { uint8_t v; int keep, b15; };

__device__ __forceinline__ uint8_t cap_q2(const DParams *__restrict__ P, int qcap, bool any_pres, uint8_t q)
{   /* cap_quality snp:1325-1332; the preserve_qual table is only consulted when -k/-K/-y set it */
    if (q > qcap && !(any_pres && P->preserve_qual[q])) return (uint8_t)qcap;
    return q;
}

/* the per-base loop body (snp:1870-1919) for the entry (column c, query index x) */
__device__ __forceinline__ void apply_col(const DParams *__restrict__ P, const DBatch &B, uint32_t rec, int c,
                                          const int32_t *__restrict__ pos_mn2, int64_t r, int rp, int x,
                                          uint8_t capq, int base, ColOp &o, bool any_pres)
{
    if (rec & CR_SKIP) return;
    if (rec & CR_KEEP) o.keep = 1;
    const int preserve = CR_PRESERVE(rec);
    if (rec & CR_TRIG) {                                   /* snp:1870-1879 (rare): x == qpos here */
        const uint32_t *cig = B.cigar + B.cig_off[r];
        const int nc = (int)(B.cig_off[r + 1] - B.cig_off[r]);
        if (dev_ref2query_pos(cig, nc, rp, pos_mn2[c]) <= x) o.v = capq | 0x80;
    }
    if (rec & CR_WIN) o.v = capq | 0x80;                   /* snp:1880-1883 */
    if (preserve || (any_pres && P->preserve_qual[o.v & 0x7f] >= 1 + ((rec & CR_PERFECT) ? 1 : 0))) o.v |= 0x80;
    if (preserve > 1) o.b15 = 1;
    if (!(o.v & 0x80)) {                                   /* snp:1906-1919 */
        const int c1 = 1 << CR_A1(rec), c2 = 1 << CR_A2(rec);
        if (base == c1 || base == c2) o.v = (uint8_t)P->qhigh;
        else if (P->reduce_qual) o.v = P->binary_qual ? P->bin2[o.v] : (uint8_t)P->qlow;
    }
}

/* one base through the general path (any CIGAR, -S, -k/-K ...) */
__device__ __forceinline__ uint8_t replay_base(const DBatch &B, const DParams *__restrict__ P,
                                            const uint32_t *__restrict__ colrec, const int32_t *__restrict__ pos_mn2,
                                            int64_t g, int64_t r,
                                            uint8_t *__restrict__ keepf, uint8_t *__restrict__ b15f)
{
    const ReadRec rr = B.rr[r];
    const int x = (int)(g - rr.so);
    const uint8_t q0 = B.quals[g];
    if (rr.end <= rr.pos) return q0 & 0x7f;                /* never entered the pileup */
    const int qcap = P->qcap;
    const bool any_pres = P->any_preserve != 0, softclip = P->softclip != 0;
    const int rp = rr.pos, re = rr.end;
    const bool simple = (rr.info & RR_SIMPLE) != 0;
    const bool lowmq = (int)(rr.info & 255) <= P->min_mqual;
    const uint8_t capq = cap_q2(P, qcap, any_pres, q0);
    const int base = B.bases[g] & 15;
    const Seg *segs = B.segs;
    int ns = 1;
    if (!simple) { segs = B.segs + B.seg_off[r]; ns = (int)(B.seg_off[r + 1] - B.seg_off[r]); }
    ColOp o; o.v = q0; o.keep = 0; o.b15 = 0;
    bool sc_tail = false;
    if (lowmq || softclip) {
        const uint32_t headrec = colrec[rp - B.p0], tailrec = colrec[re - 1 - B.p0];
        if (lowmq && !(headrec & CR_SKIP)) o.v |= 0x80;                                /* snp:1852 */
        if (softclip) {
            int qpos_head = rr.q0, qpos_tail = rr.q0 + (re - rp) - 1;
            if (!simple) {
                const Seg sl = segs[ns - 1];
                qpos_head = segs[0].qstart;
                qpos_tail = (seg_type(sl) == SEG_M) ? sl.qstart + seg_len(sl) - 1 : sl.qstart;
            }
            if (x < qpos_head) {
                /* head column: its back-fill (snp:1875) precedes its -S binning (snp:1895) */
                if ((headrec & CR_TRIG) && !(headrec & CR_SKIP)) {
                    const uint32_t *cig = B.cigar + B.cig_off[r];
                    if (dev_ref2query_pos(cig, (int)(B.cig_off[r + 1] - B.cig_off[r]), rp, pos_mn2[rp - B.p0]) <= x)
                        o.v = capq | 0x80;
                }
                if (!(headrec & (CR_SKIP | CR_KEEP))) o.v = P->bin2[o.v];
            }
            sc_tail = (re - 1 != rp) && !(tailrec & (CR_SKIP | CR_KEEP)) && x > qpos_tail;
        }
    }
    if (simple) {
        const int off = x - rr.q0;
        if (off >= 0 && off < re - rp) {
            const int c = rp + off - B.p0;
            apply_col(P, B, colrec[c], c, pos_mn2, r, rp, x, capq, base, o, any_pres);
        }
    } else {
        /* first segment with qstart >= x */
        int lo = 0, hi = ns;
        while (lo < hi) { int mid = (lo + hi) >> 1; if (segs[mid].qstart < x) lo = mid + 1; else hi = mid; }
        int i = lo;
        bool own_done = false;
        while (i < ns && segs[i].qstart == x) {            /* D/N runs that report x, then maybe x's own M op */
            const Seg sg = segs[i];
            const int len = seg_len(sg);
            if (seg_type(sg) == SEG_M) {
                const int c = rp + sg.roff - B.p0;
                apply_col(P, B, colrec[c], c, pos_mn2, r, rp, x, capq, base, o, any_pres);
                own_done = true;
                break;
            }
            for (int kk = 0; kk < len; kk++) {
                const int c = rp + sg.roff + kk - B.p0;
                apply_col(P, B, colrec[c], c, pos_mn2, r, rp, x, capq, base, o, any_pres);
            }
            i++;
        }
        if (!own_done && i > 0 && i <= ns) {               /* x strictly inside the previous M op? */
            const Seg sg = segs[i - 1];
            if (seg_type(sg) == SEG_M && sg.qstart < x && x < sg.qstart + seg_len(sg)) {
                const int c = rp + sg.roff + (x - sg.qstart) - B.p0;
                apply_col(P, B, colrec[c], c, pos_mn2, r, rp, x, capq, base, o, any_pres);
            }
        }
    }
    if (sc_tail) o.v = P->bin2[o.v];
    if (o.keep && keepf) keepf[r] = 1;                     /* same value from every lane: benign */
    if (o.b15 && b15f) b15f[r] = 1;
    return o.v & 0x7f;                                     /* snp:1093 */
}

/* RPL_NB consecutive bases per lane.  For a base of a read whose alignment is one match
 * op (and no rarely-used option is on) the byte is produced with a handful of selects:
 *   window/back-fill (snp:1880)            -> capped original
 *   preserve, or low-mapq read (snp:1852)  -> original
 *   no column (clip, insertion), skipped   -> original
 *   else (snp:1906-1919)                   -> qhigh if the base is a called allele, else qlow/bin2
 * everything else goes through replay_base(). */
#define RPL_NB 16        /* consecutive bases per lane */
__global__ void __launch_bounds__(256)
k_replay_flat(DBatch B, const DParams *__restrict__ P, const uint32_t *__restrict__ colrec,
              const int32_t *__restrict__ pos_mn2, const int32_t *__restrict__ blk_read,
              uint8_t *__restrict__ out, uint8_t *__restrict__ keepf, uint8_t *__restrict__ b15f)
{
    const int64_t g0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * RPL_NB;
    if (g0 >= B.n_bases) return;
    const int nn = (B.n_bases - g0 < RPL_NB) ? (int)(B.n_bases - g0) : RPL_NB;
    const bool any_pres = P->any_preserve != 0;
    const int min_mqual = P->min_mqual;
    /* the lane's qualities and bases first: nothing below depends on them until phase 3, so their latency runs
     * beside the read-record / column-table / column-record chain of phases 1 and 2 */
    uint32_t qw[RPL_NB / 4], bw[RPL_NB / 4];
    if (nn == RPL_NB) {
        const uint4 q4 = *(const uint4 *)(B.quals + g0), b4 = *(const uint4 *)(B.bases + g0);
        qw[0] = q4.x; qw[1] = q4.y; qw[2] = q4.z; qw[3] = q4.w;
        bw[0] = b4.x; bw[1] = b4.y; bw[2] = b4.z; bw[3] = b4.w;
    } else {
        #pragma unroll
        for (int k = 0; k < RPL_NB / 4; k++) { qw[k] = 0; bw[k] = 0; }
        for (int k = 0; k < nn; k++) { qw[k >> 2] |= (uint32_t)B.quals[g0 + k] << (8 * (k & 3)); bw[k >> 2] |= (uint32_t)B.bases[g0 + k] << (8 * (k & 3)); }
    }
    /* phase 1: owning read and column of each base (reads are laid out back to back in file
     * order; at most a few read changes inside RPL_NB bases) */
    int64_t r = blk_read[g0 >> 6];
    ReadRec rr = B.rr[r];
    while (g0 >= rr.so + rr.lq) { r++; rr = B.rr[r]; }    /* at most a few steps inside a 64-base block */
    const int64_t r0 = r;
    bool hm = false;                                       /* is_head low-mapq mark of the current read */
    if (rr.end > rr.pos && (int)(rr.info & 255) <= min_mqual) hm = !(colrec[rr.pos - B.p0] & CR_SKIP);
    int cidx[RPL_NB];
    unsigned slow = 0, hmm = 0;                            /* bit k */
    uint64_t rdelta = 0;                                   /* 4 bits per k */
    #pragma unroll
    for (int k = 0; k < RPL_NB; k++) {
        cidx[k] = -1;
        if (k < nn) {
            const int64_t g = g0 + k;
            while (g >= rr.so + rr.lq) {                   /* next read (no load unless the read changes) */
                r++; rr = B.rr[r];
                hm = false;
                if (rr.end > rr.pos && (int)(rr.info & 255) <= min_mqual) hm = !(colrec[rr.pos - B.p0] & CR_SKIP);
            }
            rdelta |= (uint64_t)(r - r0) << (4 * k);
            if (rr.end > rr.pos) {                         /* in the pileup */
                if (rr.info & RR_SIMPLE) {
                    const int off = (int)(g - rr.so) - rr.q0;
                    if (off >= 0 && off < rr.end - rr.pos) cidx[k] = rr.pos + off - B.p0;
                    if (hm) hmm |= 1u << k;
                } else {                                   /* general CIGAR: base -> column table built with the batch (k_p0_fill) */
                    const int c = B.cix[(size_t)rr.cix * CIX_UNIT + (size_t)(g - rr.so)];
                    if (c == -2) slow |= 1u << k;          /* D/N columns report this base: sequence matters */
                    else { cidx[k] = c; if (hm) hmm |= 1u << k; }
                }
            }
        }
    }
    /* phase 2: the column records (consecutive words for the common case) */
    uint32_t rec[RPL_NB];
    bool consec = cidx[0] >= 0;
    #pragma unroll
    for (int k = 1; k < RPL_NB; k++) consec = consec && cidx[k] == cidx[k - 1] + 1;
    if (consec) {                                          /* consecutive columns: RPL_NB consecutive words */
        const uint32_t *rc = colrec + cidx[0];
        #pragma unroll
        for (int k = 0; k < RPL_NB; k++) rec[k] = rc[k];
    } else {
        #pragma unroll
        for (int k = 0; k < RPL_NB; k++) { const uint32_t v = colrec[cidx[k] >= 0 ? cidx[k] : 0]; rec[k] = cidx[k] >= 0 ? v : CR_ALLREFSKIP; }
    }
    /* phase 3 */
    const int qcap = P->qcap, qhigh = P->qhigh, qlow = P->qlow;
    const bool reduce = P->reduce_qual != 0, binary = P->binary_qual != 0;
    uint32_t ow[RPL_NB / 4];
    #pragma unroll
    for (int k = 0; k < RPL_NB / 4; k++) ow[k] = 0;
    #pragma unroll
    for (int k = 0; k < RPL_NB; k++) {
        const uint32_t q0 = (qw[k >> 2] >> (8 * (k & 3))) & 255;
        const int base = (bw[k >> 2] >> (8 * (k & 3))) & 15;
        const uint32_t rcd = rec[k];
        const bool skip = (rcd & CR_SKIP) != 0;
        const int64_t rk = r0 + (int64_t)((rdelta >> (4 * k)) & 15);
        uint32_t v;
        /* a trigger column is always a window column (k_columns sets both), so its own byte is
         * the capped original whatever the back-fill start: no special case needed here */
        if ((slow >> k) & 1) {                             /* rare: rewritten by k_replay_slow */
            v = q0;
        } else {
            const uint32_t capq = any_pres ? (uint32_t)cap_q2(P, qcap, true, (uint8_t)q0) : (q0 > (uint32_t)qcap ? (uint32_t)qcap : q0);
            const int pres = CR_PRESERVE(rcd);
            bool keepv = ((hmm >> k) & 1) || skip || pres;                   /* value untouched */
            if (any_pres) keepv = keepv || P->preserve_qual[q0 & 0x7f] >= 1 + ((rcd & CR_PERFECT) ? 1 : 0);   /* snp:1885 */
            const bool match = base == (1 << CR_A1(rcd)) || base == (1 << CR_A2(rcd));
            const uint32_t qz = match ? (uint32_t)qhigh : (reduce ? (binary ? (uint32_t)P->bin2[q0] : (uint32_t)qlow) : q0);
            v = (!skip && (rcd & CR_WIN)) ? capq : (keepv ? q0 : qz);
            if (!skip && (rcd & CR_KEEP)) keepf[rk] = 1;
            if (!skip && pres > 1) b15f[rk] = 1;
        }
        ow[k >> 2] |= (v & 0x7f) << (8 * (k & 3));
    }
    if (nn == RPL_NB) { uint4 o4; o4.x = ow[0]; o4.y = ow[1]; o4.z = ow[2]; o4.w = ow[3]; *(uint4 *)(out + g0) = o4; }
    else for (int k = 0; k < nn; k++) out[g0 + k] = (uint8_t)((ow[k >> 2] >> (8 * (k & 3))) & 255);
}

/* ------------------------------------------------------------------------ */
/* Short-read batches: the replay in packed form instead of one general per-base routine.
 *   k_actions       one byte per column: everything a base needs to know about its column
 *                     bits 0-3  nt16 mask of the called alleles ((1<<A1 | 1<<A2) & 15)
 *                     bit  4    value stays (skipped column, or preserve != 0)
 *                     bit  5    window column: capped original (snp:1880)
 *                     bit  6    the column sets a per-read flag (keep_qual / bit 15): rare follow-up
 *                     bit  7    perfect column (a quality preserved at level 1 is not kept there, snp:1885)
 *   k_readflags     (round 4) the per-read flags BEFORE the pass: keep_qual / flag bit 15 from the columns k_actions
 *                   marked; a kept read's bytes are marked in forcebits (a bit per base), as k_restore marks the
 *                   back-fill ranges of the trigger columns there -- so every byte has ONE writer below
 *   k_replay_fast   one lane per 16-byte chunk of the flat base array.  For every single-match read
 *                   that overlaps the chunk, 16 qualities, 16 bases and the 16 action bytes of the
 *                   columns they would sit on are combined with packed byte arithmetic (values < 128
 *                   leave bit 7 free as a guard), ~9 VALU per base instead of ~80, and the bytes of
 *                   that read's aligned part are taken from the result; marked bytes (forcebits) take the
 *                   capped original; clips and reads outside the pileup keep the input byte, reads with a
 *                   general CIGAR are left alone.  All loads and the store are 16 bytes wide and coalesced.
 *                   STATS: the output histogram, the changed count and the quality check (input >= 128)
 *                   are taken on the way -- no separate pass over the bytes (k_stats) when nothing writes later.
 *   k_replay_reads  the reads with a general CIGAR (a few %): RR_GL lanes per read, four reads per wave,
 *                   the column of every base from the read's op table. */
__device__ __forceinline__ unsigned action_of(uint32_t r)
{
    const bool skip = (r & CR_SKIP) != 0;
    const unsigned pres = CR_PRESERVE(r);
    unsigned a = ((1u << CR_A1(r)) | (1u << CR_A2(r))) & 15u;
    if (skip || pres) a |= 16u;
    if (!skip && (r & CR_WIN)) a |= 32u;
    if (!skip && ((r & CR_KEEP) || pres > 1)) a |= 64u;
    if (r & CR_PERFECT) a |= 128u;
    return a;
}
/* four columns per thread: one 16-byte load, one 4-byte store (the arrays start 16-byte aligned) */
/* flagbits (may be null; zeroed by the caller): one bit per column = action bit 6, "the column raises a per-read flag":
 * k_readflags looks only at those columns */
__global__ void k_actions(int L, const uint32_t *__restrict__ colrec, uint8_t *__restrict__ act, uint32_t *__restrict__ flagbits)
{
    const int c = (int)(blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (c >= L) return;
    unsigned f = 0;
    if (c + 3 < L) {
        const uint4 r = *(const uint4 *)(colrec + c);
        const unsigned a0 = action_of(r.x), a1 = action_of(r.y), a2 = action_of(r.z), a3 = action_of(r.w);
        *(uint32_t *)(act + c) = a0 | (a1 << 8) | (a2 << 16) | (a3 << 24);
        f = ((a0 >> 6) & 1u) | (((a1 >> 6) & 1u) << 1) | (((a2 >> 6) & 1u) << 2) | (((a3 >> 6) & 1u) << 3);
    } else
        for (int i = c; i < L; i++) { const unsigned a = action_of(colrec[i]); act[i] = (uint8_t)a; f |= ((a >> 6) & 1u) << (i - c); }
    if (flagbits && f) atomicOr(&flagbits[c >> 5], f << (c & 31));       /* (c is a multiple of four: the bits stay in the word) */
}

/* bits [a, b) of a zeroed bit array */
#define RF_TRAILING_CB() \
    int cb = c1;                                            /* columns [c0, cb) can raise bit 15 */ \
    if (rr.lq == 0) cb = c0; \
    else if (!(rr.info & RR_SIMPLE)) { \
        const Seg *segs = B.segs + B.seg_off[r]; \
        const int ns = (int)(B.seg_off[r + 1] - B.seg_off[r]); \
        for (int i = ns - 1; i >= 0 && segs[i].qstart >= rr.lq; i--) cb = c0 + segs[i].roff; \
    }
__device__ __forceinline__ void mark_bits(uint32_t *__restrict__ bitsarr, int64_t a, int64_t b)
{
    for (int64_t w = a >> 5; w <= (b - 1) >> 5; w++) {
        const int lo = a > w * 32 ? (int)(a - w * 32) : 0, hi = b < w * 32 + 32 ? (int)(b - w * 32) : 32;
        const uint32_t m = (hi == 32 ? ~0u : ((1u << hi) - 1u)) & ~((1u << lo) - 1u);
        if (m) atomicOr(&bitsarr[w], m);
    }
}

/* The per-read flags BEFORE the replay (packed replay; the byte-wise replays raise them on the way, k_readfix):
 * keep_qual (snp:1847: any column of the read's span sets it -- every column holds an entry of the read, a base, a
 * deletion or a reference skip) and flag bit 15 (snp:1890: columns whose entry comes after the l_qseq test, i.e. not
 * the D/N columns behind the last query base, not a read without SEQ).  Only columns with their bit in flagbits are
 * looked at.  One lane per read. */
__global__ void k_readflags(DBatch B, const uint32_t *__restrict__ colrec, const uint32_t *__restrict__ flagbits,
                            uint8_t *__restrict__ rflags, uint32_t *__restrict__ forcebits)
{   /* a kept read's bytes are marked in forcebits like a back-fill range: the replay pass puts the capped original there
     * (snp:1939) and needs no flag look-up per read */
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= B.n_reads) return;
    const ReadRec rr = B.rr[r];
    if (rr.end <= rr.pos) { rflags[r] = 4; return; }
    const int c0 = rr.pos - B.p0, c1 = rr.end - B.p0;
    int keep = 0, b15 = 0;
    const int w0 = c0 >> 5, w1 = (c1 - 1) >> 5;
    if (w1 - w0 < 8) {                                      /* the usual read: does any of its columns raise a flag at all? (independent loads) */
        uint32_t any = 0;
        #pragma unroll
        for (int k = 0; k < 8; k++) {
            uint32_t bits = (w0 + k <= w1) ? flagbits[w0 + k] : 0u;
            if (k == 0) bits &= ~0u << (c0 & 31);
            if (w0 + k == w1 && ((c1 - 1) & 31) != 31) bits &= (2u << ((c1 - 1) & 31)) - 1u;
            any |= bits;
        }
        if (!any) { rflags[r] = 0; return; }
    }
    RF_TRAILING_CB();                                       /* (after the early exit: a general-CIGAR read's op walk is a chain of dependent
                                                             * loads that every wave with such a read waited for, flags or not) */
    for (int w = w0; w <= w1; w++) {
        uint32_t bits = flagbits[w];
        if (w == (c0 >> 5)) bits &= ~0u << (c0 & 31);
        if (w == ((c1 - 1) >> 5) && ((c1 - 1) & 31) != 31) bits &= (2u << ((c1 - 1) & 31)) - 1u;
        while (bits) {
            const int c = w * 32 + __builtin_ctz(bits);
            bits &= bits - 1;
            const uint32_t rec = colrec[c];
            if (rec & CR_SKIP) continue;
            if (rec & CR_KEEP) keep = 1;
            if (CR_PRESERVE(rec) > 1 && c < cb) b15 = 1;
        }
    }
    rflags[r] = (uint8_t)(keep | (b15 << 1));
    if (keep && rr.lq > 0) mark_bits(forcebits, rr.so, rr.so + rr.lq);
}
/* the same with a wave per read (long reads: hundreds of words of column flags and of byte marks per read) */
__global__ void __launch_bounds__(256)
k_readflags_wave(DBatch B, const uint32_t *__restrict__ colrec, const uint32_t *__restrict__ flagbits,
                 uint8_t *__restrict__ rflags, uint32_t *__restrict__ forcebits)
{
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= B.n_reads) return;                            /* (the whole wave) */
    const ReadRec rr = B.rr[r];
    if (rr.end <= rr.pos) { if (lane == 0) rflags[r] = 4; return; }
    const int c0 = rr.pos - B.p0, c1 = rr.end - B.p0;
    RF_TRAILING_CB();
    int keep = 0, b15 = 0;
    const int w0 = c0 >> 5, w1 = (c1 - 1) >> 5;
    for (int w = w0 + lane; w <= w1; w += 64) {
        uint32_t bits = flagbits[w];
        if (w == w0) bits &= ~0u << (c0 & 31);
        if (w == w1 && ((c1 - 1) & 31) != 31) bits &= (2u << ((c1 - 1) & 31)) - 1u;
        while (bits) {
            const int c = w * 32 + __builtin_ctz(bits);
            bits &= bits - 1;
            const uint32_t rec = colrec[c];
            if (rec & CR_SKIP) continue;
            if (rec & CR_KEEP) keep = 1;
            if (CR_PRESERVE(rec) > 1 && c < cb) b15 = 1;
        }
    }
    keep = __ballot(keep) != 0ull; b15 = __ballot(b15) != 0ull;
    if (lane == 0) rflags[r] = (uint8_t)(keep | (b15 << 1));
    if (keep && rr.lq > 0) {
        const int64_t a = rr.so, b = rr.so + rr.lq;
        for (int64_t w = (a >> 5) + lane; w <= (b - 1) >> 5; w += 64) {
            const int lo = a > w * 32 ? (int)(a - w * 32) : 0, hi = b < w * 32 + 32 ? (int)(b - w * 32) : 32;
            const uint32_t m = (hi == 32 ? ~0u : ((1u << hi) - 1u)) & ~((1u << lo) - 1u);
            if (m) atomicOr(&forcebits[w], m);
        }
    }
}

/* per-workgroup histogram of output qualities in LDS (signed counts: a corrected byte takes one out and puts one in);
 * flushed into the statistics vector at the end */
__device__ __forceinline__ void hist_flush(unsigned *h, u64 *stats, unsigned changed_net)
{
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += blockDim.x) {
        const int v = (int)h[i];
        if (v) atomicAdd(&stats[ST_HIST0 + i], (u64)(long long)v);
    }
    if (changed_net) atomicAdd(&stats[ST_BASES_CHANGED], (u64)(long long)(int)changed_net);
}

#define PB4(x)       ((uint32_t)(x) * 0x01010101u)
/* Packed-byte flags live in bit 7 of each byte; a full 0x00 / 0xff byte mask is made from them by ONE v_perm_b32 whose
 * selector bytes are 0x0c (constant 0x00) or 0x0d (constant 0xff), and a per-byte choice between two words by one v_perm_b32
 * whose selector bytes are i (byte i of b) or 4 + i (byte i of a).  (Round 5: the `* 0xff` of the rounds before is a
 * v_mul_lo_u32 -- quarter rate -- and k_replay_fast issued forty of them per sixteen bases.) */
__device__ __forceinline__ uint32_t pb_full7(uint32_t f) { return kc2_perm(0u, 0u, ((f >> 7) & 0x01010101u) | 0x0c0c0c0cu); }
__device__ __forceinline__ uint32_t pb_sel7(uint32_t f, uint32_t a, uint32_t b) { return kc2_perm(a, b, ((f >> 5) & 0x04040404u) | 0x03020100u); }
#define PB_GE(a, b)  pb_full7(((a) | 0x80808080u) - (b))     /* bytes where a >= b (a, b < 128) */
#define PB_NZ(x)     pb_full7((x) + 0x7f7f7f7fu)             /* bytes where x != 0 (x < 128)    */
/* the four-byte mask of the four bits of x: the bits spread to bit 0 of each byte (24-bit multiply: full rate), then the
 * 0x0c / 0x0d selector */
__device__ __forceinline__ uint32_t pb_nib(uint32_t x) { return kc2_perm(0u, 0u, (((x & 15u) * 0x00204081u) & 0x01010101u) | 0x0c0c0c0cu); }
#define ACT_PAD 64   /* bytes in front of act[0]: a chunk may start up to 15 columns before the read */

/* the owner of a byte range: a read (short-read batches: ReadRec) or a territory of a read (long-read batches:
 * VRec, one per reference-consuming op); both carry so, pos, end, lq, q0 */
__device__ __forceinline__ bool rf_aligned(const ReadRec &x) { return (x.info & RR_SIMPLE) && x.end > x.pos; }
__device__ __forceinline__ bool rf_aligned(const VRec &x) { return x.end > x.pos; }
/* low-mapq read whose head column is live: snp:1852 */
__device__ __forceinline__ bool rf_headmark(const ReadRec &x, const DBatch &B, int min_mqual, const uint32_t *__restrict__ colrec)
{ return (int)(x.info & 255) <= min_mqual && !(colrec[x.pos - B.p0] & CR_SKIP); }
__device__ __forceinline__ bool rf_headmark(const VRec &x, const DBatch &, int, const uint32_t *__restrict__ colrec)
{ return x.hcol >= 0 && !(colrec[x.hcol] & CR_SKIP); }
__device__ __forceinline__ int64_t rf_read(const ReadRec &, int64_t r) { return r; }
__device__ __forceinline__ int64_t rf_read(const VRec &x, int64_t) { return x.read; }
template <bool VREC> struct RfSel { typedef ReadRec T; static __device__ __forceinline__ const T *tab(const DBatch &B) { return B.rr; } };
template <> struct RfSel<true> { typedef VRec T; static __device__ __forceinline__ const T *tab(const DBatch &B) { return B.vr; } };

/* forcebits: the bytes of the kept reads (keep_qual, k_readflags) and the back-fill ranges of the trigger columns
 * (k_restore's marks) are known BEFORE this pass, so it is the one writer of every byte it owns: the marked bytes are
 * the capped original (snp:1939, 1875).  STATS: the histogram of the output and the changed count are taken
 * here, from registers, instead of by a pass of their own over output and input (k_stats: 2 B per base): bytes of
 * general-CIGAR reads are left to k_replay_reads (short reads), the bases D/N columns report are corrected by
 * k_replay_slow.  The two values nearly every output byte has (qhigh, qlow) are counted with packed compares and
 * popcounts into registers; only the other bytes go through LDS atomics.  It is also the pass that looks at every
 * INPUT quality: a value >= 128 raises *bad_qual.  Persistent: the grid strides over the chunks. */
#ifndef RF_WAVES
#define RF_ATTR
#else
#define RF_ATTR __attribute__((amdgpu_waves_per_eu(RF_WAVES, RF_WAVES)))
#endif
template <bool VREC, bool PV, bool STATS>
__global__ void __launch_bounds__(256) RF_ATTR
k_replay_fast(DBatch B, const DParams *__restrict__ P, const uint32_t *__restrict__ colrec,
              const uint8_t *__restrict__ act, const int32_t *__restrict__ chunk_read,
              uint8_t *__restrict__ out, const uint32_t *__restrict__ forcebits,
              u64 *__restrict__ stats, int *__restrict__ bad_qual)
{
    __shared__ unsigned hist[STATS ? 256 : 1];
    unsigned cnt_h = 0, cnt_l = 0, cnt_chg = 0, hibits = 0;
    if (STATS) { for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0; __syncthreads(); }
    const uint32_t QC = PB4(P->qcap), QH = PB4(P->qhigh), QL = PB4(P->qlow), CUT = PB4(P->qcutoff);
    const bool reduce = P->reduce_qual != 0, binary = P->binary_qual != 0;
    const int min_mqual = P->min_mqual, npv = PV ? P->n_pv : 0;
    const int64_t n_owner = VREC ? B.n_vrec : B.n_reads;
    typedef typename RfSel<VREC>::T Rec;
    const Rec *const recs = RfSel<VREC>::tab(B);
    const int64_t nchunks = (B.n_bases + 15) >> 4;
    for (int64_t ch = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; ch < nchunks; ch += (int64_t)gridDim.x * blockDim.x) {
    const int64_t g0 = ch * 16;
    const int64_t gend = (g0 + 16 < B.n_bases) ? g0 + 16 : B.n_bases;
    const uint4 q4 = *(const uint4 *)(B.quals + g0), b4 = *(const uint4 *)(B.bases + g0);   /* buffers carry slack */
    const unsigned fb16 = *(const uint16_t *)((const uint8_t *)forcebits + (g0 >> 3));      /* the chunk's back-fill marks */
    const uint32_t qw[4] = {q4.x, q4.y, q4.z, q4.w}, bw[4] = {b4.x, b4.y, b4.z, b4.w};
    if (gend - g0 == 16) hibits |= q4.x | q4.y | q4.z | q4.w;
    else for (int k = 0; k < (int)(gend - g0); k++) hibits |= qw[k >> 2] & (0xffu << (8 * (k & 3)));     /* (the slack behind the last base is not input) */
    uint32_t ow[4] = {qw[0] & 0x7f7f7f7fu, qw[1] & 0x7f7f7f7fu, qw[2] & 0x7f7f7f7fu, qw[3] & 0x7f7f7f7fu};
    /* the capped original of every byte (cap_quality snp:1325; -k / -K: a preserved quality is not capped): window columns,
     * back-fill ranges and kept reads take it */
    uint32_t capw[4], pv1w[4], pv2w[4];
    #pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t q = qw[k];
        uint32_t PV1 = 0, PV2 = 0;
        if (PV)
            for (int i = 0; i < npv; i++) {
                const uint32_t eq = ~PB_NZ(q ^ PB4(P->pv_val[i]));
                PV1 |= eq;
                if (P->pv_lvl[i] > 1) PV2 |= eq;
            }
        capw[k] = pb_sel7((((q | 0x80808080u) - QC) & ~PV1), QC, q);
        pv1w[k] = PV1; pv2w[k] = PV2;
    }
    int64_t r = chunk_read[g0 >> 4];                          /* the owner of byte g0 (table built with the batch) */
    Rec rr = recs[r];
    unsigned genb = 0;                                        /* bytes of general-CIGAR reads (k_replay_reads') */
    /* The chunk's bytes belong to one read, or -- at a read boundary, i.e. in some lane of nearly every
     * wave -- to two: up to two reads are gathered into slots and done in ONE pass of the packed body
     * (per-byte merge of their action bytes); only chunks that touch more than two single-match reads
     * (reads shorter than a chunk) go round again. */
    for (;;) {
        unsigned vb[2] = {0, 0};
        int c0s[2] = {0, 0};
        uint32_t hms[2] = {0, 0};
        int ns = 0;
        bool more = false;
        for (;;) {
            if (!VREC && STATS && rr.end > rr.pos && !rf_aligned(rr)) {       /* a general-CIGAR read's bytes: k_replay_reads counts them */
                const int64_t o0 = rr.so, o1 = rr.so + rr.lq;
                const int lo = (o0 > g0) ? (int)(o0 - g0) : 0, hi = (o1 < gend) ? (int)(o1 - g0) : (int)(gend - g0);
                if (lo < hi) genb |= ((1u << hi) - 1u) & ~((1u << lo) - 1u);
            }
            if (rf_aligned(rr)) {
                /* bytes of this chunk that are aligned bases of this read */
                const int64_t a0 = rr.so + rr.q0, a1 = a0 + (rr.end - rr.pos);
                const int lo = (a0 > g0) ? (int)(a0 - g0) : 0, hi = (a1 < gend) ? (int)(a1 - g0) : (int)(gend - g0);
                if (lo < hi) {
                    if (ns == 2) { more = true; break; }           /* a third read: next round */
                    const unsigned m = ((1u << hi) - 1u) & ~((1u << lo) - 1u);
                    const int c0 = rr.pos + (int)(g0 - a0) - B.p0; /* column of byte 0 (may be < 0: padded) */
                    uint32_t hm = 0;                               /* low-mapq read whose head column is live: snp:1852 */
                    if (rf_headmark(rr, B, min_mqual, colrec)) hm = 0xffffffffu;
                    if (ns == 0) { vb[0] = m; c0s[0] = c0; hms[0] = hm; }
                    else         { vb[1] = m; c0s[1] = c0; hms[1] = hm; }
                    ns++;
                }
            }
            if (rr.so + rr.lq >= gend || r + 1 >= n_owner) break;
            r++; rr = recs[r];
        }
        if (ns) {
            uint32_t a[4], a2[4] = {0, 0, 0, 0};
            #define LOAD_ACT16(dst, col) do {   /* 16 action bytes from an arbitrary byte address: aligned dwords + v_alignbyte */ \
                const uint8_t *ap_ = act + (col); \
                const unsigned sh_ = (unsigned)((uintptr_t)ap_ & 3u); \
                const uint32_t *aw_ = (const uint32_t *)(ap_ - sh_); \
                const uint32_t w0_ = aw_[0], w1_ = aw_[1], w2_ = aw_[2], w3_ = aw_[3], w4_ = aw_[4]; \
                dst[0] = kc2_alignbyte(w1_, w0_, sh_); dst[1] = kc2_alignbyte(w2_, w1_, sh_); \
                dst[2] = kc2_alignbyte(w3_, w2_, sh_); dst[3] = kc2_alignbyte(w4_, w3_, sh_); } while (0)
            LOAD_ACT16(a, c0s[0]);
            if (ns > 1) LOAD_ACT16(a2, c0s[1]);
            #undef LOAD_ACT16
            #pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t q = qw[k], b = bw[k];
                const uint32_t V1 = pb_nib(vb[0] >> (4 * k)), V2 = pb_nib(vb[1] >> (4 * k)), V = V1 | V2;   /* the first / second read's bytes */
                const uint32_t ac = (a[k] & V1) | (a2[k] & V2);
                const uint32_t hm = (hms[0] & V1) | (hms[1] & V2);
                const uint32_t PV1 = pv1w[k], PV2 = pv2w[k], capq = capw[k];      /* -k / -K: bytes of a preserved quality; cap_quality */
                /* (flags: bit 7 of each byte) */
                const uint32_t hit = b & ac & 0x0f0f0f0fu;                          /* base shares a bit with a called allele */
                const uint32_t many = b & ((b | 0x80808080u) - 0x01010101u) & 0x0f0f0f0fu;   /* >= 2 bits: never equals a call */
                const uint32_t fM = (hit + 0x7f7f7f7fu) & ~(many + 0x7f7f7f7fu);   /* base == call1 || base == call2 (snp:1908) */
                uint32_t nm = q;
                if (reduce) {
                    if (binary) nm = pb_sel7((q | 0x80808080u) - CUT, QH, QL);     /* bin2[], no -K here */
                    else nm = QL;
                }
                const uint32_t qz = pb_sel7(fM, QH, nm);
                uint32_t fU = (ac << 3) | hm;                                       /* snp:1885 */
                if (PV) fU |= PV2 | (PV1 & ~ac);
                const uint32_t v = pb_sel7(ac << 2, capq, pb_sel7(fU, q, qz)) & 0x7f7f7f7fu;      /* window columns: the capped original */
                ow[k] = (v & V) | (ow[k] & ~V);
            }
        }
        if (!more) break;
    }
    const unsigned cq = fb16;                                  /* kept read or back-fill: the capped original, whatever the columns said */
    if (cq) {
        #pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t C = pb_nib(cq >> (4 * k));
            ow[k] = (capw[k] & 0x7f7f7f7fu & C) | (ow[k] & ~C);
        }
    }
    if (gend - g0 == 16) { uint4 o4; o4.x = ow[0]; o4.y = ow[1]; o4.z = ow[2]; o4.w = ow[3]; *(uint4 *)(out + g0) = o4; }
    else for (int k = 0; k < (int)(gend - g0); k++) out[g0 + k] = (uint8_t)((ow[k >> 2] >> (8 * (k & 3))) & 255);
    if (STATS) {
        unsigned valid = (gend - g0 == 16) ? 0xffffu : ((1u << (int)(gend - g0)) - 1u);
        valid &= ~genb;
        #pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t Vm = ((((valid >> (4 * k)) & 15u) * 0x00204081u) & 0x01010101u) << 7;            /* bit 7 of the bytes counted here */
            const uint32_t o = ow[k], q = qw[k] & 0x7f7f7f7fu;
            const uint32_t eh = ~((o ^ QH) + 0x7f7f7f7fu) & Vm;                                             /* bytes equal to qhigh: (x + 0x7f) has bit 7 clear iff x == 0 */
            const uint32_t el = ~((o ^ QL) + 0x7f7f7f7fu) & Vm & ~eh;
            const uint32_t ne = ((o ^ q) + 0x7f7f7f7fu) & Vm;                                               /* bytes that differ from the input */
            cnt_h += __popc(eh); cnt_l += __popc(el); cnt_chg += __popc(ne);
            const uint32_t other = Vm & ~eh & ~el;                   /* the rest (original values, bin2 levels): LDS atomics, no loop */
            if (other & 0x00000080u) atomicAdd(&hist[o & 255u], 1u);
            if (other & 0x00008000u) atomicAdd(&hist[(o >> 8) & 255u], 1u);
            if (other & 0x00800000u) atomicAdd(&hist[(o >> 16) & 255u], 1u);
            if (other & 0x80000000u) atomicAdd(&hist[o >> 24], 1u);
        }
    }
    }
    if ((hibits & 0x80808080u) && bad_qual) *bad_qual = 1;
    if (STATS) {
        /* the three register counters: one LDS add per wave */
        #pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { cnt_h += __shfl_xor(cnt_h, d); cnt_l += __shfl_xor(cnt_l, d); cnt_chg += __shfl_xor(cnt_chg, d); }
        __shared__ unsigned chg_all;
        if (threadIdx.x == 0) chg_all = 0;
        __syncthreads();
        if ((threadIdx.x & 63) == 0) {
            if (cnt_h) atomicAdd(&hist[P->qhigh & 255], cnt_h);
            if (cnt_l) atomicAdd(&hist[P->qlow & 255], cnt_l);
            if (cnt_chg) atomicAdd(&chg_all, cnt_chg);
        }
        __syncthreads();
        hist_flush(hist, stats, threadIdx.x == 0 ? chg_all : 0u);
    }
}

#define RR_GL 16
/* STATS: this pass is the one writer of the general-CIGAR reads' bytes (k_replay_fast leaves them out of its count) */
template <bool STATS>
__global__ void __launch_bounds__(256)
k_replay_reads(DBatch B, const DParams *__restrict__ P, const uint32_t *__restrict__ colrec,
               const int32_t *__restrict__ gen_list, const int *__restrict__ n_gen,
               uint8_t *__restrict__ out, const uint32_t *__restrict__ forcebits,
               u64 *__restrict__ stats)
{
    /* RR_GL lanes per general-CIGAR read of the list the upload made (k_p0_fill; it used to find them by scanning
     * every read record: 0.34 ms on the chr20 set for 3 % of the reads).  A read is a chain of dependent look-ups
     * (list -> record -> op table -> column records) in front of a few dozen bytes of work: four reads share a wave
     * so that four chains run side by side (a wave per read: 0.46 ms on the chr20 set) */
    __shared__ unsigned hist[STATS ? 256 : 1];
    unsigned cnt_chg = 0;
    if (STATS) { for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0; __syncthreads(); }
    const int gl = threadIdx.x & (RR_GL - 1);
    const int qcap = P->qcap, qhigh = P->qhigh, qlow = P->qlow;
    const bool reduce = P->reduce_qual != 0, binary = P->binary_qual != 0, any_pres = P->any_preserve != 0;
    const int ng = *n_gen;
    const int nw = (int)(gridDim.x * (blockDim.x / RR_GL));
    for (int gi = (int)(blockIdx.x * (blockDim.x / RR_GL) + (threadIdx.x / RR_GL)); gi < ng; gi += nw) {
        const int64_t rs = gen_list[gi];
        const ReadRec rr = B.rr[rs];
        if (!(rr.lq > 0 && rr.end > rr.pos && !(rr.info & RR_SIMPLE))) continue;
        const int n = rr.lq;
        const int hms = ((int)(rr.info & 255) <= P->min_mqual && !(colrec[rr.pos - B.p0] & CR_SKIP)) ? 1 : 0;
        const int64_t s0 = rr.so;
        const int64_t sg0 = B.seg_off[rs];
        const int nsg = (int)(B.seg_off[rs + 1] - sg0);
        for (int x0 = gl; x0 < n; x0 += 4 * RR_GL) {           /* four bases per lane: the gathers of all four in flight */
            /* the column of every base from the read's op table (a few ops, wave-uniform look-ups): -1 no column, -2 a
             * D/N column reports this base (k_replay_slow, afterwards).  (Rounds 1-3 read this from a base -> column
             * table written at upload: 4 bytes per base of every general read and a kernel of its own.) */
            int c4[4];
            uint32_t q4[4], b4[4], r4[4];
            bool f4[4];
            #pragma unroll
            for (int u = 0; u < 4; u++) {
                const int x = x0 + RR_GL * u;
                c4[u] = -1; q4[u] = 0; b4[u] = 0; f4[u] = false;
                if (x < n) {
                    q4[u] = B.quals[s0 + x]; b4[u] = B.bases[s0 + x];
                    f4[u] = ((forcebits[(s0 + x) >> 5] >> ((s0 + x) & 31)) & 1u) != 0;      /* a kept read, or a trigger column's back-fill reaches it */
                }
            }
            for (int k = 0; k < nsg; k++) {
                const Seg sg = B.segs[sg0 + k];
                const bool isM = seg_type(sg) == SEG_M;
                const int col0 = rr.pos + sg.roff - B.p0;
                #pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int off = x0 + RR_GL * u - sg.qstart;
                    if (isM) { if (off >= 0 && off < seg_len(sg) && c4[u] != -2) c4[u] = col0 + off; }
                    else if (off == 0) c4[u] = -2;
                }
            }
            #pragma unroll
            for (int u = 0; u < 4; u++) if (x0 + RR_GL * u >= n) c4[u] = -1;
            #pragma unroll
            for (int u = 0; u < 4; u++) r4[u] = c4[u] >= 0 ? colrec[c4[u]] : 0u;
            #pragma unroll
            for (int u = 0; u < 4; u++) {
                const int x = x0 + RR_GL * u;
                if (x >= n) continue;
                const uint32_t q0 = q4[u];
                const uint32_t capq = any_pres ? (uint32_t)cap_q2(P, qcap, true, (uint8_t)q0) : (q0 > (uint32_t)qcap ? (uint32_t)qcap : q0);
                uint32_t v = q0;
                if (c4[u] >= 0) {
                    const uint32_t rcd = r4[u];
                    const bool skip = (rcd & CR_SKIP) != 0;
                    const int pres = CR_PRESERVE(rcd), base = (int)(b4[u] & 15);
                    const bool match = base == (1 << CR_A1(rcd)) || base == (1 << CR_A2(rcd));
                    const uint32_t qz = match ? (uint32_t)qhigh : (reduce ? (binary ? (uint32_t)P->bin2[q0] : (uint32_t)qlow) : q0);
                    bool keepv = hms || skip || pres;
                    if (any_pres) keepv = keepv || P->preserve_qual[q0 & 0x7f] >= 1 + ((rcd & CR_PERFECT) ? 1 : 0);   /* snp:1885 */
                    v = (!skip && (rcd & CR_WIN)) ? capq : (keepv ? q0 : qz);
                }
                if (f4[u]) v = capq;
                v &= 0x7f;
                out[s0 + x] = (uint8_t)v;
                if (STATS) { atomicAdd(&hist[v], 1u); cnt_chg += (v != (q0 & 0x7fu)) ? 1u : 0u; }
            }
        }
    }
    if (STATS) {
        #pragma unroll
        for (int d = 32; d >= 1; d >>= 1) cnt_chg += __shfl_xor(cnt_chg, d);
        __shared__ unsigned chg_all;
        if (threadIdx.x == 0) chg_all = 0;
        __syncthreads();
        if ((threadIdx.x & 63) == 0 && cnt_chg) atomicAdd(&chg_all, cnt_chg);
        __syncthreads();
        hist_flush(hist, stats, threadIdx.x == 0 ? chg_all : 0u);
    }
}
#undef PB4

/* the bases that D/N columns report (list built with the batch), after the chunk kernels.  forcebits (packed replay):
 * a kept read's byte and a back-fill's are the capped original whatever the columns say; STATS: the byte was
 * counted with the value the pass before left -- one out, one in. */
template <bool STATS>
__global__ void k_replay_slow(DBatch B, const DParams *__restrict__ P, const uint32_t *__restrict__ colrec,
                              const int32_t *__restrict__ pos_mn2, const int *__restrict__ n_slow, const int64_t *__restrict__ slow_g,
                              const int32_t *__restrict__ slow_r, uint8_t *__restrict__ out,
                              uint8_t *__restrict__ keepf, uint8_t *__restrict__ b15f,
                              const uint32_t *__restrict__ forcebits, u64 *__restrict__ stats)
{
    __shared__ unsigned hist[STATS ? 256 : 1];
    __shared__ unsigned chg_all;
    if (STATS) { for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0; if (threadIdx.x == 0) chg_all = 0; __syncthreads(); }
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < *n_slow && slow_r[t] >= 0) {                   /* (-1: a slot whose base another D/N op reports) */
        const int64_t g = slow_g[t];
        uint8_t v = replay_base(B, P, colrec, pos_mn2, g, slow_r[t], keepf, b15f);
        if (forcebits && ((forcebits[g >> 5] >> (g & 31)) & 1u)) v = cap_q2(P, P->qcap, P->any_preserve != 0, B.quals[g]) & 0x7f;
        if (STATS) {
            const unsigned old = out[g], qin = B.quals[g] & 0x7fu;
            if (old != v) {
                atomicAdd(&hist[old], 0xffffffffu); atomicAdd(&hist[v], 1u);
                atomicAdd(&chg_all, (unsigned)((v != qin) ? 1 : 0) - (unsigned)((old != qin) ? 1 : 0));
            }
        }
        out[g] = v;
    }
    if (STATS) { __syncthreads(); hist_flush(hist, stats, threadIdx.x == 0 ? chg_all : 0u); }
}

/* every base through the general routine (used when -S is on) */
__global__ void k_replay_general(DBatch B, const DParams *__restrict__ P, const uint32_t *__restrict__ colrec,
                                 const int32_t *__restrict__ pos_mn2, const int32_t *__restrict__ blk_read,
                                 uint8_t *__restrict__ out, uint8_t *__restrict__ keepf, uint8_t *__restrict__ b15f)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= B.n_bases) return;
    int64_t r = blk_read[g >> 6];
    while (B.seq_off[r + 1] <= g) r++;
    out[g] = replay_base(B, P, colrec, pos_mn2, g, r, keepf, b15f);
}

/* (c): back-fill of every read under a trigger column (snp:1870-1879), bases before
 * the entry's own qpos (the entry's own byte is handled by apply_col). */
__global__ void __launch_bounds__(64)
k_restore(DBatch B, const DParams *__restrict__ P, int n_trig_, const int *n_dev, const int *spec_abort, const int32_t *trigcol_pos,
          const int32_t *__restrict__ pos_mn2, const int32_t *__restrict__ trig_mx2, uint8_t *__restrict__ out, int coop,
          uint32_t *__restrict__ forcebits)
{   /* forcebits (packed replay; zeroed by the caller): the ranges are MARKED, one bit per base, and the replay pass --
     * the one writer of every byte then -- puts the capped original there; otherwise they are written here */
    if (LIST_ABORTED(spec_abort)) return;
    const int n_trig = LIST_N(n_trig_, n_dev);
    /* one wave per trigger column.  Step 1: every lane resolves one candidate read (entry, start of
     * the back-fill in query coordinates).  Step 2: the wave writes the ranges one read at a time,
     * lanes along the bytes, so a range is one or two coalesced stores instead of a byte loop per
     * lane that touches 64 different cache lines per step. */
    /* long reads (!coop): a column has a few candidates, each lane works alone -- two columns per wave, 32 lanes each
     * (the kernel waits on chains of dependent loads: half the waves, the same chains) */
    const int grp = coop ? 64 : 8;
    for (int tb = (int)blockIdx.x; (coop ? tb : tb * 8) < n_trig; tb += (int)gridDim.x) {
    const int t = coop ? tb : tb * 8 + (int)(threadIdx.x >> 3);
    if (t >= n_trig) continue;
    const int lane = coop ? (int)threadIdx.x : (int)(threadIdx.x & 7);
    const int c = trigcol_pos[t], apos = B.p0 + c, tile = c >> 6;
    const int lo = B.tile_lo[tile], hi = B.tile_hi[tile];
    const int mn2 = pos_mn2[c], qcap = P->qcap;
    const bool any_pres = P->any_preserve != 0, softclip = P->softclip != 0;
    /* The next trigger column, when no reset lies between (pos <= max_pos2 after this column, snp:1508): min_pos2 only
     * falls along such a chain, so for a one-match-op read that also covers the next column, the next column's range
     * [ref2query_pos(min_pos2'), qpos') contains this one's and writes the same bytes -- this column leaves the read
     * to it (runs of trigger columns over an STR rewrote the same bytes once per column) */
    const int nxt_apos = (t + 1 < n_trig && B.p0 + trigcol_pos[t + 1] <= trig_mx2[t]) ? B.p0 + trigcol_pos[t + 1] : INT_MAX;
    for (int r0 = lo; r0 < hi; r0 += grp) {
        const int r = r0 + lane;
        int start = 0, stop = 0;
        unsigned so_lo = 0, so_hi = 0;
        if (r < hi) {
            const ReadRec rr = B.rr[r];
            const int rp = rr.pos;
            if ((rr.info & RR_SIMPLE) && apos >= rp && apos < rr.end && !(softclip && apos == rp)) {
                /* one match op (clips around it at most): qpos and ref2query_pos (snp:1156) in closed form
                 * from the 32-byte read record instead of the op table, the offsets and the CIGAR */
                const int qp = rr.q0 + (apos - rp);
                const bool later = nxt_apos < rr.end && rr.q0 + (nxt_apos - rp) < rr.lq;
                if (qp < rr.lq && !later) {
                    start = mn2 <= rp ? 0 : rr.q0 + (mn2 - rp);
                    stop = qp;
                    so_lo = (unsigned)(uint64_t)rr.so; so_hi = (unsigned)((uint64_t)rr.so >> 32);
                }
            } else
            if (!(rr.info & RR_SIMPLE) && apos >= rp && apos < rr.end && !(softclip && apos == rp)) {   /* head column with -S: done by the replay pass */
                /* qpos of an entry is all that is needed here; the read's ops near the column are found once: the next
                 * trigger column's op by search, this column's and min_pos2's by walking back from it (they are a few
                 * ops apart at most -- a search each was three chains of dependent loads per candidate) */
                const int64_t so = rr.so, sgo = B.seg_off[r];
                const int lq = rr.lq, ns = (int)(B.seg_off[r + 1] - sgo);
                const Seg *const sg = B.segs + sgo;
                /* the same deferral for a read with several ops: the chain's next trigger column also reports a base of
                 * this read (its entry's qpos' >= this one's, its min_pos2' <= this one's): its range contains this one.
                 * Looked at first: along a chain every column but the last leaves the read alone. */
                int si = -1;
                bool later = false;
                if (lq && nxt_apos < rr.end) {
                    si = seg_find(sg, ns, nxt_apos - rp);
                    const Seg s2 = sg[si];
                    later = (seg_type(s2) == SEG_M ? s2.qstart + (nxt_apos - rp - s2.roff) : s2.qstart) < lq;
                }
                if (lq && !later) {
                    const int off = apos - rp;
                    if (si >= 0) { for (int k = 0; k < 8 && si > 0 && off < sg[si].roff; k++) si--; }
                    if (si < 0 || off < sg[si].roff) si = seg_find(sg, ns, off);
                    const Seg s1 = sg[si];
                    const int qpos = seg_type(s1) == SEG_M ? s1.qstart + (off - s1.roff) : s1.qstart;
                    if (qpos < lq) {                             /* snp:1861 / fence */
                        /* ref2query_pos(min_pos2) (dev_ref2query_seg): the first op whose end reaches it */
                        const int offm = mn2 - rp;
                        if (offm <= 0) start = 0;
                        else if (offm > off) start = dev_ref2query_seg(sg, ns, rp, mn2, lq);
                        else {
                            int km = si, k = 0;
                            for (; k < 8 && km > 0 && sg[km - 1].roff + seg_len(sg[km - 1]) >= offm; k++) km--;
                            if (k == 8) start = dev_ref2query_seg(sg, ns, rp, mn2, lq);
                            else { const Seg sm = sg[km]; start = seg_type(sm) == SEG_M ? sm.qstart + (offm - sm.roff) : sm.qstart; }
                        }
                        stop = qpos;
                        so_lo = (unsigned)(uint64_t)so; so_hi = (unsigned)((uint64_t)so >> 32);
                    }
                }
            }
        }
        if (forcebits) {
            const int64_t so = (int64_t)(((uint64_t)so_hi << 32) | so_lo);
            if (start < stop) mark_bits(forcebits, so + start, so + stop);
            continue;
        }
        if (!coop) {                                       /* long reads: few candidates per column, a lane each */
            const int64_t so = (int64_t)(((uint64_t)so_hi << 32) | so_lo);
            for (int x = start; x < stop; x += 8) {         /* eight loads in flight, then eight stores (a byte per trip waited out a load each) */
                uint8_t v8[8];
                #pragma unroll
                for (int k = 0; k < 8; k++) v8[k] = x + k < stop ? B.quals[so + x + k] : (uint8_t)0;
                #pragma unroll
                for (int k = 0; k < 8; k++) if (x + k < stop) out[so + x + k] = cap_q2(P, qcap, any_pres, v8[k]) & 0x7f;
            }
            continue;
        }
        /* four ranges per step: their loads are in flight together (one range per step left a wave waiting
         * out a full load latency for every read of the column: 0.63 -> 0.3x ms on the chr20 set) */
        unsigned long long todo = __ballot(start < stop);
        while (todo) {
            int s4[4], e4[4];
            int64_t so4[4];
            #pragma unroll
            for (int u = 0; u < 4; u++) {
                s4[u] = e4[u] = 0; so4[u] = 0;
                if (todo) {                                        /* (uniform) */
                    const int src = __builtin_ctzll(todo);
                    todo &= todo - 1;
                    s4[u] = READLANE(start, src); e4[u] = READLANE(stop, src);
                    so4[u] = (int64_t)(((uint64_t)(unsigned)READLANE(so_hi, src) << 32) | (unsigned)READLANE(so_lo, src));
                }
            }
            uint8_t v4[4];
            #pragma unroll
            for (int u = 0; u < 4; u++) { const int x = s4[u] + lane; v4[u] = x < e4[u] ? B.quals[so4[u] + x] : (uint8_t)0; }
            #pragma unroll
            for (int u = 0; u < 4; u++) { const int x = s4[u] + lane; if (x < e4[u]) out[so4[u] + x] = cap_q2(P, qcap, any_pres, v4[u]) & 0x7f; }
            #pragma unroll
            for (int u = 0; u < 4; u++)                            /* ranges longer than a wave: the rest */
                for (int x = s4[u] + lane + 64; x < e4[u]; x += 64)
                    out[so4[u] + x] = cap_q2(P, qcap, any_pres, B.quals[so4[u] + x]) & 0x7f;
        }
    }
    }
}

/* (d) + per-read flags: one thread per read; reads whose keep_qual is set are queued */
__global__ void k_readfix(DBatch B, const uint32_t *__restrict__ colrec,
                          const uint8_t *__restrict__ keepf, const uint8_t *__restrict__ b15f,
                          uint8_t *__restrict__ rflags, int32_t *__restrict__ kept, int *__restrict__ n_kept)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= B.n_reads) return;
    const ReadRec rr = B.rr[r];
    if (rr.end <= rr.pos) { rflags[r] = 4; return; }
    int keep = keepf[r];
    const int lq = rr.lq, rp = rr.pos, re = rr.end;
    /* columns that no base owns: l_qseq == 0, or D/N ops at the very end of the query (fenced) */
    if (lq == 0) {
        for (int c = rp; c < re; c++) { const uint32_t rec = colrec[c - B.p0]; if (!(rec & CR_SKIP) && (rec & CR_KEEP)) keep = 1; }
    } else if (!(rr.info & RR_SIMPLE)) {
        const Seg *segs = B.segs + B.seg_off[r];
        const int ns = (int)(B.seg_off[r + 1] - B.seg_off[r]);
        for (int i = ns - 1; i >= 0 && segs[i].qstart >= lq; i--) {
            const Seg sg = segs[i];
            for (int kk = 0; kk < seg_len(sg); kk++) {
                const uint32_t rec = colrec[rp + sg.roff + kk - B.p0];
                if (!(rec & CR_SKIP) && (rec & CR_KEEP)) keep = 1;
            }
        }
    }
    if (keep && lq > 0) kept[atomicAdd(n_kept, 1)] = (int32_t)r;
    rflags[r] = (uint8_t)((keep ? 1 : 0) | (b15f[r] ? 2 : 0));
}

/* snp:1939: a kept read gets the (capped) pileup copy back; one wave per queued read */
__global__ void __launch_bounds__(256)
k_keepfix(DBatch B, const DParams *__restrict__ P, const int32_t *__restrict__ kept,
          const int *__restrict__ n_kept, uint8_t *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int nw = (int)(gridDim.x * (blockDim.x >> 6));
    const int n = *n_kept, qcap = P->qcap;
    const bool any_pres = P->any_preserve != 0;
    for (int i = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)); i < n; i += nw) {
        const ReadRec rr = B.rr[kept[i]];
        const uint8_t *qin = B.quals + rr.so;
        uint8_t *q2 = out + rr.so;
        for (int x0 = lane; x0 < rr.lq; x0 += 256) {           /* four loads in flight, then four stores */
            uint8_t v4[4];
            #pragma unroll
            for (int u = 0; u < 4; u++) { const int x = x0 + 64 * u; v4[u] = x < rr.lq ? qin[x] : (uint8_t)0; }
            #pragma unroll
            for (int u = 0; u < 4; u++) { const int x = x0 + 64 * u; if (x < rr.lq) q2[x] = cap_q2(P, qcap, any_pres, v4[u]) & 0x7f; }
        }
    }
}

/* ------------------------------------------------------------------------ */
/* indel_only.c:83-122 as a second mode of the masking pass: a base keeps its quality iff it lies
 * within +-D query bases of (qpos + 1) of an entry with indel != 0 (the last column of an op that
 * is followed by I/D) or is_del (every D/N column: qpos is the next base).  One lane per base. */
__global__ void k_indel_only(DBatch B, const int32_t *__restrict__ blk_read, int D, int Q, uint8_t *__restrict__ out)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= B.n_bases) return;
    int64_t r = blk_read[g >> 6];
    while (B.seq_off[r + 1] <= g) r++;
    const ReadRec rr = B.rr[r];
    const uint8_t q0 = B.quals[g];
    if (rr.end <= rr.pos) { out[g] = q0; return; }         /* never reaches a column */
    const int x = (int)(g - rr.so), lq = rr.lq;
    bool keep = false;
    if (!(rr.info & RR_SIMPLE)) {
        const Seg *segs = B.segs + B.seg_off[r];
        const int ns = (int)(B.seg_off[r + 1] - B.seg_off[r]);
        for (int i = 0; i < ns && !keep; i++) {
            const Seg sg = segs[i];
            int c = -1;                                    /* qpos of a marking entry */
            if (seg_type(sg) != SEG_M) c = sg.qstart;
            else if (sg.indel_after != 0) c = sg.qstart + seg_len(sg) - 1;
            if (c < 0) continue;
            int xs = c + 1 - D, xe = c + 1 + D;
            if (xs < 0) xs = 0;
            if (xe >= lq) xe = lq - 1;
            keep = x >= xs && x <= xe;
            /* an M op followed by an indel may ALSO be preceded by D/N columns: handled by those segs */
        }
    }
    out[g] = keep ? q0 : (uint8_t)Q;
}

/* ------------------------------------------------------------------------ */
/* P-block (snp:803-834), one thread per read, in place. */
/* pblock (snp:803-834) as a streaming state machine: the bytes are read 16 at a time (one
 * load per 16 steps instead of a dependent byte load per step); a run's value is only known
 * when it ends, so the run is written back then (the reference's memset). */
__device__ __forceinline__ void pb_fill(uint8_t *qual, int j, int i, int v)
{
    for (int k = j; k < i; k++) qual[k] = (uint8_t)v;
}
__device__ inline void dev_pblock(uint8_t *qual, int len, int level, int qcap, const uint8_t *pq, bool any_pres)
{
    int j = 0, qmin = INT_MAX, qmax = INT_MIN, last_qmin = 0, last_qmax = 0;
    bool skipping = false;                 /* inside the `while (preserve_qual[qual[i]]) i++` of snp:823 */
    level *= 2;
    int i = 0;
    while (i < len) {
        /* fetch up to 16 bytes: bytes up to the next 16-byte boundary, then whole aligned vectors */
        uint32_t w[4] = {0, 0, 0, 0};
        int nbuf;
        const uintptr_t addr = (uintptr_t)(qual + i);
        if ((addr & 15) == 0 && i + 16 <= len) {
            const uint4 v = *(const uint4 *)(qual + i);
            w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; nbuf = 16;
        } else {
            nbuf = 16 - (int)(addr & 15);
            if (nbuf > len - i) nbuf = len - i;
            for (int k = 0; k < nbuf; k++) w[k >> 2] |= (uint32_t)qual[i + k] << (8 * (k & 3));
        }
        for (int k = 0; k < nbuf; k++, i++) {
            const int q = (int)((w[k >> 2] >> (8 * (k & 3))) & 255);
            const bool pres = any_pres && pq[q];
            if (skipping) {
                if (pres) continue;                     /* still skipping preserved values */
                qmin = qmax = q; j = i; skipping = false;
                last_qmin = qmin; last_qmax = qmax;
                continue;
            }
            if (qmin > q) qmin = q;
            if (qmax < q) qmax = q;
            if (qmax - qmin > level || pres) {
                int mid = (last_qmin + last_qmax) / 2;
                if (mid > qcap) mid = qcap;
                pb_fill(qual, j, i, mid);
                if (pres) { skipping = true; j = i; continue; }
                qmin = qmax = q; j = i;
            }
            last_qmin = qmin; last_qmax = qmax;
        }
    }
    if (!skipping) pb_fill(qual, j, len, (last_qmin + last_qmax) / 2);   /* final run: not capped (snp:832) */
}

__global__ void k_pblock(int64_t n_reads, const int64_t *seq_off, uint8_t *out,
                         const uint8_t *rflags, const DParams *P)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    if (rflags && (rflags[r] & 2)) return;            /* flag bit 15: snp:1098 */
    dev_pblock(out + seq_off[r], (int)(seq_off[r + 1] - seq_off[r]), P->pblock, P->qcap, P->preserve_qual,
               P->any_preserve != 0);
}

/* ------------------------------------------------------------------------ */
/* P-block, segment-parallel.  pblock() is a greedy left-to-right segmentation, but some run
 * boundaries do not depend on where the current run started:
 *   - position i with |q[i] - q[i-1]| > 2*level: any run holding both would exceed the range test
 *     of snp:817, so the run holding i-1 ends at i and a new one starts there with qmin = qmax = q[i];
 *   - a preserved value (snp:817,822-823): the run ends, preserved bytes are skipped untouched, and
 *     the first byte after them starts a fresh run;
 *   - the first base of a read.
 * Between two such "stops" the bytes form a segment whose result depends on nothing outside it.
 *   k_pb_marks       the stops of all bases at once (one 64-bit word per 64 bases)
 *   k_pb_readstarts  adds the read starts
 *   k_pb_segments    every segment goes to the lane that owns the 64-base word of its first byte;
 *                    the lane replays the reference's state machine over the segment, 16 bytes per
 *                    load, and only RECORDS the runs: a bit at each run start and the run's value
 *                    (the capped mid-point of its min and max) in a side byte array
 *   k_pb_fill        one lane per 16 bytes: each byte takes the value of the run that holds it
 *                    (coalesced 16-byte loads and stores; no divergent byte loops anywhere)
 * Long reads (a 15 kb HiFi read is ~10^3 segments) no longer serialise on one thread. */
#define PB_KEEP 0xffu            /* run value: leave the bytes as they are (preserved, or flag bit 15) */

__global__ void __launch_bounds__(256)
k_pb_marks(int64_t n_bases, const uint8_t *__restrict__ q, const DParams *__restrict__ P,
           u64 *__restrict__ stop, u64 *__restrict__ rstart, u64 *__restrict__ runs, uint16_t *__restrict__ wmm)
{
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t g0 = w * 64;
    if (g0 >= n_bases) return;
    const int level2 = P->pblock * 2;
    const bool any_pres = P->any_preserve != 0;
    /* the preserved values as (at most four) scalars: a table look-up per byte is a vector load per byte */
    const int npv = P->n_pv;
    const int pv0 = npv > 0 ? (int)P->pv_val[0] : -1, pv1 = npv > 1 ? (int)P->pv_val[1] : -1,
              pv2 = npv > 2 ? (int)P->pv_val[2] : -1, pv3 = npv > 3 ? (int)P->pv_val[3] : -1;
#define PB_IS_PRES(v_) (any_pres && (npv < 0 ? P->preserve_qual[v_] != 0 : ((v_) == pv0 || (v_) == pv1 || (v_) == pv2 || (v_) == pv3)))
    const int nn = (n_bases - g0 < 64) ? (int)(n_bases - g0) : 64;
    int prev = g0 > 0 ? (int)q[g0 - 1] : -1000;
    bool prev_pres = g0 > 0 && PB_IS_PRES(prev);
    u64 m = 0;
    int wmin = 255, wmax = 0;                            /* of the word's bytes: lets a long segment step 64 bases at once */
    for (int k0 = 0; k0 < nn; k0 += 16) {
        uint32_t wd[4] = {0, 0, 0, 0};
        if (k0 + 16 <= nn) { const uint4 v = *(const uint4 *)(q + g0 + k0); wd[0] = v.x; wd[1] = v.y; wd[2] = v.z; wd[3] = v.w; }
        else for (int k = k0; k < nn; k++) wd[(k - k0) >> 2] |= (uint32_t)q[g0 + k] << (8 * (k & 3));
        #pragma unroll
        for (int k = 0; k < 16; k++) {
            const int v = (int)((wd[k >> 2] >> (8 * (k & 3))) & 255);
            const bool pres = PB_IS_PRES(v);
            int d = v - prev; d = d < 0 ? -d : d;
            if (k0 + k < nn && (pres || prev_pres || d > level2)) m |= 1ull << (k0 + k);
            if (k0 + k < nn) { wmin = v < wmin ? v : wmin; wmax = v > wmax ? v : wmax; }
            prev = v; prev_pres = pres;
        }
    }
#undef PB_IS_PRES
    stop[w] = m;
    wmm[w] = (uint16_t)(wmin | (wmax << 8));
    rstart[w] = 0;
    runs[w] = 0;
}

__global__ void k_pb_readstarts(int64_t n_reads, const int64_t *__restrict__ seq_off,
                                u64 *__restrict__ stop, u64 *__restrict__ rstart)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const int64_t g = seq_off[r];
    if (seq_off[r + 1] == g) return;
    if (stop) atomicOr((unsigned long long *)&stop[g >> 6], 1ull << (g & 63));
    if (rstart) atomicOr((unsigned long long *)&rstart[g >> 6], 1ull << (g & 63));
}

__global__ void __launch_bounds__(256)
k_pb_segments(int64_t n_bases, int64_t n_reads, const int64_t *__restrict__ seq_off, const uint8_t *__restrict__ out,
              const uint8_t *__restrict__ rflags, const int32_t *__restrict__ blk_read,
              const u64 *__restrict__ stop, const u64 *__restrict__ rstart, u64 *__restrict__ runs,
              uint8_t *__restrict__ runval, const uint16_t *__restrict__ wmm, uint8_t *__restrict__ dirty,
              const DParams *__restrict__ P)
{
    /* dirty[word] = 1 for the words under a run that changes a byte (not all its bytes equal its value): k_pb_fill
     * skips the others -- after the replay most long-read bases sit in runs of one repeated value */
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nwords = (n_bases + 63) >> 6;
    if (w >= nwords) return;
    u64 starts = stop[w];
    if (!starts) return;
    const int level2 = P->pblock * 2, qcap = P->qcap;
    const bool any_pres = P->any_preserve != 0;
    int64_t r = rflags ? blk_read[w] : 0;
    /* One flat loop: a trip opens the lane's next segment if none is open, consumes up to 16 bytes (one aligned
     * 16-byte load; a whole 64-base word when it lies inside) of the open one, and closes it when that was its
     * end -- a segment of one chunk is one trip, not three (0.3 x config 4: 5.65 -> 4.26 ms; the kernel is bound by
     * VALU issue, 86 % busy); a wave takes as long as its busiest lane has work. */
    bool open = false, at_read_end = false;
    int64_t j = 0, i = 0, e = 0;
    int qmin = 0, qmax = 0, last_qmin = 0, last_qmax = 0;
    for (;;) {
        if (!open) {
            if (!starts) break;
            const int b = __builtin_ctzll(starts);
            starts &= starts - 1;
            const int64_t g = w * 64 + b;
            bool keep = any_pres && P->preserve_qual[out[g]];   /* a stop, but not a start */
            if (!keep && rflags) {
                while (seq_off[r + 1] <= g) r++;
                keep = (rflags[r] & 2) != 0;                    /* flag bit 15: snp:1098 */
            }
            if (keep) { runval[g] = (uint8_t)PB_KEEP; continue; }
            /* fresh state of snp:807 at g; the next stop ends the segment */
            j = g; i = g;
            qmin = INT_MAX; qmax = INT_MIN; last_qmin = last_qmax = 0;
            int64_t ww = w;
            u64 rest = stop[ww] & ~((2ull << b) - 1);
            while (!rest && ++ww < nwords) rest = stop[ww];
            if (ww >= nwords) { e = n_bases; at_read_end = true; }
            else { const int bb = __builtin_ctzll(rest); e = ww * 64 + bb; at_read_end = (rstart[ww] >> bb) & 1; }
            open = true;
        }
        if (i < e) {
            bool adv = false;
            if (!(i & 63) && i + 64 <= e) {                     /* a whole 64-base word lies inside the segment */
                const unsigned mm = wmm[i >> 6];
                const int cmin = (int)(mm & 255), cmax = (int)(mm >> 8);
                const int nmin = cmin < qmin ? cmin : qmin, nmax = cmax > qmax ? cmax : qmax;
                if (nmax - nmin <= level2) {
                    qmin = last_qmin = nmin; qmax = last_qmax = nmax;
                    i += 64;
                    adv = true;
                }
            }
            if (!adv) {
            const int64_t base = i & ~(int64_t)15;
            const uint4 v = *(const uint4 *)(out + base);       /* the buffer has >= 16 bytes of slack */
            const uint32_t wd[4] = {v.x, v.y, v.z, v.w};
            /* fast path: min and max of the chunk's bytes inside [i, e) with packed byte arithmetic
             * (values are < 128, so bit 7 of every byte is free as a borrow guard); if the run's
             * range test still holds with them, no run ends in this chunk */
            {
                const int lo = (int)(i - base), hi = (e - base < 16) ? (int)(e - base) : 16;
                const unsigned vb = ((1u << hi) - 1u) & ~((1u << lo) - 1u);
                uint32_t mn[4], mx[4];
                #pragma unroll
                for (int t = 0; t < 4; t++) {
                    const uint32_t M = ((((vb >> (4 * t)) & 15u) * 0x00204081u) & 0x01010101u) * 0xffu;
                    mn[t] = (wd[t] & M) | (~M & 0x7f7f7f7fu);
                    mx[t] = wd[t] & M;
                }
                #define PB_MIN(a, b, m) (((b) & (m)) | ((a) & ~(m)))       /* m = bytes where a >= b */
                #define PB_MAX(a, b, m) (((a) & (m)) | ((b) & ~(m)))
                uint32_t g01 = PB_GE(mn[0], mn[1]), g23 = PB_GE(mn[2], mn[3]);
                uint32_t a = PB_MIN(mn[0], mn[1], g01), b2 = PB_MIN(mn[2], mn[3], g23);
                uint32_t gab = PB_GE(a, b2);
                const uint32_t m4 = PB_MIN(a, b2, gab);
                g01 = PB_GE(mx[0], mx[1]); g23 = PB_GE(mx[2], mx[3]);
                a = PB_MAX(mx[0], mx[1], g01); b2 = PB_MAX(mx[2], mx[3], g23);
                gab = PB_GE(a, b2);
                const uint32_t x4 = PB_MAX(a, b2, gab);
                #undef PB_MIN
                #undef PB_MAX
                int cmin = (int)(m4 & 255), cmax = (int)(x4 & 255);
                #pragma unroll
                for (int t = 1; t < 4; t++) {
                    const int u0 = (int)((m4 >> (8 * t)) & 255), u1 = (int)((x4 >> (8 * t)) & 255);
                    cmin = u0 < cmin ? u0 : cmin; cmax = u1 > cmax ? u1 : cmax;
                }
                const int nmin = cmin < qmin ? cmin : qmin, nmax = cmax > qmax ? cmax : qmax;
                if (nmax - nmin <= level2) {
                    qmin = last_qmin = nmin; qmax = last_qmax = nmax;
                    i = base + hi;
                    adv = true;
                }
            }
            if (!adv) {
            bool hit = false;
            int hitk = 0, hitq = 0;
            #pragma unroll
            for (int k = 0; k < 16; k++) {
                const int64_t pos = base + k;
                if (!hit && pos >= i && pos < e) {
                    const int q = (int)((wd[k >> 2] >> (8 * (k & 3))) & 255);
                    if (qmin > q) qmin = q;
                    if (qmax < q) qmax = q;
                    if (qmax - qmin > level2) { hit = true; hitk = k; hitq = q; }
                    else { last_qmin = qmin; last_qmax = qmax; }
                }
            }
            if (hit) {                                          /* a run ends inside the segment (snp:817-826) */
                const int64_t pos = base + hitk;
                int mid = (last_qmin + last_qmax) / 2;
                if (mid > qcap) mid = qcap;
                runval[j] = (uint8_t)mid;
                if (last_qmin != last_qmax || mid != last_qmin)
                    for (int64_t dw = j >> 6; dw <= (pos - 1) >> 6; dw++) dirty[dw] = 1;
                atomicOr((unsigned long long *)&runs[pos >> 6], 1ull << (pos & 63));   /* the next run starts at pos */
                qmin = qmax = last_qmin = last_qmax = hitq; j = pos; i = pos + 1;
            } else i = (base + 16 < e) ? base + 16 : e;
            }
            }
        }
        if (i >= e) {
            int mid = (last_qmin + last_qmax) / 2;
            if (!at_read_end && mid > qcap) mid = qcap;         /* the read's final run is not capped (snp:832) */
            runval[j] = (uint8_t)mid;
            if (last_qmin != last_qmax || mid != last_qmin)
                for (int64_t dw = j >> 6; dw <= (e - 1) >> 6; dw++) dirty[dw] = 1;
            open = false;
        }
    }
}

/* ------------------------------------------------------------------------ */
/* P-block, word-parallel (round 3; replaces k_pb_segments and the per-read walk where it is faster -- see DESIGN).
 * k_pb_segments gives a segment to the lane that owns its first byte and pays a loop trip per segment: a 64-byte word
 * with thirty one-base segments around preserved values is thirty trips for one lane while its wave waits, and a
 * 2000-byte run is thirty dependent loads for another.  Here
 *   k_pb_words    a wave takes 64 words (4 KiB, loaded coalesced into LDS).  First the bit masks of every word -- stop,
 *                 preserved, differs-from-its-predecessor, above qcap -- with the wave's lanes as the word's 64 BYTES:
 *                 a compare per mask and word (the wave's condition mask IS the word's bit mask), handed to the lane
 *                 that owns the word.  Then one lane per word: ONE pass of the reference's state machine (snp:807-833)
 *                 over the word's bytes from its first stop on -- every segment and run that starts in the word, up to
 *                 the word's end.  Run starts go to the word's `runs` bits; a run that changes a byte gets its value in
 *                 runval[] and its bit in chgrun[].  A segment still open at the word's end, with no stop at the next
 *                 word's first byte, leaves its state in carry[w] (8 bytes).  mm[w] = (min, max) of the word if it has
 *                 no stop, else of its head (the bytes before its first stop): what k_pb_chains steps over.
 *   k_pb_chains   one lane per carried state: walks on word by word -- a whole word or a head whose (min, max) keeps the
 *                 run inside 2 * level is one step -- to the segment's end; bytes are looked at only where a run breaks.
 *                 The next word's stop mask and (min, max) are in flight while this word is looked at.
 *   k_pb_fillw    a wave per 64 words: the words some changing run touches (dirty[]) are gathered, four lanes each;
 *                 every byte takes the value of the run that holds it.  dirty[w] also tells how far before the word
 *                 the (changing) run that holds its first byte started: no walk back through the masks.
 * The bytes are read once by k_pb_words and then only where something changes. */
#define PBW_ROW 80              /* LDS bytes per word row (64 used; 16-byte aligned, off the 64-byte bank period) */
#define PBW_TILE (64 * PBW_ROW)
#define PBW_CARRY_VALID (1ull << 63)
#define PBW_CARRY(jl, qmin, qmax, lmin, lmax) (PBW_CARRY_VALID | (u64)(jl) | ((u64)(qmin) << 8) | ((u64)(qmax) << 16) | ((u64)(lmin) << 24) | ((u64)(lmax) << 32))
#define PB_DIRTY 0x80000000u    /* dirty[w]: a run that changes a byte touches word w; low 31 bits: how far before the word's
                                 * first byte the changing run that holds it starts (0: it starts here, or changes nothing) */

/* |a - b| of two byte values in one operation; eight wave-uniform 32-bit values handed to lane I (v_writelane_b32).
 * One asm block for the eight, behind two wait states: on gfx940 / gfx950 a VALU instruction must not read an SGPR
 * in the two slots after a VALU instruction (here: v_cmp) wrote it, hipcc's hazard pass does not look into an asm
 * statement's operands, and with one statement per value it scheduled each right behind the compare that made its
 * operand -- a stale mask bit once in ~10^5 words (found by test_pblock_long_read_shapes on the GPU, never on the
 * emulator). */
#ifdef CRUMBLE_EMU
#define PBW_ABSDIFF(a_, b_) ((a_) > (b_) ? (a_) - (b_) : (b_) - (a_))
template <int I> __device__ __forceinline__ void pbw_put8(uint32_t (&m)[8], u64 a, u64 b, u64 c, u64 d)
{
    if ((int)(threadIdx.x & 63) == I) {
        m[0] = (uint32_t)a; m[1] = (uint32_t)(a >> 32); m[2] = (uint32_t)b; m[3] = (uint32_t)(b >> 32);
        m[4] = (uint32_t)c; m[5] = (uint32_t)(c >> 32); m[6] = (uint32_t)d; m[7] = (uint32_t)(d >> 32);
    }
}
#else
#define PBW_ABSDIFF(a_, b_) __builtin_amdgcn_sad_u8((a_), (b_), 0u)
template <int I> __device__ __forceinline__ void pbw_put8(uint32_t (&m)[8], u64 a, u64 b, u64 c, u64 d)
{
    asm("s_nop 1\n\t"
        "v_writelane_b32 %0, %8, %16\n\t"
        "v_writelane_b32 %1, %9, %16\n\t"
        "v_writelane_b32 %2, %10, %16\n\t"
        "v_writelane_b32 %3, %11, %16\n\t"
        "v_writelane_b32 %4, %12, %16\n\t"
        "v_writelane_b32 %5, %13, %16\n\t"
        "v_writelane_b32 %6, %14, %16\n\t"
        "v_writelane_b32 %7, %15, %16"
        : "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]), "+v"(m[6]), "+v"(m[7])
        : "s"(__builtin_amdgcn_readfirstlane((uint32_t)a)), "s"(__builtin_amdgcn_readfirstlane((uint32_t)(a >> 32))),
          "s"(__builtin_amdgcn_readfirstlane((uint32_t)b)), "s"(__builtin_amdgcn_readfirstlane((uint32_t)(b >> 32))),
          "s"(__builtin_amdgcn_readfirstlane((uint32_t)c)), "s"(__builtin_amdgcn_readfirstlane((uint32_t)(c >> 32))),
          "s"(__builtin_amdgcn_readfirstlane((uint32_t)d)), "s"(__builtin_amdgcn_readfirstlane((uint32_t)(d >> 32))),
          "n"(I));
}
#endif

/* k_pb_words, lane = byte: the bit masks of the tile's words I, I + 1, ... 63 (see the kernel), each handed to the lane
 * that owns the word.  m[] = stop, keep, chg, big as (low, high) halves. */
template <int I, bool FULL> __device__ __forceinline__ void
pbw_masks(const uint8_t *cur, const uint8_t *prv, const uint8_t *ptab, int64_t left, unsigned level2, unsigned qcap,
          u64 lastpres, uint32_t (&m)[8])
{
    /* FULL: all 4096 bytes of the tile are data; else bytes from `left` on are not */
    const unsigned v = cur[I * PBW_ROW], pv = prv[I * PBW_ROW];
    const int64_t rem = left - I * 64;
    const u64 okm = (FULL || rem >= 64) ? ~0ull : (rem <= 0 ? 0ull : ((1ull << rem) - 1ull));
    const u64 bk = __ballot(ptab[v] != 0) & okm;
    const u64 bs = (__ballot(PBW_ABSDIFF(v, pv) > level2) & okm) | bk | (bk << 1) | lastpres;
    const u64 bc = __ballot(v != pv), bb = __ballot(v > qcap);
    pbw_put8<I>(m, bs, bk, bc, bb);
    if constexpr (I + 1 < 64) pbw_masks<I + 1, FULL>(cur, prv, ptab, left, level2, qcap, bk >> 63, m);
}

/* one byte q at word position k_ (global g0 + k_) without a stop, inside an open segment: snp:813-829 */
#define PBW_STEP(q_, k_, ON_BREAK) do { \
    const int nmin_ = (q_) < qmin ? (q_) : qmin, nmax_ = (q_) > qmax ? (q_) : qmax; \
    if (nmax_ - nmin_ > level2) { \
        int mid_ = (lmin + lmax) >> 1; \
        mid_ = mid_ > qcap ? qcap : mid_; \
        ON_BREAK(mid_, k_); \
        qmin = qmax = lmin = lmax = (q_); \
    } else { qmin = lmin = nmin_; qmax = lmax = nmax_; } } while (0)

__global__ void __launch_bounds__(256)
k_pb_words(int64_t n_bases, int64_t n_reads, const int64_t *__restrict__ seq_off, const uint8_t *__restrict__ out,
           const uint8_t *__restrict__ rflags, const int32_t *__restrict__ blk_read, const u64 *__restrict__ rstart,
           u64 *__restrict__ stop, u64 *__restrict__ runs, u64 *__restrict__ chgrun, uint8_t *__restrict__ runval,
           u64 *__restrict__ carry, uint16_t *__restrict__ mm, uint32_t *__restrict__ dirty, const DParams *__restrict__ P)
{
    /* per wave: 80 bytes ahead of the tile (the byte before it sits at tile[-17], where lane 0 of row 0 looks for its
     * predecessor), the tile, and the preserved-quality table */
    __shared__ __attribute__((aligned(16))) uint8_t sbytes[4 * (PBW_ROW + PBW_TILE + 256)];
    const int lane = threadIdx.x & 63, wv = UNIFORM((int)(threadIdx.x >> 6));
    uint8_t *const tile = sbytes + wv * (PBW_ROW + PBW_TILE + 256) + PBW_ROW;
    uint8_t *const ptab = tile + PBW_TILE;
    const int64_t nwords = (n_bases + 63) >> 6;
    const int64_t w0 = ((int64_t)blockIdx.x * (blockDim.x >> 6) + wv) * 64;
    if (w0 >= nwords) return;                                   /* (the whole wave) */
    const int64_t gbase = w0 * 64, w = w0 + lane, g0 = w * 64;
    const bool has = w < nwords;
    /* every load the lane may need goes out before the first one is looked at */
    uint4 v4[4];
    #pragma unroll
    for (int t = 0; t < 4; t++) {
        const int64_t off = gbase + t * 1024 + lane * 16;
        v4[t].x = v4[t].y = v4[t].z = v4[t].w = 0;
        if (off < n_bases) v4[t] = *(const uint4 *)(out + off);   /* (d_out carries 64 bytes of slack) */
    }
    const u64 rs = has ? rstart[w] : 0ull, rs1 = (w + 1 < nwords) ? rstart[w + 1] : 0ull;
    int64_t r = (rflags && has) ? blk_read[w] : 0;
    const int pb0 = gbase > 0 ? (int)out[gbase - 1] : 0;                    /* the byte before the tile */
    const int nxb = gbase + 4096 < n_bases ? (int)out[gbase + 4096] : -1;   /* and the one behind it */
    const int level2 = P->pblock * 2, qcap = P->qcap;
    {
        uint32_t pt = 0;
        if (P->any_preserve) pt = *(const uint32_t *)(P->preserve_qual + 4 * lane);
        *(uint32_t *)(ptab + 4 * lane) = pt;
    }
    #pragma unroll
    for (int t = 0; t < 4; t++) *(uint4 *)(tile + (t * 16 + (lane >> 2)) * PBW_ROW + (lane & 3) * 16) = v4[t];
    if (lane == 0) tile[-17] = (uint8_t)pb0;
    WAVE_SYNC();
    /* bit masks over every word's bytes; lane = byte here, a compare per mask and word:
     *   stop  a jump of more than 2 * level from the byte before, a preserved value, or the byte behind one (the read
     *         starts are added from rstart[] below)
     *   keep  the byte's value stays: a preserved quality (snp:817,822) -- bytes of a read with flag bit 15 (snp:1098)
     *         are added below
     *   chg   the byte differs from the byte before it: a segment without such a byte (other than at its start) is
     *         one run of one value -- its bytes do not change unless that value is above qcap
     *   big   a byte above qcap */
    u64 m_stop = 0, keep = 0, chg = 0, big = 0;
    {
        const uint8_t *const cur = tile + lane, *const prv = tile + (lane ? lane - 1 : -17);
        uint32_t m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const u64 pres0 = UNIFORM((int)ptab[pb0]) ? 1ull : 0ull;   /* is the byte before the tile preserved? */
        if (n_bases - gbase >= 4096) pbw_masks<0, true>(cur, prv, ptab, 4096, (unsigned)level2, (unsigned)qcap, pres0, m);
        else pbw_masks<0, false>(cur, prv, ptab, n_bases - gbase, (unsigned)level2, (unsigned)qcap, pres0, m);
        m_stop = ((u64)m[1] << 32) | m[0]; keep = ((u64)m[3] << 32) | m[2];
        chg = ((u64)m[5] << 32) | m[4]; big = ((u64)m[7] << 32) | m[6];
        if (w == 0) { m_stop |= 1ull; chg |= 1ull; }            /* (no byte before the first one) */
    }
#define PBW_IS_PRES(v_) (ptab[v_] != 0)
    /* is the byte behind the word a stop?  (bit 0 of the next word's mask; the tile's last word looks at the byte) */
    bool nst0;
    {
        const u64 nxt = __shfl_down(m_stop, 1);
        nst0 = ((nxt | rs1) & 1ull) != 0;
        if (lane == 63 && nxb >= 0) {
            const int b63 = (int)tile[63 * PBW_ROW + 63];
            int d = nxb - b63; d = d < 0 ? -d : d;
            nst0 = (rs1 & 1ull) || d > level2 || PBW_IS_PRES(nxb) || PBW_IS_PRES(b63);
        }
    }
#undef PBW_IS_PRES
    if (!has) return;
    const int nn = (n_bases - g0 < 64) ? (int)(n_bases - g0) : 64;
    const u64 valid = nn >= 64 ? ~0ull : ((1ull << nn) - 1ull);     /* the end of the data is the last read's end */
    const u64 st = (m_stop | rs) & valid;
    const uint8_t *const byt = tile + lane * PBW_ROW;
    {   /* (min, max) of the word without a stop / of the head before the first stop: the first byte and every byte that
         * differs from the one before it */
        const u64 region = (st ? (st & (0ull - st)) - 1ull : ~0ull) & valid;
        int mn = 255, mx = 0;
        if (region) {
            mn = mx = (int)byt[0];
            u64 c = chg & region & ~1ull;
            while (c) {
                const int q = (int)byt[__builtin_ctzll(c)];
                c &= c - 1;
                mn = q < mn ? q : mn; mx = q > mx ? q : mx;
            }
        }
        mm[w] = (uint16_t)(mn | (mx << 8));                     /* (no byte before a stop at bit 0: 255 | 0 -- joins any run) */
    }
    stop[w] = st;
    if (!st) { runs[w] = 0; chgrun[w] = 0; carry[w] = 0; return; }
    if (rflags) {
        while (r < n_reads && seq_off[r] < g0 + nn) {
            const int64_t a = seq_off[r], b = seq_off[r + 1];
            if (b > g0 && (rflags[r] & 2)) {
                const int lo = a > g0 ? (int)(a - g0) : 0, hi = b < g0 + 64 ? (int)(b - g0) : 64;
                if (lo < hi) keep |= ((hi >= 64 ? ~0ull : ((1ull << hi) - 1ull)) & ~((1ull << lo) - 1ull));
            }
            r++;
        }
    }
    const u64 seg_starts = st & ~keep;                          /* a kept stop ends a segment and starts none */
    /* segments to walk: those with a changing byte inside, all of them if a byte is above qcap (its run is capped) */
    u64 walk = (big & ~keep & valid) ? seg_starts : 0ull;
    {
        u64 c = chg & ~st & valid;                              /* changing bytes that are not themselves stops */
        while (c) {
            const int b = __builtin_ctzll(c);
            const u64 below = st & ((1ull << b) - 1ull);        /* the stops before it: the last one starts its segment */
            const u64 nx = st & ~((2ull << b) - 1ull);          /* the stops behind it: the first one ends the segment */
            const u64 upto = nx ? (nx & (0ull - nx)) - 1ull : ~0ull;
            if (below) {                                        /* (no stop before it: the tail of an earlier word's segment, k_pb_chains) */
                const int sb = 63 - __builtin_clzll(below);
                if ((seg_starts >> sb) & 1ull) walk |= 1ull << sb;
            }
            c &= ~upto;                                         /* the segment's other changing bytes */
        }
    }
    uint8_t *const rowv = runval + g0;
    u64 newruns = 0, chgbits = 0;
    /* a run [j, .) of the word ends with value mid_: recorded only if it changes a byte */
#define PBW_CLOSE(j_, mid_, qmin_, qmax_) do { \
        if (((qmin_) != (qmax_)) | ((mid_) != (qmin_))) { rowv[j_] = (uint8_t)(mid_); chgbits |= 1ull << (j_); } } while (0)
    /* the state the word's last segment is left in (if it is still open at the word's end) */
    const int last_stop = 63 - __builtin_clzll(st);
    int cj = last_stop, cmin = (int)byt[last_stop], cmax = cmin;
    const bool last_open = ((seg_starts >> last_stop) & 1ull) != 0;
    while (walk) {
        const int s0 = __builtin_ctzll(walk);
        walk &= walk - 1;
        const u64 nx = st & ~((2ull << s0) - 1ull);
        const int e = nx ? __builtin_ctzll(nx) : nn;
        int j = s0, qmin = (int)byt[s0], qmax = qmin;
        for (int kd = (s0 + 1) & ~3; kd < e; kd += 4) {         /* four bytes per LDS read */
            const uint32_t d4 = *(const uint32_t *)(byt + kd);
            #pragma unroll
            for (int u = 0; u < 4; u++) {
                const int k = kd + u;
                if (k > s0 && k < e) {
                    const int q = (int)((d4 >> (8 * u)) & 255u);
                    const int nmin = q < qmin ? q : qmin, nmax = q > qmax ? q : qmax;
                    if (nmax - nmin > level2) {                  /* snp:817-826 */
                        int mid = (qmin + qmax) >> 1;
                        mid = mid > qcap ? qcap : mid;
                        PBW_CLOSE(j, mid, qmin, qmax);
                        newruns |= 1ull << k; j = k; qmin = qmax = q;
                    } else { qmin = nmin; qmax = nmax; }
                }
            }
        }
        if (nx || nn < 64) {                                    /* ends inside the word: at a stop, or with the data */
            int mid = (qmin + qmax) >> 1;
            const bool at_read_end = nx ? ((rs >> e) & 1ull) != 0 : true;
            if (!at_read_end && mid > qcap) mid = qcap;         /* a read's final run is not capped (snp:832) */
            PBW_CLOSE(j, mid, qmin, qmax);
        } else { cj = j; cmin = qmin; cmax = qmax; }             /* (the word's last segment) */
    }
    u64 cy = 0;
    if (last_open && nn == 64) {
        const bool last = w + 1 >= nwords;
        if (last || nst0) {                                     /* the next byte is a stop (or there is none): the segment ends here */
            const bool at_read_end = last ? true : (rs1 & 1ull) != 0;
            int mid = (cmin + cmax) >> 1;
            if (!at_read_end && mid > qcap) mid = qcap;
            PBW_CLOSE(cj, mid, cmin, cmax);
        } else cy = PBW_CARRY(cj, cmin, cmax, cmin, cmax);
    }
#undef PBW_CLOSE
    carry[w] = cy;
    runs[w] = newruns;
    chgrun[w] = chgbits;
    if (chgbits) dirty[w] = PB_DIRTY;
}

__global__ void __launch_bounds__(256)
k_pb_chains(int64_t n_bases, const uint8_t *__restrict__ out, const u64 *__restrict__ stop, const u64 *__restrict__ rstart,
            u64 *__restrict__ runs, u64 *__restrict__ chgrun, uint8_t *__restrict__ runval, const u64 *__restrict__ carry,
            const uint16_t *__restrict__ mm, uint32_t *__restrict__ dirty, const DParams *__restrict__ P)
{
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nwords = (n_bases + 63) >> 6;
    if (w >= nwords) return;
    const u64 cy = carry[w];
    if (!(cy & PBW_CARRY_VALID)) return;
    /* (the first word of the walk goes out with the parameters) */
    u64 s_nx = w + 1 < nwords ? stop[w + 1] : 1ull;
    unsigned mm_nx = w + 1 < nwords ? mm[w + 1] : 0u;
    const int level2 = P->pblock * 2, qcap = P->qcap;
    int64_t j = w * 64 + (int64_t)(cy & 63u);
    int qmin = (int)((cy >> 8) & 255u), qmax = (int)((cy >> 16) & 255u), lmin = (int)((cy >> 24) & 255u), lmax = (int)((cy >> 32) & 255u);
    /* a run [j, pos) ends: if it changes a byte, its value, its bit, and the words it touches -- those behind the
     * run's first one learn where the run that holds their first byte starts (k_pb_fillw) */
#define PBC_CLOSE(mid_, pos_) do { \
        if (lmin != lmax || (mid_) != lmin) { \
            runval[j] = (uint8_t)(mid_); \
            atomicOr((unsigned long long *)&chgrun[j >> 6], 1ull << (j & 63)); \
            atomicOr(&dirty[j >> 6], PB_DIRTY); \
            for (int64_t dw_ = (j >> 6) + 1; dw_ <= ((pos_) - 1) >> 6; dw_++) atomicOr(&dirty[dw_], PB_DIRTY | (uint32_t)(dw_ * 64 - j)); \
        } } while (0)
#define PBC_BREAK(mid_, k_) do { const int64_t pos_ = wbase + (k_); PBC_CLOSE(mid_, pos_); \
        atomicOr((unsigned long long *)&runs[pos_ >> 6], 1ull << (pos_ & 63)); j = pos_; } while (0)
    for (int64_t ww = w + 1;; ww++) {
        const bool beyond = ww >= nwords;
        const u64 s = s_nx;
        const unsigned mw = mm_nx;
        if (ww + 1 < nwords) { s_nx = stop[ww + 1]; mm_nx = mm[ww + 1]; } else { s_nx = 1ull; mm_nx = 0u; }
        const int64_t wbase = ww * 64;
        const int wn = beyond ? 0 : ((n_bases - wbase < 64) ? (int)(n_bases - wbase) : 64);
        /* bytes [0, e) of this word belong to the segment: e = the first stop, or all of it */
        const int e = beyond ? 0 : (s ? __builtin_ctzll(s) : wn);
        bool joined = false;
        if (e > 0 && (s || wn == 64)) {                             /* the whole word, or its head, as one step (mm[]) */
            const int cmin = (int)(mw & 255u), cmax = (int)(mw >> 8);
            const int nmin = cmin < qmin ? cmin : qmin, nmax = cmax > qmax ? cmax : qmax;
            if (nmax - nmin <= level2) { qmin = lmin = nmin; qmax = lmax = nmax; joined = true; }
        }
        if (joined && !s) continue;                                 /* the whole word joins the run */
        if (e > 0 && !joined) {
            const uint4 *src = (const uint4 *)(out + wbase);
            #pragma unroll 1
            for (int c = 0; c < 4 && 16 * c < e; c++) {
                const uint4 v = src[c];
                const uint32_t wd[4] = {v.x, v.y, v.z, v.w};
                #pragma unroll
                for (int kk = 0; kk < 16; kk++) {
                    const int k = 16 * c + kk;
                    if (k < e) { const int q = (int)((wd[kk >> 2] >> (8 * (kk & 3))) & 255u); PBW_STEP(q, k, PBC_BREAK); }
                }
            }
        }
        if (beyond || s || wn < 64) {                               /* the segment ends in this word */
            const bool at_read_end = (beyond || !s) ? true : ((rstart[ww] >> e) & 1ull) != 0;
            int mid = (lmin + lmax) >> 1;
            if (!at_read_end && mid > qcap) mid = qcap;
            const int64_t pos = beyond ? n_bases : wbase + e;
            PBC_CLOSE(mid, pos);
            break;
        }
    }
#undef PBC_BREAK
#undef PBC_CLOSE
}
#undef PBW_STEP

/* Short reads without -k/-K.  One lane per read keeps a wave's lanes level, but a lane that walks its
 * own read touches a different cache line than its neighbours at every step (5.9 ms at -9 on the
 * chr20 contig).  So the bytes are read ONCE, coalesced, by k_pb_chunkmm (min and max of every aligned
 * 16-byte chunk); k_pb_reads then walks a read through its chunks' (min, max) pairs -- contiguous,
 * 2 bytes per 16 bases -- and only looks at bytes in the unaligned head and tail of the read and in
 * the rare chunk where a run ends.  Runs are recorded (run-start bit, run value) and the words that
 * hold a run whose bytes actually change are flagged, so that k_pb_fill skips everything else. */
__global__ void __launch_bounds__(256)
k_pb_chunkmm(int64_t n_bases, const uint8_t *__restrict__ out, uint16_t *__restrict__ cmm)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c * 16 >= n_bases) return;
    const uint4 v = *(const uint4 *)(out + c * 16);
    #define PBR_GE(a, b) (((((a) | 0x80808080u) - (b)) & 0x80808080u) >> 7) * 0xffu
    uint32_t g01 = PBR_GE(v.x, v.y), g23 = PBR_GE(v.z, v.w);
    const uint32_t mn01 = (v.y & g01) | (v.x & ~g01), mn23 = (v.w & g23) | (v.z & ~g23);
    const uint32_t mx01 = (v.x & g01) | (v.y & ~g01), mx23 = (v.z & g23) | (v.w & ~g23);
    const uint32_t gmn = PBR_GE(mn01, mn23), gmx = PBR_GE(mx01, mx23);
    const uint32_t m4 = (mn23 & gmn) | (mn01 & ~gmn), x4 = (mx01 & gmx) | (mx23 & ~gmx);
    #undef PBR_GE
    unsigned cmin = m4 & 255, cmax = x4 & 255;
    #pragma unroll
    for (int t = 1; t < 4; t++) {
        const unsigned u0 = (m4 >> (8 * t)) & 255, u1 = (x4 >> (8 * t)) & 255;
        cmin = u0 < cmin ? u0 : cmin; cmax = u1 > cmax ? u1 : cmax;
    }
    cmm[c] = (uint16_t)(cmin | (cmax << 8));
}

__global__ void __launch_bounds__(256)
k_pb_reads(int64_t n_reads, const int64_t *__restrict__ seq_off, const uint8_t *__restrict__ out,
           const uint8_t *__restrict__ rflags, const uint16_t *__restrict__ cmm, u64 *__restrict__ runs,
           uint8_t *__restrict__ runval, uint8_t *__restrict__ dirty, const DParams *__restrict__ P)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const int64_t so = seq_off[r], e = seq_off[r + 1];
    if (e == so) return;
    if (rflags && (rflags[r] & 2)) { runval[so] = (uint8_t)PB_KEEP; return; }   /* flag bit 15: snp:1098 */
    const int level2 = P->pblock * 2, qcap = P->qcap;
    int64_t j = so, i = so;
    int qmin = INT_MAX, qmax = INT_MIN, last_qmin = 0, last_qmax = 0;
    bool anydirty = false;
    while (i < e) {
        if (!(i & 15) && i + 16 <= e) {                         /* a whole aligned chunk of this read */
            const unsigned mm = cmm[i >> 4];
            const int cmin = (int)(mm & 255), cmax = (int)(mm >> 8);
            const int nmin = cmin < qmin ? cmin : qmin, nmax = cmax > qmax ? cmax : qmax;
            if (nmax - nmin <= level2) { qmin = last_qmin = nmin; qmax = last_qmax = nmax; i += 16; continue; }
        }
        /* the read's unaligned head or tail, or a chunk in which a run ends: byte by byte, every run end
         * of the chunk in the same pass (quantised data changes level every few dozen bases) */
        const int64_t base = i & ~(int64_t)15;
        const uint4 v = *(const uint4 *)(out + base);           /* the buffer has >= 16 bytes of slack */
        const uint32_t wd[4] = {v.x, v.y, v.z, v.w};
        const int lo = (int)(i - base), hi = (e - base < 16) ? (int)(e - base) : 16;
        const unsigned vb = ((1u << hi) - 1u) & ~((1u << lo) - 1u);
        unsigned newruns = 0;
        #pragma unroll
        for (int k = 0; k < 16; k++) {
            if ((vb >> k) & 1u) {
                const int q = (int)((wd[k >> 2] >> (8 * (k & 3))) & 255);
                qmin = q < qmin ? q : qmin;
                qmax = q > qmax ? q : qmax;
                if (qmax - qmin > level2) {                  /* a run ends here (snp:817-826) */
                    int mid = (last_qmin + last_qmax) >> 1;
                    mid = mid > qcap ? qcap : mid;
                    runval[j] = (uint8_t)mid;
                    anydirty |= (last_qmin != last_qmax) | (mid != last_qmin);
                    newruns |= 1u << k;
                    qmin = qmax = q; j = base + k;
                }
                last_qmin = qmin; last_qmax = qmax;
            }
        }
        if (newruns) atomicOr((unsigned long long *)&runs[base >> 6], (unsigned long long)newruns << (base & 63));
        i = base + hi;
    }
    const int mid = (last_qmin + last_qmax) / 2;                 /* the final run is not capped (snp:832) */
    runval[j] = (uint8_t)mid;
    anydirty |= (last_qmin != last_qmax) | (mid != last_qmin);
    if (anydirty)                                                /* some run of this read changes bytes: its words go through k_pb_fill */
        for (int64_t w = so >> 6; w <= (e - 1) >> 6; w++) dirty[w] = 1;
}

/* every byte takes the value of the run that holds it */
__global__ void __launch_bounds__(256)
k_pb_fill(int64_t n_bases, uint8_t *__restrict__ out, const u64 *__restrict__ stop, const u64 *__restrict__ runs,
          const uint8_t *__restrict__ runval, const uint8_t *__restrict__ dirty)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t g0 = t * 16;
    if (g0 >= n_bases) return;
    const int64_t w = g0 >> 6;
    if (dirty && !dirty[w]) return;                       /* no run that changes a byte touches this word */
    const int sh = (int)(g0 & 63);
    const u64 m = stop[w] | runs[w];
    /* run start governing the first byte: the last set bit at or before g0 */
    int64_t s;
    {
        u64 below = m & ((2ull << sh) - 1);
        int64_t ww = w;
        while (!below && ww > 0) { ww--; below = stop[ww] | runs[ww]; }
        s = below ? ww * 64 + (63 - __builtin_clzll(below)) : 0;
    }
    uint32_t val = runval[s];
    const uint4 v = *(const uint4 *)(out + g0);
    if (!((m >> sh) & 0xfffeu) && g0 + 16 <= n_bases) {
        /* no run starts inside the chunk (bit 0 would be its own start, already in val): one value */
        if (val == PB_KEEP) return;
        const uint32_t vv = val * 0x01010101u;
        if (v.x != vv || v.y != vv || v.z != vv || v.w != vv) { uint4 o; o.x = o.y = o.z = o.w = vv; *(uint4 *)(out + g0) = o; }
        return;
    }
    const uint4 rv4 = *(const uint4 *)(runval + g0);        /* only the bytes at run starts are meaningful */
    uint32_t wd[4] = {v.x, v.y, v.z, v.w};
    const uint32_t rv[4] = {rv4.x, rv4.y, rv4.z, rv4.w};
    const unsigned bits = (unsigned)((m >> sh) & 0xffffu);
    bool changed = false;
    #pragma unroll
    for (int k = 0; k < 16; k++) {
        if (k > 0 && ((bits >> k) & 1)) val = (rv[k >> 2] >> (8 * (k & 3))) & 255;
        const uint32_t old = (wd[k >> 2] >> (8 * (k & 3))) & 255;
        if (val != PB_KEEP && old != val && g0 + k < n_bases) {
            wd[k >> 2] = (wd[k >> 2] & ~(255u << (8 * (k & 3)))) | (val << (8 * (k & 3)));
            changed = true;
        }
    }
    if (changed) {
        if (g0 + 16 <= n_bases) { uint4 o; o.x = wd[0]; o.y = wd[1]; o.z = wd[2]; o.w = wd[3]; *(uint4 *)(out + g0) = o; }
        else for (int k = 0; g0 + k < n_bases; k++) out[g0 + k] = (uint8_t)((wd[k >> 2] >> (8 * (k & 3))) & 255);
    }
}

/* the word-parallel P-block's fill: a wave looks at the dirty[] entries of 64 words, gathers the words some changing run
 * touches and gives each to four lanes (16 bytes per lane): a byte takes the value of the run that holds it if that
 * run's chgrun bit is set.  The run that holds a chunk's first byte starts at the last run start at or before it in
 * the word, else dirty[w] tells how far before the word (0: that run changes nothing). */
__global__ void __launch_bounds__(256)
k_pb_fillw(int64_t n_bases, uint8_t *__restrict__ out, const u64 *__restrict__ stop, const u64 *__restrict__ runs,
           const u64 *__restrict__ chgrun, const uint8_t *__restrict__ runval, const uint32_t *__restrict__ dirty)
{
    __shared__ uint8_t slots[4 * 64];
    const int lane = threadIdx.x & 63;
    uint8_t *const slot = slots + (threadIdx.x >> 6) * 64;
    const int64_t nwords = (n_bases + 63) >> 6;
    const int64_t w0 = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 64;
    if (w0 >= nwords) return;                                   /* (the whole wave) */
    const uint32_t dz = w0 + lane < nwords ? dirty[w0 + lane] : 0u;
    const u64 dm = __ballot(dz != 0u);
    if (!dm) return;
    const int nd = __popcll(dm);
    if (dz) slot[__popcll(dm & ((1ull << lane) - 1ull))] = (uint8_t)lane;
    WAVE_SYNC();
    for (int p = 0; p * 16 < nd; p++) {
        const int kq = p * 16 + (lane >> 2);
        if (kq >= nd) continue;
        const int64_t w = w0 + (int64_t)slot[kq];
        const int sh = (lane & 3) * 16;
        const int64_t g0 = w * 64 + sh;
        if (g0 >= n_bases) continue;
        const u64 m = stop[w] | runs[w], cr = chgrun[w];
        const uint32_t dist = dirty[w] & ~PB_DIRTY;
        const uint4 v = *(const uint4 *)(out + g0);
        const uint4 rv4 = *(const uint4 *)(runval + g0);        /* only the bytes at starts of changing runs are meaningful */
        /* the run that holds the chunk's first byte */
        uint32_t val = PB_KEEP;
        {
            const u64 below = m & ((2ull << sh) - 1ull);
            if (below) {
                const int sb = 63 - __builtin_clzll(below);
                if ((cr >> sb) & 1ull) val = runval[w * 64 + sb];
            } else if (dist) val = runval[w * 64 - (int64_t)dist];
        }
        const unsigned bits = (unsigned)((m >> sh) & 0xffffu), cbits = (unsigned)((cr >> sh) & 0xffffu);
        if (!(bits & 0xfffeu) && g0 + 16 <= n_bases) {
            /* no run starts inside the chunk (bit 0 would be its own start, already in val): one value */
            if (val == PB_KEEP) continue;
            const uint32_t vv = val * 0x01010101u;
            if (v.x != vv || v.y != vv || v.z != vv || v.w != vv) { uint4 o; o.x = o.y = o.z = o.w = vv; *(uint4 *)(out + g0) = o; }
            continue;
        }
        uint32_t wd[4] = {v.x, v.y, v.z, v.w};
        const uint32_t rv[4] = {rv4.x, rv4.y, rv4.z, rv4.w};
        bool changed = false;
        #pragma unroll
        for (int k = 0; k < 16; k++) {
            if (k > 0 && ((bits >> k) & 1u)) val = ((cbits >> k) & 1u) ? ((rv[k >> 2] >> (8 * (k & 3))) & 255u) : (uint32_t)PB_KEEP;
            const uint32_t old = (wd[k >> 2] >> (8 * (k & 3))) & 255u;
            if (val != PB_KEEP && old != val && g0 + k < n_bases) {
                wd[k >> 2] = (wd[k >> 2] & ~(255u << (8 * (k & 3)))) | (val << (8 * (k & 3)));
                changed = true;
            }
        }
        if (changed) {
            if (g0 + 16 <= n_bases) { uint4 o; o.x = wd[0]; o.y = wd[1]; o.z = wd[2]; o.w = wd[3]; *(uint4 *)(out + g0) = o; }
            else for (int k = 0; g0 + k < n_bases; k++) out[g0 + k] = (uint8_t)((wd[k >> 2] >> (8 * (k & 3))) & 255u);
        }
    }
}

/* ------------------------------------------------------------------------ */
/* Output-quality histogram + changed-byte count (feeds the RCCL all-reduce).
 * 16 bytes per lane per load; the LDS histogram is replicated 32x with the copy
 * index in the bank bits, so the dominant output values (qhigh, qlow) do not
 * serialise the LDS atomics. */
#define HIST_COPIES 32
__global__ void __launch_bounds__(256)
k_stats(int64_t b_lo, int64_t b_hi, int64_t n_reads, const uint8_t *out, const uint8_t *in, u64 *stats, int *bad_qual)
{
    /* bytes [b_lo, b_hi) of the batch (a window of a contig counts the reads it owns): the unaligned head is counted
     * here by workgroup 0, the rest as 16-byte vectors from the next aligned offset */
    const int64_t a_lo = ((b_lo + 15) & ~(int64_t)15) < b_hi ? ((b_lo + 15) & ~(int64_t)15) : b_hi;
    const int64_t head_lo = b_lo, head_hi = a_lo, bases_total = b_hi - b_lo;
    out += a_lo; in += a_lo;
    const int64_t n_bases = b_hi - a_lo;
    __shared__ unsigned int h[256 * HIST_COPIES];
    __shared__ unsigned int hchg;
    for (int i = threadIdx.x; i < 256 * HIST_COPIES; i += blockDim.x) h[i] = 0;
    if (threadIdx.x == 0) hchg = 0;
    __syncthreads();
    const int copy = threadIdx.x & (HIST_COPIES - 1);
    const int64_t nvec = n_bases >> 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned int changed = 0, hibits = 0;                  /* hibits: an INPUT quality >= 128: this is the pass that
                                                            * looks at every input quality, so the input validation of the qualities is here */
    const uint4 *o4 = (const uint4 *)out, *i4 = (const uint4 *)in;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        const uint4 a = o4[i], b = i4[i];
        hibits |= b.x | b.y | b.z | b.w;
        const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
        #pragma unroll
        for (int k = 0; k < 4; k++) {
            #pragma unroll
            for (int j = 0; j < 4; j++) {
                const unsigned o = (aw[k] >> (8 * j)) & 255, q = (bw[k] >> (8 * j)) & 255;
                atomicAdd(&h[o * HIST_COPIES + copy], 1u);
                changed += (o != q);
            }
        }
    }
    /* head and tail bytes */
    if (blockIdx.x == 0) {
        for (int64_t i = (nvec << 4) + threadIdx.x; i < n_bases; i += blockDim.x) {
            const unsigned o = out[i];
            atomicAdd(&h[o * HIST_COPIES + copy], 1u);
            changed += (o != in[i]);
            hibits |= in[i];
        }
        for (int64_t i = head_lo - a_lo + threadIdx.x; i < head_hi - a_lo; i += blockDim.x) {   /* (negative offsets) */
            const unsigned o = out[i];
            atomicAdd(&h[o * HIST_COPIES + copy], 1u);
            changed += (o != in[i]);
            hibits |= in[i];
        }
    }
    if (hibits & 0x80808080u) *bad_qual = 1;
    if (changed) atomicAdd(&hchg, changed);
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += blockDim.x) {
        unsigned int v = 0;
        for (int c = 0; c < HIST_COPIES; c++) v += h[i * HIST_COPIES + c];
        if (v) atomicAdd(&stats[ST_HIST0 + i], (u64)v);
    }
    if (threadIdx.x == 0) {
        if (hchg) atomicAdd(&stats[ST_BASES_CHANGED], (u64)hchg);
        if (blockIdx.x == 0) {
            atomicAdd(&stats[ST_READS], (u64)n_reads);
            atomicAdd(&stats[ST_BASES], (u64)bases_total);
        }
    }
}

/* ------------------------------------------------------------------------ */
/* The result in coded form (crumble_hip_encode; include/crumble_hip.h "coded download"): what crosses PCIe on the way
 * out.  After the pass almost every quality byte is one of two values (qhigh / qlow at the numeric presets: the two
 * fullest bins among the bytes the pass changed are taken, k_encode_pick) or the read's own input quality after cap_quality
 * (snp:1325-1332: kept reads, STR windows, back-fill ranges, preserved values) -- which the host still holds.  So the
 * device hands back 2 bits per base, 16 bases to a 32-bit word (base i in bits 2*(i & 15)):
 *   0 = value[0], 1 = value[1], 2 = cap[input quality], 3 = the next byte of the escape stream
 * plus the escape bytes themselves (P-block averages, bin2[] levels that are neither of the two values), in base order
 * inside every 4096-base block; esc_off[block] is where the block's escapes start (blocks reserve their slice with
 * ONE atomic each, and only blocks that have escapes do).  The host expands it while it writes the records
 * (bamio.c) -- a quarter of the bytes over the link, the same bytes in the file. */
#define ENC_BLOCK 4096
/* the two values of codes 0 and 1: the two fullest bins among the bytes the pass CHANGED (a byte that still equals its
 * capped input is code 2 whatever its value), from a sample of 16-byte chunks spread over the batch */
__global__ void __launch_bounds__(256)
k_encode_pick(int64_t nb, const uint8_t *__restrict__ out, const uint8_t *__restrict__ in, const DParams *__restrict__ P,
              uint8_t *__restrict__ pick, unsigned long long *__restrict__ esc_total)
{
    __shared__ unsigned hist[128];
    __shared__ uint8_t cap[256];
    if (threadIdx.x < 128) hist[threadIdx.x] = 0;
    {
        const int qcap = P->qcap, i = threadIdx.x;
        cap[i] = (uint8_t)((i > qcap && !P->preserve_qual[i]) ? qcap : i);
    }
    __syncthreads();
    const int64_t nch = nb >> 4;
    const int64_t want = 8192, step = nch > want ? nch / want : 1;
    for (int64_t k = threadIdx.x; k * step < nch && k < want; k += blockDim.x) {
        const int64_t b0 = (k * step) << 4;
        const uint4 o4 = *(const uint4 *)(out + b0), q4 = *(const uint4 *)(in + b0);
        const uint32_t ow[4] = {o4.x, o4.y, o4.z, o4.w}, qw[4] = {q4.x, q4.y, q4.z, q4.w};
        #pragma unroll
        for (int j = 0; j < 16; j++) {
            const unsigned o = (ow[j >> 2] >> (8 * (j & 3))) & 255, q = (qw[j >> 2] >> (8 * (j & 3))) & 255;
            if (o != cap[q] && o < 128) atomicAdd(&hist[o], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x) return;
    int a = -1, b = -1;
    unsigned ca = 0, cb = 0;
    for (int i = 0; i < 128; i++) {
        const unsigned c = hist[i];
        if (c > ca) { b = a; cb = ca; a = i; ca = c; }
        else if (c > cb) { b = i; cb = c; }
    }
    if (a < 0) a = P->qhigh;
    if (b < 0) b = (P->qlow != a) ? P->qlow : ((a + 1) & 127);
    pick[0] = (uint8_t)a; pick[1] = (uint8_t)b;
    *esc_total = 0;
}

__global__ void __launch_bounds__(256)
k_encode(int64_t nb, const uint8_t *__restrict__ out, const uint8_t *__restrict__ in, const DParams *__restrict__ P,
         const uint8_t *__restrict__ pick, uint32_t *__restrict__ codes, int64_t *__restrict__ esc_off,
         uint8_t *__restrict__ esc, unsigned long long *__restrict__ esc_total)
{
    __shared__ uint8_t cap[256];
    __shared__ unsigned wtot[4];
    __shared__ unsigned long long base_s;
    {
        const int qcap = P->qcap;
        const int i = threadIdx.x;
        cap[i] = (uint8_t)((i > qcap && !P->preserve_qual[i]) ? qcap : i);      /* cap_quality snp:1325-1332 */
    }
    const unsigned va = pick[0], vb = pick[1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t nblk = (nb + ENC_BLOCK - 1) / ENC_BLOCK;
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        __syncthreads();                                    /* cap[] filled; wtot / base_s of the last round read */
        const int64_t b0 = blk * ENC_BLOCK + (int64_t)threadIdx.x * 16;
        uint32_t code = 0, emask = 0;
        uint32_t ow[4] = {0, 0, 0, 0};
        if (b0 < nb) {
            /* (both arrays carry at least 64 bytes of slack behind their last byte; bytes beyond nb are masked out) */
            const uint4 o4 = *(const uint4 *)(out + b0), q4 = *(const uint4 *)(in + b0);
            ow[0] = o4.x; ow[1] = o4.y; ow[2] = o4.z; ow[3] = o4.w;
            const uint32_t qw[4] = {q4.x, q4.y, q4.z, q4.w};
            const int nv = nb - b0 < 16 ? (int)(nb - b0) : 16;
            #pragma unroll
            for (int k = 0; k < 16; k++) {
                const unsigned o = (ow[k >> 2] >> (8 * (k & 3))) & 255, q = (qw[k >> 2] >> (8 * (k & 3))) & 255;
                unsigned c = o == va ? 0u : o == vb ? 1u : o == cap[q] ? 2u : 3u;
                if (k >= nv) c = 0;
                code |= c << (2 * k);
                emask |= (c == 3u ? 1u : 0u) << k;
            }
        }
        const unsigned ne = (unsigned)__popc(emask);
        unsigned incl = ne;                                 /* inclusive prefix of the escape counts over the wave */
        #pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const unsigned t = __shfl_up(incl, d); if (lane >= d) incl += t; }
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        unsigned before = incl - ne, total = 0;
        #pragma unroll
        for (int w = 0; w < 4; w++) { if (w < wave) before += wtot[w]; total += wtot[w]; }
        if (threadIdx.x == 0) {
            const unsigned long long base = total ? atomicAdd(esc_total, (unsigned long long)total) : 0ull;
            base_s = base;
            esc_off[blk] = (int64_t)base;
        }
        if (b0 < nb) codes[b0 >> 4] = code;
        if (total) {                                        /* (uniform over the workgroup) */
            __syncthreads();
            uint8_t *e = esc + base_s + before;
            #pragma unroll
            for (int k = 0; k < 16; k++)
                if ((emask >> k) & 1u) *e++ = (uint8_t)((ow[k >> 2] >> (8 * (k & 3))) & 255);
        }
    }
}

/* reads / bases of a batch whose histogram the replay kernels took */
__global__ void k_stats_totals(int64_t n_reads, int64_t n_bases, u64 *stats)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) { atomicAdd(&stats[ST_READS], (u64)n_reads); atomicAdd(&stats[ST_BASES], (u64)n_bases); }
}

/* this batch's counters into the engine's accumulated vector (what the multi-GPU all-reduce sums) */
/* (a run that ends in an error adds nothing: the partial counters of a batch that is rejected -- CRUMBLE_EOVERFLOW, an
 * input quality >= 128 seen by the replay / k_stats (overflow[4]), the expansion's late errors: unsorted input, tile
 * span, hit lists (p0_err, hit_err) -- must not reach the all-reduced vector) */
__global__ void k_stats_fold(u64 *__restrict__ total, const u64 *__restrict__ batch, int n, const int *__restrict__ overflow,
                             const int *__restrict__ spec_abort, const int *__restrict__ p0_err, const int *__restrict__ hit_err)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && !overflow[0] && !overflow[4] && !*p0_err && !*hit_err && !LIST_ABORTED(spec_abort))
        total[i] += batch[i];                                                        /* (an aborted fast run is repeated) */
}

#endif /* CRUMBLE_KERNELS_H */
