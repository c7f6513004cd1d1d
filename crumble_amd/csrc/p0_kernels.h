/*
 * p0_kernels.h -- "P0 expand" on the device: everything the column engine needs beyond the raw
 * BAM-shaped arrays of a crumble_batch is derived here, so crumble_hip_upload() is a set of
 * host->device copies plus a few launches (no per-read host loops).  What htslib's bam_plp_auto and
 * pileup_callback (snp_score.c:1113-1153) do per read, done once per batch:
 *
 *   k_p0_classify   per read: pileup_callback's routing (snp_score.c:1130-1149) + htslib's default
 *                   flag filter, reference length / end, number of reference-consuming CIGAR ops,
 *                   input validation (the reference asserts instead: snp_score.c:1075); per 1024-read
 *                   block: the sums of the counts and the largest start / end (the scan's partials)
 *   k_p0_mid        one workgroup: exclusive scan of the block partials (offsets, prefix maxima), totals
 *   k_p0_fill       the block's own scan, then per read (a lane each, or a wave each for long reads):
 *                   Seg + SegRec tables with htslib's resolve_cigar2 `indel` peek,
 *                   ReadRec, the monotone search keys, coordinate order of the reads that enter the pileup
 *                   (snp_score.c:1075), the list of bases that D/N columns report (replay), the list of
 *                   general-CIGAR reads, block -> read table, territories of long reads (VRec)
 *   k_p0_cix        base -> column table of the general-CIGAR reads of a short-read batch (a wave per read)
 *   k_p0_tiles      candidate ranges per 64-column tile (reads and ops), keep-BED cursor per tile
 *   k_p0_bases      the bases as the kernels read them (one byte per base, consensus class in the high
 *                   nibble) from BAM's packed nibbles, staged through LDS; chunk -> read table
 *   k_p0_rtcount .. k_p0_hitfill   long reads: the records that reach each tile
 * (input qualities >= 128, SURVEY.md Q13, are found by the run's last pass over the qualities, k_stats.)
 */
#ifndef CRUMBLE_P0_KERNELS_H
#define CRUMBLE_P0_KERNELS_H

#include "dev_common.h"

/* per-read counts of the classify pass (16 bytes: one vector load in the fill pass) */
struct P0Cnt32 { int32_t nseg, ncixu /* base -> column entries, in units of CIX_UNIT */, nslow, pad; };
/* per 1024-read block: the block's sums and maxima from k_p0_classify, turned by k_p0_mid into what lies before the block
 * (exclusive sums, prefix maxima); entry [nblocks] holds the totals */
struct P0Part { int64_t nseg, ncixu, nslow, ngen /* general-CIGAR reads with bases */; int32_t maxpos, maxend;
                int32_t minpos /* smallest start of a read in the pileup: with the totals' maxend, the batch's extent */, pad_; };
#define P0_CHUNK 1024

#define P0_ERR_NEG      1
#define P0_ERR_QLEN     2
#define P0_ERR_COORD    3
#define P0_ERR_UNSORTED 4
#define P0_ERR_QUAL     5
#define P0_ERR_SPAN     6       /* the reads over one 64-column tile span more than 3.7 GB of bases (k_columns2's 32-bit offsets) */
#define P0_ERR_HITS     7       /* internal: tile hit lists inconsistent */
#define P0_ERR_TOTALS   8       /* n_bases / n_cigar_ops differ from the sums of the per-read lengths (the device-scan path: the host path checks them before the copy) */

__device__ __forceinline__ bool p0_is_mdnx(int op) { return op == 0 || op == 2 || op == 3 || op == 7 || op == 8; }
__device__ __forceinline__ bool p0_is_query(int op) { return op == 0 || op == 1 || op == 4 || op == 7 || op == 8; }

/* bam_pileup1_t.indel on the last column of ref-op k (htslib resolve_cigar2's peek at the
 * next op) */
__device__ __forceinline__ int p0_indel_after(const uint32_t *cig, int n, int k)
{
    if (k + 1 >= n) return 0;
    const int op = cig[k] & 15, op2 = cig[k + 1] & 15, l2 = (int)(cig[k + 1] >> 4);
    int indel = 0;
    if (op2 == 2 && op != 2) {
        indel = -l2;
        for (int j = k + 2; j < n; j++) { if ((cig[j] & 15) == 2) indel -= (int)(cig[j] >> 4); else break; }
    } else if (op2 == 1) {
        indel = l2;
        for (int j = k + 2; j < n; j++) {
            const int o = cig[j] & 15;
            if (o == 1) indel += (int)(cig[j] >> 4); else if (o != 6) break;
        }
    } else if (op2 == 6 && k + 2 < n) {
        int l3 = 0;
        for (int j = k + 2; j < n; j++) {
            const int o = cig[j] & 15;
            if (o == 1) l3 += (int)(cig[j] >> 4);
            else if (o == 2 || o == 0 || o == 3 || o == 7 || o == 8) break;
        }
        if (l3 > 0) indel = l3;
    }
    return indel;
}

/* one read: routing, validation, counts.  mn / mx: its span when it enters the pileup */
__device__ __forceinline__ void p0_classify_read(int64_t r, int tid, const int32_t *__restrict__ pos, const uint16_t *__restrict__ flag,
                              const int64_t *__restrict__ seq_off, const int64_t *__restrict__ cig_off,
                              const uint32_t *__restrict__ cigar, uint8_t *__restrict__ inplp,
                              int32_t *__restrict__ end, P0Cnt32 &c, int *err, int &mn, int &mx,
                              int64_t nb, int64_t nc, int64_t n)
{
    const int64_t co0 = cig_off[r], co1 = cig_off[r + 1], so0 = seq_off[r], so1 = seq_off[r + 1];
    const int64_t ncg64 = co1 - co0, lq = so1 - so0;
    inplp[r] = 0; end[r] = pos[r];
    /* offsets must run 0 <= off[r] <= off[r+1] <= total before anything is dereferenced through them */
    if (lq < 0 || ncg64 < 0 || lq > 0x3fffffff || ncg64 > 0x3fffffff || co0 < 0 || co1 > nc || so0 < 0 || so1 > nb ||
        (r == 0 && (co0 != 0 || so0 != 0))) { atomicMax(err, P0_ERR_NEG); return; }
    /* the caller's totals size every buffer: an overstated one would have the host and the kernels read past the arrays */
    if (r == n - 1 && (co1 != nc || so1 != nb)) { atomicMax(err, P0_ERR_TOTALS); return; }
    const uint32_t *cig = cigar + co0;
    const int ncg = (int)ncg64;
    /* pileup_callback routing (snp:1130-1149) + htslib's default filter */
    if (tid < 0 || (flag[r] & (4 | 256 | 512 | 1024))) return;
    int64_t x = 0, y = 0;
    int nseg = 0, ndn = 0, first_type = -1, first_ia = 0;
    for (int k = 0; k < ncg; k++) {
        const int op = cig[k] & 15, l = (int)(cig[k] >> 4);
        if (p0_is_mdnx(op)) {
            if (l > 0) {
                if (!nseg) { first_type = (op == 2) ? 1 : (op == 3) ? 2 : 0; first_ia = p0_indel_after(cig, ncg, k); }
                nseg++;
                if ((op == 2 || op == 3) && y < lq) ndn++;                 /* (an upper bound: runs of D/N ops report one base) */
            }
            x += l;
            if (op != 2 && op != 3) y += l;
        } else if (op == 1 || op == 4) y += l;
    }
    if (!nseg || x <= 0) return;                                       /* no reference-consuming op */
    if (y != lq && lq != 0) { atomicMax(err, P0_ERR_QLEN); return; }
    if (pos[r] < 0 || (int64_t)pos[r] + x >= (int64_t)INT_MAX - 64) { atomicMax(err, P0_ERR_COORD); return; }
    inplp[r] = 1;
    end[r] = pos[r] + (int32_t)x;
    const bool simple = nseg == 1 && first_type == 0 && first_ia == 0 && lq > 0;
    c.nseg = nseg; c.ncixu = simple ? 0 : (int32_t)((lq + CIX_UNIT - 1) / CIX_UNIT); c.nslow = simple ? 0 : ndn;
    mn = pos[r]; mx = end[r];                              /* extent of the pileup */
}

/* 1024 reads per workgroup, a lane each; the workgroup's sums / maxima are the partials of the scan that k_p0_fill
 * finishes (one atomic pair per workgroup for the batch's extent: an atomic pair per read, 12.6 M on two words, was
 * most of this kernel's time once) */
__global__ void __launch_bounds__(1024)
k_p0_classify(int64_t n, int tid, const int32_t *__restrict__ pos, const uint16_t *__restrict__ flag,
              const int64_t *__restrict__ seq_off, const int64_t *__restrict__ cig_off,
              const uint32_t *__restrict__ cigar, uint8_t *__restrict__ inplp,
              int32_t *__restrict__ end, P0Cnt32 *__restrict__ cnt, P0Part *__restrict__ part, int *err,
              int64_t nb, int64_t nc)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int mn = INT_MAX, mx = INT_MIN;
    P0Cnt32 c = {0, 0, 0, 0};
    if (r < n) {
        p0_classify_read(r, tid, pos, flag, seq_off, cig_off, cigar, inplp, end, c, err, mn, mx, nb, nc, n);
        cnt[r] = c;
    }
    __shared__ int wmn[16], wmx[16], wpk[16], wns[16], wsl[16], wgn[16];
    __shared__ long long wcx[16];
    int pk = mn == INT_MAX ? INT_MIN : mn;                 /* start of a read in the pileup (the search key's source) */
    int ns = c.nseg, sl = c.nslow;
    long long cx = c.ncixu;
    const int gn = __popcll(__ballot(c.ncixu > 0));        /* general-CIGAR reads with bases: the ones with table entries */
    #pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const int a = __shfl_xor(mn, d), b = __shfl_xor(mx, d), k2 = __shfl_xor(pk, d);
        mn = a < mn ? a : mn; mx = b > mx ? b : mx; pk = k2 > pk ? k2 : pk;
        ns += __shfl_xor(ns, d); sl += __shfl_xor(sl, d); cx += __shfl_xor(cx, d);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { wmn[w] = mn; wmx[w] = mx; wpk[w] = pk; wns[w] = ns; wsl[w] = sl; wcx[w] = cx; wgn[w] = gn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        P0Part t;
        t.nseg = 0; t.ncixu = 0; t.nslow = 0; t.ngen = 0;
        for (int k = 0; k < (int)(blockDim.x >> 6); k++) {
            if (k) { mn = wmn[k] < mn ? wmn[k] : mn; mx = wmx[k] > mx ? wmx[k] : mx; pk = wpk[k] > pk ? wpk[k] : pk; }
            t.nseg += wns[k]; t.ncixu += wcx[k]; t.nslow += wsl[k]; t.ngen += wgn[k];
        }
        t.maxpos = pk; t.maxend = mx; t.minpos = mn; t.pad_ = 0;
        part[blockIdx.x] = t;                              /* (the extent comes out of k_p0_mid's totals: an atomic pair per
                                                            * workgroup on two words was still most of this kernel's 0.33 ms) */
    }
}

/* ONE 1024-thread workgroup: the block partials become what lies in front of each block (exclusive sums, prefix
 * maxima); [nblocks] receives the totals.  A contiguous slice per thread, the thread sums scanned through LDS. */
__global__ void __launch_bounds__(1024)
k_p0_mid(P0Part *part, int nblocks)
{
    __shared__ P0Part sh[1024];
    const int t = threadIdx.x, nt = blockDim.x;
    const int per = (nblocks + nt - 1) / nt;
    const int b = t * per, e = b + per < nblocks ? b + per : nblocks;
    P0Part acc;
    acc.nseg = acc.ncixu = acc.nslow = acc.ngen = 0; acc.maxpos = acc.maxend = INT_MIN; acc.minpos = INT_MAX; acc.pad_ = 0;
    for (int i = b; i < e; i++) {
        const P0Part v = part[i];
        acc.nseg += v.nseg; acc.ncixu += v.ncixu; acc.nslow += v.nslow; acc.ngen += v.ngen;
        acc.maxpos = v.maxpos > acc.maxpos ? v.maxpos : acc.maxpos; acc.maxend = v.maxend > acc.maxend ? v.maxend : acc.maxend;
        acc.minpos = v.minpos < acc.minpos ? v.minpos : acc.minpos;
    }
    sh[t] = acc;
    __syncthreads();
    for (int d = 1; d < nt; d <<= 1) {                    /* Hillis-Steele inclusive scan of the thread sums */
        P0Part v;
        v.nseg = v.ncixu = v.nslow = v.ngen = 0; v.maxpos = v.maxend = INT_MIN; v.minpos = INT_MAX; v.pad_ = 0;
        if (t >= d) v = sh[t - d];
        __syncthreads();
        P0Part x = sh[t];
        x.nseg += v.nseg; x.ncixu += v.ncixu; x.nslow += v.nslow; x.ngen += v.ngen;
        x.maxpos = v.maxpos > x.maxpos ? v.maxpos : x.maxpos; x.maxend = v.maxend > x.maxend ? v.maxend : x.maxend;
        x.minpos = v.minpos < x.minpos ? v.minpos : x.minpos;
        sh[t] = x;
        __syncthreads();
    }
    P0Part run;
    run.nseg = run.ncixu = run.nslow = run.ngen = 0; run.maxpos = run.maxend = INT_MIN; run.minpos = INT_MAX; run.pad_ = 0;
    if (t > 0) run = sh[t - 1];
    for (int i = b; i < e; i++) {
        const P0Part v = part[i];
        part[i] = run;
        run.nseg += v.nseg; run.ncixu += v.ncixu; run.nslow += v.nslow; run.ngen += v.ngen;
        run.maxpos = v.maxpos > run.maxpos ? v.maxpos : run.maxpos; run.maxend = v.maxend > run.maxend ? v.maxend : run.maxend;
        run.minpos = v.minpos < run.minpos ? v.minpos : run.minpos;
    }
    if (t == nt - 1) part[nblocks] = sh[nt - 1];
}

/* ---- k_p0_fill: the tables of every read.
 * Phase A, a lane per read: the block's own scan of the counts (wave scans + the sixteen wave totals through LDS) on top
 * of k_p0_mid's block prefixes gives every read its slice of the op tables; the same scan carries the prefix maxima of
 * start and end over the reads in the pileup -- the monotone search keys of k_p0_tiles (a read outside the pileup
 * inherits its predecessor's) -- and with them the coordinate-order test of snp:1075: a read in the pileup must not start
 * before the largest start in front of it.
 * Phase B, the ops of each read: GW = 1 here, a lane per read (short reads: one to three ops); GW = 0 leaves them to
 * k_p0_fill_long, a wave per read (long reads: hundreds of ops).  Every record is written once and never read back. */
struct P0Op {            /* one CIGAR op of a read with what lies in front of it */
    int k, op, l, x0, y0, si;
};
template <bool LONG>
__device__ __forceinline__ void p0_emit_op(const P0Op &o, int64_t r, int ncg, const uint32_t *__restrict__ cig, int rpos, int rend, int64_t so, int lq,
                                           unsigned mq, unsigned slot, int nseg, int64_t segbase, int64_t cixu, int p0,
                                           Seg *__restrict__ segs, SegRec *__restrict__ sr, ReadRec *__restrict__ rr,
                                           int64_t *__restrict__ slow_g, int32_t *__restrict__ slow_r, int64_t slow_slot,
                                           VRec *__restrict__ vr, int32_t *__restrict__ chunk_vr, int64_t nb, int hcol)
{
    const uint32_t ty = (o.op == 2) ? SEG_D : (o.op == 3) ? SEG_N : SEG_M;
    const int ia = p0_indel_after(cig, ncg, o.k);
    Seg s;
    s.roff = o.x0; s.qstart = o.y0; s.len_type = (uint32_t)o.l | (ty << 30); s.indel_after = ia;
    segs[segbase + o.si] = s;
    SegRec q;
    q.pos = rpos + o.x0; q.end = q.pos + o.l; q.so = so; q.lq = lq;
    q.qoff = (ty == SEG_M) ? o.y0 - q.pos : o.y0;
    q.indel_after = ia;
    q.info = mq | (ty << 8) | (slot << 16);
    if (o.si == 0) { q.info |= SR_FIRST; if (o.y0 > 0) q.info |= SR_HEADCLIP; }
    if (o.si == nseg - 1) {                                /* the last op carries the tail flags */
        const int qt = (ty == SEG_M) ? o.y0 + o.l - 1 : o.y0;
        q.info |= SR_LAST;
        if (qt + 1 < lq) q.info |= SR_TAILCLIP;
    }
    sr[segbase + o.si] = q;
    if (o.si == 0) {                                       /* the read's record goes with its first op */
        const bool simple = nseg == 1 && ty == SEG_M && ia == 0 && lq > 0;
        ReadRec x;
        x.pos = rpos; x.end = rend; x.so = so; x.lq = lq; x.q0 = o.y0;
        x.info = mq | (simple ? RR_SIMPLE : 0u); x.cix = simple ? 0u : (uint32_t)cixu;
        rr[r] = x;
    }
    /* bases that D/N columns report (htslib shows them the NEXT base): once per base -- a D/N op behind another one
     * with no query base in between reports the same base.  Every such op owns ONE slot of the list (slow_slot: the
     * read's slice from the scan of k_p0_classify's counts + the op's ordinal among the read's D/N ops), a repeated
     * base leaves its slot empty (read -1): no counter shared by all reads -- an atomic add per entry on one word was
     * 0.4 ms of the chr20 expansion and 2.3 ms of config 4's */
    if (ty != SEG_M && o.y0 < lq) {
        bool dup = false;
        for (int j = o.k - 1; j >= 0; j--) {
            const int oj = cig[j] & 15, lj = (int)(cig[j] >> 4);
            if (lj == 0) continue;
            if (p0_is_query(oj)) break;
            if (oj == 2 || oj == 3) { dup = true; break; }
        }
        slow_g[slow_slot] = so + o.y0; slow_r[slow_slot] = dup ? -1 : (int32_t)r;
    }
    if (LONG && vr) {
        /* the op's territory (VRec, dev_common.h): from its first query base up to the next reference op's */
        int qn = o.y0 + ((ty == SEG_M) ? o.l : 0);
        bool last = true;
        for (int j = o.k + 1; j < ncg; j++) {
            const int oj = cig[j] & 15, lj = (int)(cig[j] >> 4);
            if (p0_is_mdnx(oj) && lj > 0) { last = false; break; }
            if (oj == 1 || oj == 4) qn += lj;
        }
        if (last) qn = lq;
        const int qs = o.y0 < lq ? o.y0 : lq;
        if (qn > lq) qn = lq;
        if (qn < qs) qn = qs;
        VRec v;
        v.pos = v.end = 0; v.q0 = 0; v.hcol = hcol; v.read = (int32_t)r;
        if (ty == SEG_M) {
            const int len = o.l < lq - qs ? o.l : lq - qs;             /* fence: query bases that exist */
            v.pos = rpos + o.x0; v.end = v.pos + (len > 0 ? len : 0);
        }
        v.so = so + qs; v.lq = qn - qs;
        const int64_t vslot = segbase + r + 1 + o.si;
        vr[vslot] = v;
        const int64_t a = v.so, b = v.so + v.lq;
        for (int64_t k2 = (a + 15) >> 4; k2 * 16 < b; k2++) chunk_vr[k2] = (int32_t)vslot;
        if (b > a && b == nb) chunk_vr[(nb + 15) >> 4] = (int32_t)vslot;           /* the entry past the last base */
        if (o.si == 0) {                                   /* the territory in front of the first op */
            VRec f;
            f.pos = f.end = 0; f.q0 = 0; f.hcol = hcol; f.read = (int32_t)r; f.so = so; f.lq = qs;
            vr[vslot - 1] = f;
            for (int64_t k2 = (so + 15) >> 4; k2 * 16 < so + qs; k2++) chunk_vr[k2] = (int32_t)(vslot - 1);
            if (qs > 0 && so + qs == nb) chunk_vr[(nb + 15) >> 4] = (int32_t)(vslot - 1);
        }
    }
}

template <int GW>
__global__ void __launch_bounds__(1024)
k_p0_fill(int64_t n, int64_t nb, int p0, const int32_t *__restrict__ pos, const uint8_t *__restrict__ mapq,
          const int64_t *__restrict__ seq_off, const int64_t *__restrict__ cig_off,
          const uint32_t *__restrict__ cigar, const uint8_t *__restrict__ inplp,
          const int32_t *__restrict__ end, P0Cnt32 *__restrict__ cnt, const P0Part *__restrict__ part, int nblocks,
          Seg *__restrict__ segs, SegRec *__restrict__ sr, ReadRec *__restrict__ rr, int64_t *__restrict__ seg_off,
          int32_t *__restrict__ kpos, int32_t *__restrict__ kend,
          int64_t *__restrict__ slow_g, int32_t *__restrict__ slow_r, int *__restrict__ n_slow,
          int32_t *__restrict__ genlist, int *__restrict__ n_gen, const uint8_t *__restrict__ mq_slot,
          int32_t *__restrict__ blk_read, VRec *__restrict__ vr, int32_t *__restrict__ chunk_vr, int min_mqual, int *__restrict__ err)
{
    __shared__ int w_ns[16], w_ps[16], w_en[16], w_gn[16], w_sl[16];
    __shared__ long long w_cx[16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = r < n;
    /* ---- phase A */
    P0Cnt32 c = {0, 0, 0, 0};
    if (live) c = cnt[r];
    const bool in = live && inplp[r] != 0;
    const int rpos = live ? pos[r] : 0, rend = live ? end[r] : 0;
    const int ps = in ? rpos : INT_MIN, en = in ? rend : INT_MIN;
    const int i_ns = kc2_scan_add(c.nseg), i_ps = kc2_scan_max(ps), i_en = kc2_scan_max(en), i_sl = kc2_scan_add(c.nslow);
    long long i_cx = c.ncixu;
    #pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const long long o = __shfl_up(i_cx, d); if (lane >= d) i_cx += o; }
    int x_ps = __shfl_up(i_ps, 1);                         /* the largest start in front of this read, inside the wave */
    if (lane == 0) x_ps = INT_MIN;
    /* general-CIGAR reads with bases are the ones with table entries: their list (k_p0_cix, k_replay_reads) in file order */
    const bool general = c.ncixu > 0;
    const unsigned long long gm = __ballot(general);
    if (lane == 63) { w_ns[w] = i_ns; w_ps[w] = i_ps; w_en[w] = i_en; w_cx[w] = i_cx; w_gn[w] = __popcll(gm); w_sl[w] = i_sl; }
    __syncthreads();
    const P0Part pb = part[blockIdx.x];
    int64_t o_ns = pb.nseg + (i_ns - c.nseg), o_cx = pb.ncixu + (i_cx - c.ncixu);
    int64_t o_gn = pb.ngen + __popcll(gm & ((1ull << lane) - 1ull));
    int64_t o_sl = pb.nslow + (i_sl - c.nslow);             /* the read's slice of the list of bases under D/N ops */
    int k_ps = pb.maxpos, k_en = pb.maxend;
    for (int j = 0; j < w; j++) {
        o_ns += w_ns[j]; o_cx += w_cx[j]; o_gn += w_gn[j]; o_sl += w_sl[j];
        k_ps = w_ps[j] > k_ps ? w_ps[j] : k_ps; k_en = w_en[j] > k_en ? w_en[j] : k_en;
    }
    x_ps = x_ps > k_ps ? x_ps : k_ps;
    k_ps = i_ps > k_ps ? i_ps : k_ps; k_en = i_en > k_en ? i_en : k_en;
    int64_t so = 0;
    int lq = 0;
    if (live) {
        so = seq_off[r];
        lq = (int)(seq_off[r + 1] - so);
        if (in && rpos < x_ps) atomicMax(err, P0_ERR_UNSORTED);            /* snp:1075 */
        seg_off[r] = o_ns; kpos[r] = k_ps; kend[r] = k_en;
        if (r == n - 1) { seg_off[n] = part[nblocks].nseg; *n_gen = (int)part[nblocks].ngen; *n_slow = (int)part[nblocks].nslow; }
        if (GW == 0) { cnt[r].pad = (int32_t)(uint32_t)o_cx; cnt[r].nslow = (int32_t)o_sl; }   /* k_p0_fill_long takes them from there */
        if (general && genlist) genlist[o_gn] = (int32_t)r;
        if (!in) {                                         /* never in the pileup: only the epilogue touches it */
            ReadRec x;
            x.pos = rpos; x.end = rend; x.so = so; x.lq = lq; x.q0 = 0; x.info = mapq[r]; x.cix = 0;
            rr[r] = x;
            if (GW == 0 && vr) {                           /* its one territory: the whole read */
                VRec f;
                f.pos = f.end = 0; f.q0 = 0; f.hcol = -1; f.read = (int32_t)r; f.so = so; f.lq = lq;
                const int64_t vslot = o_ns + r;
                vr[vslot] = f;
                for (int64_t k2 = (so + 15) >> 4; k2 * 16 < so + lq; k2++) chunk_vr[k2] = (int32_t)vslot;
                if (lq > 0 && so + lq == nb) chunk_vr[(nb + 15) >> 4] = (int32_t)vslot;
            }
        }
        if (GW == 1 && blk_read && lq > 0) {               /* the read owns the 64-base blocks whose first base is its own */
            for (int64_t k2 = (so + 63) >> 6; k2 * 64 < so + lq; k2++) blk_read[k2] = (int32_t)r;
            if (so + lq == nb) blk_read[(nb + 63) >> 6] = (int32_t)r;       /* the entry past the last base (readers clamp) */
        }
    }
    /* ---- phase B (GW == 0: k_p0_fill_long does it, a wave per read) */
    if (GW == 1) {
        if (in) {
            const uint32_t *cig = cigar + cig_off[r];
            const int ncg = (int)(cig_off[r + 1] - cig_off[r]);
            const unsigned mq = mapq[r], slot = mq_slot[mq];
            const int hcol = ((int)mq <= min_mqual) ? rpos - p0 : -1;
            P0Op o;
            o.x0 = 0; o.y0 = 0; o.si = 0;
            int dn = 0;                                    /* D/N ops with a base to report so far (k_p0_classify's count) */
            for (int k = 0; k < ncg; k++) {
                o.k = k; o.op = cig[k] & 15; o.l = (int)(cig[k] >> 4);
                if (p0_is_mdnx(o.op)) {
                    if (o.l > 0) {
                        p0_emit_op<false>(o, r, ncg, cig, rpos, rend, so, lq, mq, slot, c.nseg, o_ns, o_cx, p0, segs, sr, rr,
                                          slow_g, slow_r, o_sl + dn, vr, chunk_vr, nb, hcol);
                        if ((o.op == 2 || o.op == 3) && o.y0 < lq) dn++;
                        o.si++;
                    }
                    o.x0 += o.l;
                    if (o.op != 2 && o.op != 3) o.y0 += o.l;
                } else if (o.op == 1 || o.op == 4) o.y0 += o.l;
            }
        }
    }
}

/* Phase B for long reads: ONE WAVE PER READ, 64 CIGAR ops per step with wave scans for the reference / query offsets
 * (hundreds of ops per read; a lane per read left 331 k lanes walking 300 ops each, and the waves of a 1024-read
 * workgroup taking their 64 reads one after another left the chip a few hundred workgroups).  k_p0_fill<0> has
 * written every read's slice of the tables (seg_off, the base -> column offset in cnt[r].pad) before. */
__global__ void __launch_bounds__(256)
k_p0_fill_long(int64_t n, int64_t nb, int p0, const int32_t *__restrict__ pos, const uint8_t *__restrict__ mapq,
               const int64_t *__restrict__ seq_off, const int64_t *__restrict__ cig_off,
               const uint32_t *__restrict__ cigar, const uint8_t *__restrict__ inplp,
               const int32_t *__restrict__ end, const P0Cnt32 *__restrict__ cnt,
               Seg *__restrict__ segs, SegRec *__restrict__ sr, ReadRec *__restrict__ rr, const int64_t *__restrict__ seg_off,
               int64_t *__restrict__ slow_g, int32_t *__restrict__ slow_r,
               const uint8_t *__restrict__ mq_slot, int32_t *__restrict__ blk_read, VRec *__restrict__ vr,
               int32_t *__restrict__ chunk_vr, int min_mqual)
{
    const int lane = threadIdx.x & 63;
    const int64_t ri = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (ri >= n) return;                                   /* (the whole wave) */
    const int64_t so_i = seq_off[ri];
    const int lq_i = (int)(seq_off[ri + 1] - so_i);
    if (lq_i > 0 && blk_read) {                            /* block -> read table, the wave's lanes over the read's blocks */
        for (int64_t k2 = ((so_i + 63) >> 6) + lane; k2 * 64 < so_i + lq_i; k2 += 64) blk_read[k2] = (int32_t)ri;
        if (lane == 0 && so_i + lq_i == nb) blk_read[(nb + 63) >> 6] = (int32_t)ri;
    }
    if (!inplp[ri]) return;
    const P0Cnt32 c = cnt[ri];
    const int rpos_i = pos[ri], rend_i = end[ri], nseg_i = c.nseg;
    const int64_t ons_i = seg_off[ri], ocx_i = (int64_t)(uint32_t)c.pad, osl_i = (int64_t)(uint32_t)c.nslow;
    const int64_t co = cig_off[ri];
    const uint32_t *cig = cigar + co;
    const int ncg = (int)(cig_off[ri + 1] - co);
    const unsigned mq = mapq[ri], slot = mq_slot[mq];
    const int hcol = ((int)mq <= min_mqual) ? rpos_i - p0 : -1;
    int cx0 = 0, cy0 = 0, cs0 = 0, cd0 = 0;
    for (int k0 = 0; k0 < ncg; k0 += 64) {
        const int k = k0 + lane;
        P0Op o;
        o.k = k; o.op = -1; o.l = 0;
        if (k < ncg) { o.op = cig[k] & 15; o.l = (int)(cig[k] >> 4); }
        const bool isref = o.op >= 0 && p0_is_mdnx(o.op);
        const int dx = isref ? o.l : 0;
        const int dy = (o.op == 0 || o.op == 1 || o.op == 4 || o.op == 7 || o.op == 8) ? o.l : 0;
        const int ds = (isref && o.l > 0) ? 1 : 0;
        const int ix = kc2_scan_add(dx), iy = kc2_scan_add(dy), is = kc2_scan_add(ds);
        o.x0 = cx0 + ix - dx; o.y0 = cy0 + iy - dy; o.si = cs0 + is - ds;
        const int dd = (ds && (o.op == 2 || o.op == 3) && o.y0 < lq_i) ? 1 : 0;      /* a D/N op with a base to report */
        const int id = kc2_scan_add(dd);
        if (ds)
            p0_emit_op<true>(o, ri, ncg, cig, rpos_i, rend_i, so_i, lq_i, mq, slot, nseg_i, ons_i, ocx_i, p0, segs, sr, rr,
                             slow_g, slow_r, osl_i + cd0 + id - dd, vr, chunk_vr, nb, hcol);
        cx0 += __shfl(ix, 63); cy0 += __shfl(iy, 63); cs0 += __shfl(is, 63); cd0 += __shfl(id, 63);
    }
}

/* base -> column table of the general-CIGAR reads of a short-read batch (replay): a wave per read of the list, lanes
 * over its bases; -1 no column, -2 a D/N column reports this base (k_replay_slow) */
__global__ void __launch_bounds__(256)
k_p0_cix(int p0, const int32_t *__restrict__ genlist, const int *__restrict__ n_gen, const ReadRec *__restrict__ rr,
         const int64_t *__restrict__ seg_off, const Seg *__restrict__ segs, int32_t *__restrict__ cix)
{
    const int lane = threadIdx.x & 63;
    const int ng = *n_gen, nw = (int)(gridDim.x * (blockDim.x >> 6));
    for (int gi = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)); gi < ng; gi += nw) {
        const int64_t r = genlist[gi];
        const ReadRec x = rr[r];
        const int64_t s0 = seg_off[r];
        const int ns = (int)(seg_off[r + 1] - s0);
        int32_t *cx = cix + (size_t)x.cix * CIX_UNIT;
        for (int i0 = 0; i0 < x.lq; i0 += 64) {
            const int i = i0 + lane;
            int val = -1;
            bool mark = false;
            for (int k = 0; k < ns; k++) {
                const Seg sg = segs[s0 + k];
                const int off = i - sg.qstart;
                if (seg_type(sg) == SEG_M) { if (off >= 0 && off < seg_len(sg)) val = (x.pos + sg.roff + off) - p0; }
                else if (off == 0) mark = true;
            }
            if (i < x.lq) cx[i] = mark ? -2 : val;
        }
    }
}

/* For tile t = columns [c0, c0+63]: hi = first read whose (prefix-max) start key > c0+63, lo = min(hi, first read whose
 * (prefix-max) end key > c0): exactly the two-pointer walk over position-sorted reads (entries between lo and hi that
 * ended earlier are skipped by the kernels).  The same range in op records is [seg_off[lo], seg_off[hi]): a read's ops
 * carry its start, and the prefix maximum of `end` over op records is the one over reads.  With keep-regions: the first
 * region whose prefix-max of `end` reaches the tile's first column -- every column's own cursor (bed_preserve,
 * kernels.h) is at or behind it. */
__global__ void k_p0_tiles(int n_tiles, int p0, int64_t n, const int32_t *__restrict__ kpos, const int32_t *__restrict__ kend,
                           const int64_t *__restrict__ seg_off, const SegRec *__restrict__ sr,
                           int32_t *__restrict__ tlo, int32_t *__restrict__ thi, int32_t *__restrict__ slo, int32_t *__restrict__ shi,
                           int n_bed, const int32_t *__restrict__ bed_pmax, int32_t *__restrict__ tbed, int *__restrict__ err)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    const int64_t c0 = (int64_t)p0 + (int64_t)t * 64, c1 = c0 + 63;
    int64_t lo = 0, hi = n;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if ((int64_t)kpos[mid] <= c1) lo = mid + 1; else hi = mid; }
    const int64_t h = lo;
    lo = 0; hi = h;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if ((int64_t)kend[mid] <= c0) lo = mid + 1; else hi = mid; }
    tlo[t] = (int32_t)lo; thi[t] = (int32_t)h;
    const int64_t s0 = seg_off[lo], s1 = seg_off[h];
    slo[t] = (int32_t)s0; shi[t] = (int32_t)s1;
    /* k_columns2 addresses the bases of a tile's reads with 32-bit offsets from the first one */
    if (s1 > s0 && sr[s1 - 1].so - sr[s0].so > 0xf0000000ll) atomicMax(err, P0_ERR_SPAN);
    if (n_bed > 0) {
        int a = 0, b = n_bed;
        while (a < b) { const int mid = (a + b) >> 1; if ((int64_t)bed_pmax[mid] < c0) a = mid + 1; else b = mid; }
        tbed[t] = a;
    }
}

/* The bases as the kernels read them: one byte per base, the nt16 code in the low nibble and the consensus class in the
 * high one (what the reference derives per pileup entry, snp:603-609: A C G T -> 0..3, anything else -> 5 "N"; 4 is
 * '*'), so that k_columns2 stages classes with a shift instead of a per-byte decode.  Every reader of `bases` masks
 * with 15. */
__device__ __forceinline__ uint32_t p0_class4(uint32_t w)
{
    const uint32_t c = w & 0x0f0f0f0fu;
    const uint32_t idx = ((c >> 1) & 0x07070707u) - ((c >> 3) & 0x01010101u);          /* one-hot: 1 2 4 8 -> 0 1 2 3 */
    const uint32_t many = c & ((c | 0x80808080u) - 0x01010101u) & 0x0f0f0f0fu;         /* >= 2 bits set */
    const uint32_t nz_many = (((many + 0x7f7f7f7fu) & 0x80808080u) >> 7) * 0xffu;
    const uint32_t nz_c = (((c + 0x7f7f7f7fu) & 0x80808080u) >> 7) * 0xffu;
    const uint32_t bad = nz_many | ~nz_c;                                               /* not exactly one bit */
    const uint32_t cls = (idx & ~bad) | (0x05050505u & bad);
    return c | (cls << 4);
}
/* qualities >= 128 of a batch that is not run (crumble_hip_indel_only): the run's own check is k_stats */
__global__ void k_p0_qualcheck(int64_t nb, const uint8_t *__restrict__ quals, int *err)
{
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    if (i >= nb) return;
    bool bad = false;
    if (i + 16 <= nb) { const uint4 v = *(const uint4 *)(quals + i); bad = ((v.x | v.y | v.z | v.w) & 0x80808080u) != 0; }
    else for (int64_t k = i; k < nb; k++) bad = bad || (quals[k] & 0x80);
    if (bad) atomicMax(err, P0_ERR_QUAL);
}

/* k_p0_bases.  PACKED: crumble_batch.bases4 (BAM's two codes per byte, the earlier base in the high nibble; read i from
 * byte (seq_off[i] >> 1) + i) -> one classed byte per base; otherwise the class is added to the one-code-per-byte
 * input in place.  OWNER: chunk_read[] (the read that owns the first base of every 16-base chunk: k_replay_fast starts
 * there) is written on the way.  A 256-lane workgroup takes 4096 consecutive bases, a lane 16 of them (one aligned
 * 16-byte store).  The reads that own them are consecutive and so are their packed bytes: their offsets and the ~2 KB
 * of packed bytes are staged in LDS with coalesced loads (first / last read from the block table: wave-uniform
 * look-ups), a chunk's owner comes from a scatter over the staged reads, and a lane's 8 or 9 packed bytes are three
 * LDS words and a byte funnel -- no per-lane chain of dependent global loads (block table -> offsets -> bytes: 1.66 ms
 * for chr20's 1.9 G bases, now bound by the 0.5 + 1 bytes per base it moves).  A chunk that straddles reads is made once
 * per read as if the whole chunk lay in that read and merged by byte mask.  Workgroups whose bases belong to more than
 * P0B_MAXR reads (reads of a few bases) take the same steps from global memory. */
#define P0B_THREADS 256
#ifndef P0B_CPL
#define P0B_CPL     4                                  /* 16-base chunks per lane, P0B_THREADS chunks apart (coalesced stores) */
#endif
#define P0B_CHUNKS  (P0B_THREADS * P0B_CPL)
#define P0B_BASES   (P0B_CHUNKS * 16)
#define P0B_MAXR    512
#define P0B_SEARCH  32                                 /* at most this many reads over a workgroup's bases: chunk owners by search */
#define P0_ILV_LO(X, Y) (((X) & 0xffu) | (((Y) & 0xffu) << 8) | (((X) & 0xff00u) << 8) | (((Y) & 0xff00u) << 16))      /* x0 y0 x1 y1 */
#define P0_ILV_HI(X, Y) ((((X) >> 16) & 0xffu) | ((((Y) >> 16) & 0xffu) << 8) | ((((X) >> 16) & 0xff00u) << 8) | ((((Y) >> 16) & 0xff00u) << 16))   /* x2 y2 x3 y3 */
/* sixteen classed bytes from the packed bytes d0 d1 d2 (twelve bytes from the one that holds the chunk's first base);
 * odd: the first base is that byte's low nibble */
__device__ __forceinline__ void p0_unpack16(uint32_t d0, uint32_t d1, uint32_t d2, bool odd, uint32_t (&v)[4])
{
    const uint32_t h0 = (d0 >> 4) & 0x0f0f0f0fu, l0 = d0 & 0x0f0f0f0fu, h1 = (d1 >> 4) & 0x0f0f0f0fu, l1 = d1 & 0x0f0f0f0fu;
    uint32_t xa, ya, xb, yb;                              /* output byte 2j = xa/xb[j], 2j + 1 = ya/yb[j] */
    if (!odd) { xa = h0; ya = l0; xb = h1; yb = l1; }
    else {                                                /* base 2j is the low nibble of byte j, base 2j + 1 the high one of byte j + 1 */
        const uint32_t h2 = (d2 >> 4) & 0x0f0f0f0fu;
        xa = l0; ya = (h0 >> 8) | (h1 << 24); xb = l1; yb = (h1 >> 8) | (h2 << 24);
    }
    v[0] = p0_class4(P0_ILV_LO(xa, ya)); v[1] = p0_class4(P0_ILV_HI(xa, ya));
    v[2] = p0_class4(P0_ILV_LO(xb, yb)); v[3] = p0_class4(P0_ILV_HI(xb, yb));
}
template <bool PACKED, bool OWNER>
__global__ void __launch_bounds__(P0B_THREADS)
k_p0_bases(int64_t nb, int64_t n, const int64_t *__restrict__ seq_off, const int32_t *__restrict__ blk_read,
           const uint8_t *__restrict__ b4, uint8_t *__restrict__ bases, int32_t *__restrict__ chunk_read)
{
    __shared__ long long s_so[P0B_MAXR + 2];
    __shared__ uint32_t s_b4[PACKED ? (P0B_BASES / 2 + P0B_MAXR + 64) / 4 : 4];
    __shared__ uint16_t s_owner[P0B_CHUNKS];
    __shared__ uint16_t s_rest[PACKED ? P0B_CHUNKS : 1];   /* the chunks that do not lie inside one read: done after the others */
    __shared__ unsigned s_nrest;
    const int64_t G0 = (int64_t)blockIdx.x * P0B_BASES;
    if (G0 >= nb) return;
    if (threadIdx.x == 0) s_nrest = 0;
    const int64_t G1 = G0 + P0B_BASES < nb ? G0 + P0B_BASES : nb;
    /* first and last owner: wave-uniform walks from the block table */
    int64_t r0 = blk_read[G0 >> 6], r1 = blk_read[(G1 - 1) >> 6];
    while (r0 + 1 < n && seq_off[r0 + 1] <= G0) r0++;
    while (r1 + 1 < n && seq_off[r1 + 1] <= G1 - 1) r1++;
    const int nr = (int)(r1 - r0 + 1);
    const bool staged = nr <= P0B_MAXR;
    int64_t B0 = 0;
    if (staged) {
        for (int j = threadIdx.x; j <= nr; j += P0B_THREADS) s_so[j] = seq_off[r0 + j];
        if (PACKED) {
            const int64_t so0 = seq_off[r0];
            B0 = ((so0 >> 1) + r0 + ((G0 - so0) >> 1)) & ~(int64_t)3;
            const int64_t so1 = seq_off[r1];
            const int64_t B1 = (so1 >> 1) + r1 + ((G1 - 1 - so1) >> 1) + 1 + 12;       /* a lane looks at twelve bytes */
            const int nw16 = (int)((B1 - B0 + 15) >> 4);
            const uint4 *src = (const uint4 *)(b4 + (B0 & ~(int64_t)15));               /* (b4 is 16-byte aligned; the array carries slack) */
            const int sh = (int)(B0 & 15) >> 2;                                           /* B0 is a multiple of 4 */
            for (int j = threadIdx.x; j <= nw16; j += P0B_THREADS) {
                const uint4 v = src[j];
                const int o = 4 * j - sh;
                if (o >= 0) s_b4[o] = v.x;
                if (o + 1 >= 0) s_b4[o + 1] = v.y;
                if (o + 2 >= 0) s_b4[o + 2] = v.z;
                if (o + 3 >= 0) s_b4[o + 3] = v.w;
            }
        }
        __syncthreads();
        /* owner of every chunk, the read that holds its first base: by scatter from the reads when they are many and
         * short (a few chunks each); a lane searches the staged offsets for its own chunks when they are few and long
         * (one lane writing a long read's thousand entries kept the workgroup waiting) */
        if (nr > P0B_SEARCH) {
            for (int j = threadIdx.x; j < nr; j += P0B_THREADS) {
                const int64_t a = s_so[j] > G0 ? s_so[j] : G0, b = s_so[j + 1] < G1 ? s_so[j + 1] : G1;
                for (int64_t k = (a + 15) >> 4; k * 16 < b; k++) s_owner[(int)(k - (G0 >> 4))] = (uint16_t)j;
            }
            __syncthreads();
        }
    } else __syncthreads();                                       /* (s_nrest = 0 is seen by every lane) */
    /* Round 5: the chunks that lie inside ONE read (nine in ten on 150-base reads) take a straight path here -- owner, its
     * packed bytes from LDS, one unpack, one 16-byte store; the others (a read boundary inside the chunk, bytes in front of
     * the staged range, the batch's last partial chunk, workgroups of tiny reads) are listed and done afterwards by as
     * many lanes as there are such chunks.  Before, every wave ran the general per-owner loop twice for each of its four
     * chunks because SOME lane's chunk straddled two reads: 1296 VALU per lane, the kernel's bound (VALU 96 % busy). */
    #pragma unroll
    for (int u = 0; u < P0B_CPL; u++) {
        const int ci = u * P0B_THREADS + (int)threadIdx.x;
        const int64_t g = G0 + (int64_t)ci * 16;
        if (g >= nb) continue;
        const int64_t gend = g + 16 < nb ? g + 16 : nb;
        int64_t r;
        if (staged && nr > P0B_SEARCH) r = r0 + s_owner[ci];
        else if (staged) {                                        /* last staged read whose first base is at or before g */
            int lo = 0, hi = nr - 1;
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_so[mid] <= g) lo = mid; else hi = mid - 1; }
            while (lo + 1 < nr && s_so[lo + 1] <= g) lo++;        /* (zero-length reads share an offset: the owner is the last of them) */
            r = r0 + lo;
        }
        else { r = blk_read[g >> 6]; while (r + 1 < n && seq_off[r + 1] <= g) r++; }
        if (OWNER) {
            chunk_read[g >> 4] = (int32_t)r;
            if (gend == nb) chunk_read[(nb + 15) >> 4] = (int32_t)r1;                 /* the entry past the last base (readers clamp) */
        }
        if (!PACKED) {
            if (gend - g == 16) {
                uint4 b = *(const uint4 *)(bases + g);
                b.x = p0_class4(b.x); b.y = p0_class4(b.y); b.z = p0_class4(b.z); b.w = p0_class4(b.w);
                *(uint4 *)(bases + g) = b;
            } else for (int64_t k = g; k < gend; k++) bases[k] = (uint8_t)(p0_class4(bases[k]) & 0xffu);
            continue;
        }
        bool fast = staged && gend - g == 16;
        if (fast) {
            const int64_t so = (int64_t)s_so[r - r0], e = (int64_t)s_so[r - r0 + 1];
            const int64_t x = g - so, sb = (so >> 1) + r + (x >> 1);
            fast = so <= g && e >= gend && sb >= B0;
            if (fast) {
                const int o = (int)(sb - B0), wi = o >> 2;
                const unsigned sh = (unsigned)o & 3u;
                const uint32_t w0 = s_b4[wi], w1 = s_b4[wi + 1], w2 = s_b4[wi + 2], w3 = s_b4[wi + 3];
                uint32_t v[4];
                p0_unpack16(kc2_alignbyte(w1, w0, sh), kc2_alignbyte(w2, w1, sh), kc2_alignbyte(w3, w2, sh), (x & 1) != 0, v);
                uint4 o4; o4.x = v[0]; o4.y = v[1]; o4.z = v[2]; o4.w = v[3];
                *(uint4 *)(bases + g) = o4;
            }
        }
        if (!fast) s_rest[atomicAdd(&s_nrest, 1u)] = (uint16_t)ci;
    }
    if (!PACKED) return;
    __syncthreads();
    const int nrest = (int)s_nrest;
    for (int i = (int)threadIdx.x; i < nrest; i += P0B_THREADS) {
        const int ci = (int)s_rest[i];
        const int64_t g = G0 + (int64_t)ci * 16;
        const int64_t gend = g + 16 < nb ? g + 16 : nb;
        int64_t r;                                                /* the chunk's owner once more */
        if (staged && nr > P0B_SEARCH) r = r0 + s_owner[ci];
        else if (staged) {
            int lo = 0, hi = nr - 1;
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_so[mid] <= g) lo = mid; else hi = mid - 1; }
            while (lo + 1 < nr && s_so[lo + 1] <= g) lo++;
            r = r0 + lo;
        } else { r = blk_read[g >> 6]; while (r + 1 < n && seq_off[r + 1] <= g) r++; }
        uint32_t ow[4] = {0, 0, 0, 0};
        for (;; r++) {
            const int64_t so = staged ? (int64_t)s_so[r - r0] : seq_off[r], e = staged ? (int64_t)s_so[r - r0 + 1] : seq_off[r + 1];
            const int lo = so > g ? (int)(so - g) : 0, hi = e < gend ? (int)(e - g) : (int)(gend - g);
            if (lo < hi) {
                /* as if the whole chunk lay in this read (x may be negative: the bytes before the read are masked away) */
                const int64_t x = g - so;                         /* (>> 1 floors, & 1 is the parity, also when negative) */
                const int64_t sb = (so >> 1) + r + (x >> 1);
                uint32_t d0, d1, d2;
                if (staged && sb >= B0) {
                    const int o = (int)(sb - B0), wi = o >> 2;
                    const unsigned sh = (unsigned)o & 3u;
                    const uint32_t w0 = s_b4[wi], w1 = s_b4[wi + 1], w2 = s_b4[wi + 2], w3 = s_b4[wi + 3];
                    d0 = kc2_alignbyte(w1, w0, sh); d1 = kc2_alignbyte(w2, w1, sh); d2 = kc2_alignbyte(w3, w2, sh);
                } else {                                          /* bytes in front of the staged range (a chunk's bytes before its read), or no staging */
                    uint32_t t3[3] = {0, 0, 0};
                    for (int k = 0; k < 12; k++) if (sb + k >= 0) t3[k >> 2] |= (uint32_t)b4[sb + k] << (8 * (k & 3));
                    d0 = t3[0]; d1 = t3[1]; d2 = t3[2];
                }
                uint32_t v[4];
                p0_unpack16(d0, d1, d2, (x & 1) != 0, v);
                const unsigned vb = ((1u << hi) - 1u) & ~((1u << lo) - 1u);
                #pragma unroll
                for (int t = 0; t < 4; t++) {
                    const uint32_t M = ((((vb >> (4 * t)) & 15u) * 0x00204081u) & 0x01010101u) * 0xffu;
                    ow[t] = (v[t] & M) | (ow[t] & ~M);
                }
            }
            if (e >= gend || r + 1 >= n) break;
        }
        if (gend - g == 16) { uint4 o; o.x = ow[0]; o.y = ow[1]; o.z = ow[2]; o.w = ow[3]; *(uint4 *)(bases + g) = o; }
        else for (int64_t t = g; t < gend; t++) bases[t] = (uint8_t)(ow[(t - g) >> 2] >> (8 * ((t - g) & 3)));
    }
}
#undef P0_ILV_LO
#undef P0_ILV_HI

/* Long-read batches (k_columns2<MULTI>): the records (reference-consuming ops) that reach each 64-column tile, listed
 * once per batch in file order.  A tile's candidate range [stile_lo, stile_hi) holds every op of ~20 overlapping
 * reads, ~4000 records of which ~25 reach the tile: screening them in every run was 18 % of the kernel on config 4.
 * Tile t = columns [c0, c0 + 63], c0 = p0 + 64 t; a record reaches it iff end > c0 && pos <= c0 + 63.  A read's ops
 * are position-sorted and contiguous (end == the next op's pos), so the ops of read r that reach its k-th tile are a
 * RANGE F_k .. L_k of record indices, and L_k follows from F_(k+1):
 *   k_p0_rtcount    one lane per read: the number of tiles it reaches (exclusive scan: rt_off[])
 *   k_p0_first      one wave per read, lanes over its ops: F_k for every tile of the read, a plain store per
 *                   (read, tile); bit 31 = the op starts on the tile's first column (or the read does): it does not
 *                   reach the tile before
 *   k_p0_hitcount   sixteen lanes per tile, a lane per overlapping read: L_k - F_k + 1, summed per tile
 *                   (exclusive scan: hit_off[])
 *   k_p0_hitfill    the same walk, the ranges written out read by read: file order, what the column kernel's sums
 *                   need.  No atomics anywhere, no record is looked at after k_p0_first.
 * (First version: one atomic add per (record, tile) pair, then a screening pass over the candidate ranges: 6.1 ms per
 * upload of config 4.  Second: counted and filled in one kernel with one atomic add per tile on ONE cursor -- 4 M
 * same-address atomics, 11.7 ms.) */
#define P0_FIRST_STARTS 0x80000000u
__global__ void k_p0_rtcount(int64_t n, int p0, const ReadRec *__restrict__ rr, const int64_t *__restrict__ seg_off,
                             int64_t *__restrict__ rt)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const ReadRec x = rr[r];
    rt[r] = (seg_off[r + 1] > seg_off[r] && x.end > x.pos) ? (int64_t)(((x.end - 1 - p0) >> 6) - ((x.pos - p0) >> 6) + 1) : 0;
}
__global__ void __launch_bounds__(256)
k_p0_first(int64_t n, int p0, const ReadRec *__restrict__ rr, const int64_t *__restrict__ seg_off, const SegRec *__restrict__ sr,
           const int64_t *__restrict__ rt_off, uint32_t *__restrict__ first)
{
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= n) return;
    const int64_t s0 = seg_off[r], s1 = seg_off[r + 1];
    if (s1 <= s0) return;
    const int t0 = (rr[r].pos - p0) >> 6;
    const int64_t ro = rt_off[r];
    for (int64_t i = s0 + lane; i < s1; i += 64) {
        const int2 pe = *(const int2 *)&sr[i];              /* pos, end */
        if (pe.y <= pe.x) continue;
        const int ta = (pe.x - p0) >> 6, tb = (pe.y - 1 - p0) >> 6;
        /* the op before ends where this one starts: it reaches tile ta too unless this op starts the tile */
        const bool starts = i == s0 || ((pe.x - p0) & 63) == 0;
        if (starts) first[ro + (ta - t0)] = (uint32_t)i | P0_FIRST_STARTS;
        for (int t = ta + 1; t <= tb; t++) first[ro + (t - t0)] = (uint32_t)i;
    }
}
#define P0_HIT_GROUP 16         /* lanes per tile in k_p0_hitcount / k_p0_hitfill */
/* the lane's read of round k_ over tile `tile`: the range [fi_, fi_ + cnt_) of its records that reach the tile */
#define P0_HIT_READ(k_, fi_, cnt_) do { \
        const int r_ = lo + (k_) * P0_HIT_GROUP + gl; \
        (fi_) = 0; (cnt_) = 0; \
        if (r_ < hi) { \
            const ReadRec x_ = rr[r_]; \
            const int64_t s1_ = seg_off[r_ + 1]; \
            if (s1_ > seg_off[r_] && x_.end > c0 && x_.pos <= c1) { \
                const int k0_ = tile - ((x_.pos - p0) >> 6); \
                const int64_t ro_ = rt_off[r_]; \
                (fi_) = (int)(first[ro_ + k0_] & ~P0_FIRST_STARTS); \
                if (tile < ((x_.end - 1 - p0) >> 6)) { \
                    const uint32_t nx_ = first[ro_ + k0_ + 1]; \
                    (cnt_) = (int)(nx_ & ~P0_FIRST_STARTS) - (fi_) + ((nx_ & P0_FIRST_STARTS) ? 0 : 1); \
                } else (cnt_) = (int)(s1_ - (fi_)); \
            } \
        } } while (0)
__global__ void __launch_bounds__(256)
k_p0_hitcount(int n_tiles, int p0, const int32_t *__restrict__ tlo, const int32_t *__restrict__ thi,
              const ReadRec *__restrict__ rr, const int64_t *__restrict__ seg_off, const int64_t *__restrict__ rt_off,
              const uint32_t *__restrict__ first, int64_t *__restrict__ cnt_out)
{
    const int gl = threadIdx.x & (P0_HIT_GROUP - 1);
    const int tile = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / P0_HIT_GROUP);
    const bool live = tile < n_tiles;                       /* (every lane stays for the group's exchanges) */
    const int c0 = p0 + tile * 64, c1 = c0 + 63;
    const int lo = live ? tlo[tile] : 0, hi = live ? thi[tile] : 0;
    int total = 0;
    for (int k = 0; lo + k * P0_HIT_GROUP < hi; k++) {
        int fi, cnt;
        P0_HIT_READ(k, fi, cnt);
        (void)fi;
        total += cnt;
    }
    for (int d = 1; d < P0_HIT_GROUP; d <<= 1) total += __shfl_xor(total, d);
    if (live && gl == 0) cnt_out[tile] = total;
}
__global__ void __launch_bounds__(256)
k_p0_hitfill(int n_tiles, int p0, const int32_t *__restrict__ tlo, const int32_t *__restrict__ thi,
             const ReadRec *__restrict__ rr, const int64_t *__restrict__ seg_off, const int64_t *__restrict__ rt_off,
             const uint32_t *__restrict__ first, const int64_t *__restrict__ hit_off, int32_t *__restrict__ hits, int *__restrict__ err)
{
    const int gl = threadIdx.x & (P0_HIT_GROUP - 1);
    const int tile = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / P0_HIT_GROUP);
    const bool live = tile < n_tiles;
    const int c0 = p0 + tile * 64, c1 = c0 + 63;
    const int lo = live ? tlo[tile] : 0, hi = live ? thi[tile] : 0;
    int64_t o = live ? hit_off[tile] : 0;
    const int64_t oe = live ? hit_off[tile + 1] : 0;
    int nround = (hi - lo + P0_HIT_GROUP - 1) / P0_HIT_GROUP;
    for (int d = P0_HIT_GROUP; d < 64; d <<= 1) { const int v = __shfl_xor(nround, d); nround = v > nround ? v : nround; }   /* the wave's tiles in step */
    const int glast = ((int)(threadIdx.x & 63) & ~(P0_HIT_GROUP - 1)) + P0_HIT_GROUP - 1;
    for (int k = 0; k < nround; k++) {
        int fi, cnt;
        P0_HIT_READ(k, fi, cnt);
        int inc = cnt;                                      /* inclusive prefix over the group's lanes */
        for (int d = 1; d < P0_HIT_GROUP; d <<= 1) { const int v = __shfl_up(inc, d, P0_HIT_GROUP); if (gl >= d) inc += v; }
        const int64_t w = o + (inc - cnt);
        for (int j = 0; j < cnt; j++) if (w + j < oe) hits[w + j] = fi + j;
        o += __shfl(inc, glast);
    }
    if (live && gl == 0 && o != oe) *err = 1;               /* (the two kernels walk the same ranges) */
}
#undef P0_HIT_READ

/* chunked exclusive scan over int64 (the hit lists' offsets): a 256-thread workgroup per chunk of P0_CHUNK values,
 * coalesced (a thread per chunk walking its 1024 values touched 64 cache lines per load instruction) */
__global__ void __launch_bounds__(256)
k_i64scan_reduce(int64_t n, const int64_t *__restrict__ v, int64_t *__restrict__ partial, int nchunks)
{
    __shared__ long long ws[4];
    const int64_t b = (int64_t)blockIdx.x * P0_CHUNK;
    long long acc = 0;
    #pragma unroll
    for (int k = 0; k < P0_CHUNK / 256; k++) { const int64_t i = b + k * 256 + threadIdx.x; if (i < n) acc += v[i]; }
    #pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
__global__ void __launch_bounds__(1024)
k_i64scan_partials(int64_t *partial, int nchunks)
{
    __shared__ int64_t sh[1024];
    partial += (size_t)blockIdx.x * ((size_t)nchunks + 1);      /* (grid > 1: several arrays of nchunks + 1 values, one workgroup each) */
    const int t = threadIdx.x, nt = blockDim.x;
    const int per = (nchunks + nt - 1) / nt;
    const int b = t * per, e = b + per < nchunks ? b + per : nchunks;
    int64_t acc = 0;
    for (int i = b; i < e; i++) acc += partial[i];
    sh[t] = acc;
    __syncthreads();
    for (int d = 1; d < nt; d <<= 1) {
        int64_t v = 0;
        if (t >= d) v = sh[t - d];
        __syncthreads();
        sh[t] += v;
        __syncthreads();
    }
    int64_t run = t > 0 ? sh[t - 1] : 0;
    for (int i = b; i < e; i++) { const int64_t v = partial[i]; partial[i] = run; run += v; }
    if (t == nt - 1) partial[nchunks] = sh[nt - 1];
}
/* exclusive scan in place; v[n] receives the total */
__global__ void __launch_bounds__(256)
k_i64scan_apply(int64_t n, int64_t *__restrict__ v, const int64_t *__restrict__ partial, int nchunks)
{
    __shared__ long long ws[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t i0 = (int64_t)blockIdx.x * P0_CHUNK + (int64_t)threadIdx.x * 4;      /* four consecutive values per thread */
    long long x[4], tot = 0;
    #pragma unroll
    for (int k = 0; k < 4; k++) { x[k] = i0 + k < n ? v[i0 + k] : 0; tot += x[k]; }
    long long inc = tot;
    #pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const long long o = __shfl_up(inc, d); if (lane >= d) inc += o; }
    if (lane == 63) ws[w] = inc;
    __syncthreads();
    long long run = partial[blockIdx.x] + inc - tot;
    for (int j = 0; j < w; j++) run += ws[j];
    #pragma unroll
    for (int k = 0; k < 4; k++) { if (i0 + k < n) v[i0 + k] = run; run += x[k]; }
    if (blockIdx.x == (unsigned)(nchunks - 1) && threadIdx.x == 255) v[n] = partial[blockIdx.x] + ws[0] + ws[1] + ws[2] + (w == 3 ? inc : 0);
}


/* The offset arrays from BAM's own per-read fields (crumble_batch.l_qseq / .n_cigar): a read's 32-bit l_seq and 16-bit
 * n_cigar_op cross PCIe (6 bytes per read) instead of two 64-bit offsets (16), and the exclusive prefix sums
 * seq_off[n + 1] / cig_off[n + 1] every later kernel reads are made here.  Three launches: chunk sums of both
 * lengths, k_i64scan_partials over the two partial arrays (grid 2), the offsets.  A negative l_qseq gives offsets
 * that run backwards, which k_p0_classify reports like any other bad offset table. */
__global__ void __launch_bounds__(256)
k_p0_len_reduce(int64_t n, const int32_t *__restrict__ lq, const uint16_t *__restrict__ ncg, int64_t *__restrict__ partial, int nchunks)
{
    __shared__ long long ws[8];
    const int64_t b = (int64_t)blockIdx.x * P0_CHUNK;
    long long aq = 0, ac = 0;
    #pragma unroll
    for (int k = 0; k < P0_CHUNK / 256; k++) { const int64_t i = b + k * 256 + threadIdx.x; if (i < n) { aq += lq[i]; ac += ncg[i]; } }
    #pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { aq += __shfl_xor(aq, d); ac += __shfl_xor(ac, d); }
    if ((threadIdx.x & 63) == 0) { ws[threadIdx.x >> 6] = aq; ws[4 + (threadIdx.x >> 6)] = ac; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
        partial[(size_t)nchunks + 1 + blockIdx.x] = ws[4] + ws[5] + ws[6] + ws[7];
    }
}
__global__ void __launch_bounds__(256)
k_p0_len_apply(int64_t n, const int32_t *__restrict__ lq, const uint16_t *__restrict__ ncg, const int64_t *__restrict__ partial, int nchunks,
               int64_t *__restrict__ seq_off, int64_t *__restrict__ cig_off)
{
    __shared__ long long ws[8];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t i0 = (int64_t)blockIdx.x * P0_CHUNK + (int64_t)threadIdx.x * 4;      /* four consecutive reads per thread */
    long long xq[4], xc[4], tq = 0, tc = 0;
    #pragma unroll
    for (int k = 0; k < 4; k++) { xq[k] = i0 + k < n ? lq[i0 + k] : 0; xc[k] = i0 + k < n ? ncg[i0 + k] : 0; tq += xq[k]; tc += xc[k]; }
    long long iq = tq, ic = tc;
    #pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const long long oq = __shfl_up(iq, d), oc = __shfl_up(ic, d); if (lane >= d) { iq += oq; ic += oc; } }
    if (lane == 63) { ws[w] = iq; ws[4 + w] = ic; }
    __syncthreads();
    long long rq = partial[blockIdx.x] + iq - tq, rc = partial[(size_t)nchunks + 1 + blockIdx.x] + ic - tc;
    for (int j = 0; j < w; j++) { rq += ws[j]; rc += ws[4 + j]; }
    #pragma unroll
    for (int k = 0; k < 4; k++) { if (i0 + k < n) { seq_off[i0 + k] = rq; cig_off[i0 + k] = rc; } rq += xq[k]; rc += xc[k]; }
    if (blockIdx.x == (unsigned)(nchunks - 1) && threadIdx.x == 255) {
        seq_off[n] = partial[blockIdx.x] + ws[0] + ws[1] + ws[2] + iq;
        cig_off[n] = partial[(size_t)nchunks + 1 + blockIdx.x] + ws[4] + ws[5] + ws[6] + ic;
    }
}

/* ---- k_p0_quals: the coded qualities of an upload (crumble_batch.qualsp: 2 or 4 bits per base into a table of up to
 * sixteen values) -> one byte per base, the form every other kernel reads.  Sixteen bases per thread: 4 or 8 bytes in, 16 out
 * (1.25-1.5 B per base of HBM traffic for 0.5-0.75 B per base less over PCIe).  The table lives in four registers and a
 * v_perm_b32 whose selector IS the codes picks four values at once. */
template <int BITS>
__global__ void __launch_bounds__(256)
k_p0_quals(int64_t nb, const uint8_t *__restrict__ qp, uint32_t l0, uint32_t l1, uint32_t l2, uint32_t l3, uint8_t *__restrict__ quals)
{
    const int64_t nch = (nb + 15) >> 4;
    for (int64_t ch = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; ch < nch; ch += (int64_t)gridDim.x * blockDim.x) {
        uint32_t o[4];
        if (BITS == 2) {
            const uint32_t w = *(const uint32_t *)(qp + ch * 4);             /* (the array carries slack) */
            #pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t t = (w >> (8 * k)) & 255u;                     /* four codes: one per byte, 0..3 = bytes of l0 */
                o[k] = kc2_perm(0u, l0, (t | (t << 6) | (t << 12) | (t << 18)) & 0x03030303u);
            }
        } else {
            const uint2 w2 = *(const uint2 *)(qp + ch * 8);
            #pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t t = ((k < 2 ? w2.x : w2.y) >> (16 * (k & 1))) & 0xffffu;      /* four codes of four bits */
                const uint32_t x = (t | (t << 8)) & 0x00ff00ffu;
                const uint32_t c = (x | (x << 4)) & 0x0f0f0f0fu;                             /* one per byte */
                const uint32_t sel = c & 0x07070707u;
                const uint32_t lo = kc2_perm(l1, l0, sel), hi = kc2_perm(l3, l2, sel);
                o[k] = kc2_perm(hi, lo, ((c >> 1) & 0x04040404u) | 0x03020100u);           /* code >= 8: the upper half of the table */
            }
        }
        uint4 v; v.x = o[0]; v.y = o[1]; v.z = o[2]; v.w = o[3];
        *(uint4 *)(quals + ch * 16) = v;                                      /* (up to 15 bytes of slack behind the last base) */
    }
}

#endif
