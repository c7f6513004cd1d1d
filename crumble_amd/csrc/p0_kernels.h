/*
 * p0_kernels.h -- "P0 expand" on the device: everything the column engine needs beyond the raw
 * BAM-shaped arrays of a crumble_batch is derived here, so crumble_hip_upload() is a set of
 * host->device copies plus a few launches (no per-read host loops).
 *
 *   k_p0_classify   per read: pileup_callback's routing (snp_score.c:1130-1149) + htslib's default
 *                   flag filter, reference length / end, number of reference-consuming CIGAR ops,
 *                   input validation (the reference asserts instead: snp_score.c:1075)
 *   k_p0_basequal   input qualities must be < 128; consensus class of every base into its high nibble
 *   k_p0_sorted     coordinate order of the reads that enter the pileup (snp_score.c:1075)
 *   k_p0scan_*      exclusive prefix sums -> offsets of each read's ops / base->column entries
 *   k_p0_fill       per read: Seg + SegRec tables with htslib's resolve_cigar2 `indel` peek
 *                   (SURVEY.md App. C), ReadRec, base->column table and the list of bases that
 *                   D/N columns report (replay)
 *   k_p0_keys, k_pmax_*, k_p0_tiles   candidate ranges per 64-column tile (reads and ops)
 *   k_p0_blkread    owning read of every 64-base block of the flat base array (k_p0_chunkread: of every 16-base chunk)
 */
#ifndef CRUMBLE_P0_KERNELS_H
#define CRUMBLE_P0_KERNELS_H

#include "dev_common.h"

struct P0Cnt { int64_t nseg, ncix, nslow; };

#define P0_ERR_NEG      1
#define P0_ERR_QLEN     2
#define P0_ERR_COORD    3
#define P0_ERR_UNSORTED 4
#define P0_ERR_QUAL     5

__device__ __forceinline__ bool p0_is_mdnx(int op) { return op == 0 || op == 2 || op == 3 || op == 7 || op == 8; }

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

__device__ __forceinline__ void p0_classify_read(int64_t r, int tid, const int32_t *__restrict__ pos, const uint16_t *__restrict__ flag,
                              const int64_t *__restrict__ seq_off, const int64_t *__restrict__ cig_off,
                              const uint32_t *__restrict__ cigar, uint8_t *__restrict__ inplp,
                              int32_t *__restrict__ end, P0Cnt *__restrict__ cnt, int *err, int &mn, int &mx,
                              int64_t nb, int64_t nc)
{
    const int64_t co0 = cig_off[r], co1 = cig_off[r + 1], so0 = seq_off[r], so1 = seq_off[r + 1];
    const int64_t ncg64 = co1 - co0, lq = so1 - so0;
    P0Cnt c = {0, 0, 0};
    inplp[r] = 0; end[r] = pos[r];
    /* offsets must run 0 <= off[r] <= off[r+1] <= total before anything is dereferenced through them */
    if (lq < 0 || ncg64 < 0 || lq > 0x3fffffff || co0 < 0 || co1 > nc || so0 < 0 || so1 > nb ||
        (r == 0 && (co0 != 0 || so0 != 0))) { atomicMax(err, P0_ERR_NEG); cnt[r] = c; return; }
    const uint32_t *cig = cigar + co0;
    const int ncg = (int)ncg64;
    /* pileup_callback routing (snp:1130-1149) + htslib's default filter */
    if (tid < 0 || (flag[r] & (4 | 256 | 512 | 1024))) { cnt[r] = c; return; }
    int64_t x = 0, y = 0;
    int nseg = 0, ndn = 0, first_type = -1, first_ia = 0;
    for (int k = 0; k < ncg; k++) {
        const int op = cig[k] & 15, l = (int)(cig[k] >> 4);
        if (p0_is_mdnx(op)) {
            if (l > 0) {
                if (!nseg) { first_type = (op == 2) ? 1 : (op == 3) ? 2 : 0; first_ia = p0_indel_after(cig, ncg, k); }
                nseg++;
                if ((op == 2 || op == 3) && y < lq) ndn++;
            }
            x += l;
            if (op != 2 && op != 3) y += l;
        } else if (op == 1 || op == 4) y += l;
    }
    if (!nseg || x <= 0) { cnt[r] = c; return; }                      /* no reference-consuming op */
    if (y != lq && lq != 0) { atomicMax(err, P0_ERR_QLEN); cnt[r] = c; return; }
    if (pos[r] < 0 || (int64_t)pos[r] + x >= (int64_t)INT_MAX - 64) { atomicMax(err, P0_ERR_COORD); cnt[r] = c; return; }
    inplp[r] = 1;
    end[r] = pos[r] + (int32_t)x;
    const bool simple = nseg == 1 && first_type == 0 && first_ia == 0 && lq > 0;
    c.nseg = nseg; c.ncix = simple ? 0 : ((lq + CIX_UNIT - 1) & ~(int64_t)(CIX_UNIT - 1)); c.nslow = simple ? 0 : ndn;
    cnt[r] = c;
    mn = pos[r]; mx = end[r];                              /* extent of the pileup */
}

__global__ void __launch_bounds__(1024)
k_p0_classify(int64_t n, int tid, const int32_t *__restrict__ pos, const uint16_t *__restrict__ flag,
                              const int64_t *__restrict__ seq_off, const int64_t *__restrict__ cig_off,
                              const uint32_t *__restrict__ cigar, uint8_t *__restrict__ inplp,
                              int32_t *__restrict__ end, P0Cnt *__restrict__ cnt, int *err, int *minmax,
                              int64_t nb, int64_t nc)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int mn = INT_MAX, mx = INT_MIN;
    if (r < n) p0_classify_read(r, tid, pos, flag, seq_off, cig_off, cigar, inplp, end, cnt, err, mn, mx, nb, nc);
    /* one atomic pair per 1024-thread workgroup (an atomic pair per read, 12.6 M on two words, was most of this
     * kernel's time; one per wave still left 4.4 ms of it: a hot word takes ~90 atomics per microsecond) */
    __shared__ int wmn[16], wmx[16];
    #pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const int a = __shfl_xor(mn, d), b = __shfl_xor(mx, d);
        mn = a < mn ? a : mn; mx = b > mx ? b : mx;
    }
    if ((threadIdx.x & 63) == 0) { wmn[threadIdx.x >> 6] = mn; wmx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); w++) { mn = wmn[w] < mn ? wmn[w] : mn; mx = wmx[w] > mx ? wmx[w] : mx; }
        if (mn != INT_MAX) atomicMin(&minmax[0], mn);
        if (mx != INT_MIN) atomicMax(&minmax[1], mx);
    }
}

__global__ void k_p0_sorted(int64_t n, const int32_t *__restrict__ pos, const uint8_t *__restrict__ inplp, int *err)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n || !inplp[r]) return;
    int64_t q = r - 1;
    while (q >= 0 && !inplp[q]) q--;
    if (q >= 0 && pos[r] < pos[q]) atomicMax(err, P0_ERR_UNSORTED);   /* snp:1075 */
}

/* One pass over the flat base / quality arrays: input qualities must be < 128, and every base
 * byte gets its consensus class in the high nibble next to the nt16 code in the low one (what the reference
 * derives per pileup entry, snp:603-609: A C G T -> 0..3, anything else -> 5 "N"; 4 is '*'), so that k_columns2
 * stages classes with a shift instead of a per-byte decode.  Every reader of `bases` masks with 15. */
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
/* (do_bases == 0: the base bytes already carry their class, k_p0_unpack4 wrote them) */
__global__ void k_p0_basequal(int64_t nb, uint8_t *__restrict__ bases, const uint8_t *__restrict__ quals, int *err, int do_bases)
{
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    if (i >= nb) return;
    bool bad = false;
    if (i + 16 <= nb) {
        const uint4 v = *(const uint4 *)(quals + i);
        bad = ((v.x | v.y | v.z | v.w) & 0x80808080u) != 0;
        if (do_bases) {
            uint4 b = *(const uint4 *)(bases + i);
            b.x = p0_class4(b.x); b.y = p0_class4(b.y); b.z = p0_class4(b.z); b.w = p0_class4(b.w);
            *(uint4 *)(bases + i) = b;
        }
    } else {
        for (int64_t k = i; k < nb; k++) {
            bad = bad || (quals[k] & 0x80);
            if (do_bases) bases[k] = (uint8_t)(p0_class4(bases[k]) & 0xffu);
        }
    }
    if (bad) atomicMax(err, P0_ERR_QUAL);
}

/* crumble_batch.bases4 (BAM's two codes per byte, the earlier base in the high nibble; read i from byte
 * (seq_off[i] >> 1) + i) -> one byte per base with the consensus class in its high nibble, i.e. what k_p0_basequal
 * leaves.  One lane per 16 output bytes (its owner read from blk_read[], so every lane is independent and the store is
 * one aligned 16-byte vector): 8 or 9 packed bytes in (three dword loads that need no alignment), the nibbles
 * interleaved by shifts; a chunk that straddles reads goes byte by byte. */
typedef uint32_t __attribute__((aligned(1))) p0_u32u;
#define P0_ILV_LO(X, Y) (((X) & 0xffu) | (((Y) & 0xffu) << 8) | (((X) & 0xff00u) << 8) | (((Y) & 0xff00u) << 16))      /* x0 y0 x1 y1 */
#define P0_ILV_HI(X, Y) ((((X) >> 16) & 0xffu) | ((((Y) >> 16) & 0xffu) << 8) | ((((X) >> 16) & 0xff00u) << 8) | ((((Y) >> 16) & 0xff00u) << 16))   /* x2 y2 x3 y3 */
__global__ void __launch_bounds__(256)
k_p0_unpack4(int64_t nb, int64_t n, const int64_t *__restrict__ seq_off, const int32_t *__restrict__ blk_read,
             const uint8_t *__restrict__ b4, uint8_t *__restrict__ bases)
{
    const int64_t g = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    if (g >= nb) return;
    int64_t r = blk_read[g >> 6];
    while (r + 1 < n && seq_off[r + 1] <= g) r++;
    /* the chunk's bytes belong to one read, or -- at a read boundary -- to two or more: the sixteen bytes are made as if
     * the whole chunk lay in the read at hand (x may be negative: the bytes before the read are masked away) and merged */
    const int64_t gend = g + 16 < nb ? g + 16 : nb;
    uint32_t ow[4] = {0, 0, 0, 0};
    for (;; r++) {
        const int64_t so = seq_off[r], e = seq_off[r + 1];
        const int lo = so > g ? (int)(so - g) : 0, hi = e < gend ? (int)(e - g) : (int)(gend - g);
        if (lo < hi) {
            const int64_t x = g - so;                             /* (>> 1 floors, & 1 is the parity, also when negative) */
            const int64_t sb = (so >> 1) + r + (x >> 1);
            uint32_t d0, d1, d2;
            if (sb >= 0) { const uint8_t *src = b4 + sb; d0 = *(const p0_u32u *)src; d1 = *(const p0_u32u *)(src + 4); d2 = *(const p0_u32u *)(src + 8); }
            else {                                                /* the very first read of the batch, chunk bytes before it */
                uint32_t t3[3] = {0, 0, 0};
                for (int k = 0; k < 12; k++) if (sb + k >= 0) t3[k >> 2] |= (uint32_t)b4[sb + k] << (8 * (k & 3));
                d0 = t3[0]; d1 = t3[1]; d2 = t3[2];
            }
            const uint32_t h0 = (d0 >> 4) & 0x0f0f0f0fu, l0 = d0 & 0x0f0f0f0fu, h1 = (d1 >> 4) & 0x0f0f0f0fu, l1 = d1 & 0x0f0f0f0fu;
            uint32_t xa, ya, xb, yb;                              /* output byte 2j = xa/xb[j], 2j + 1 = ya/yb[j] */
            if (!(x & 1)) { xa = h0; ya = l0; xb = h1; yb = l1; }
            else {                                                /* base 2j is the low nibble of byte j, base 2j + 1 the high one of byte j + 1 */
                const uint32_t h2 = (d2 >> 4) & 0x0f0f0f0fu;
                xa = l0; ya = (h0 >> 8) | (h1 << 24); xb = l1; yb = (h1 >> 8) | (h2 << 24);
            }
            const uint32_t v[4] = {p0_class4(P0_ILV_LO(xa, ya)), p0_class4(P0_ILV_HI(xa, ya)), p0_class4(P0_ILV_LO(xb, yb)), p0_class4(P0_ILV_HI(xb, yb))};
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
#undef P0_ILV_LO
#undef P0_ILV_HI

/* chunked exclusive scan of P0Cnt over reads; totals -> partial[nchunks] */
#define P0_CHUNK 1024
__device__ __forceinline__ void p0cnt_add(P0Cnt &a, const P0Cnt &b) { a.nseg += b.nseg; a.ncix += b.ncix; a.nslow += b.nslow; }
__global__ void k_p0scan_reduce(int64_t n, const P0Cnt *__restrict__ cnt, P0Cnt *__restrict__ partial, int nchunks)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nchunks) return;
    P0Cnt acc = {0, 0, 0};
    const int64_t b = (int64_t)t * P0_CHUNK, e = b + P0_CHUNK < n ? b + P0_CHUNK : n;
    for (int64_t i = b; i < e; i++) p0cnt_add(acc, cnt[i]);
    partial[t] = acc;
}
/* exclusive scan of the chunk sums by ONE 1024-thread workgroup: a contiguous slice per thread, the thread sums
 * scanned through LDS (it was one thread walking 12 000 chunks: 1.7 ms per upload) */
__global__ void __launch_bounds__(1024)
k_p0scan_partials(P0Cnt *partial, int nchunks)
{
    __shared__ P0Cnt sh[1024];
    const int t = threadIdx.x, nt = blockDim.x;
    const int per = (nchunks + nt - 1) / nt;
    const int b = t * per, e = b + per < nchunks ? b + per : nchunks;
    P0Cnt acc = {0, 0, 0};
    for (int i = b; i < e; i++) p0cnt_add(acc, partial[i]);
    sh[t] = acc;
    __syncthreads();
    for (int d = 1; d < nt; d <<= 1) {                    /* Hillis-Steele inclusive scan of the thread sums */
        P0Cnt v = {0, 0, 0};
        if (t >= d) v = sh[t - d];
        __syncthreads();
        p0cnt_add(sh[t], v);
        __syncthreads();
    }
    P0Cnt run = {0, 0, 0};
    if (t > 0) run = sh[t - 1];
    for (int i = b; i < e; i++) { const P0Cnt v = partial[i]; partial[i] = run; p0cnt_add(run, v); }
    if (t == nt - 1) partial[nchunks] = sh[nt - 1];
}
__global__ void k_p0scan_apply(int64_t n, P0Cnt *__restrict__ cnt, const P0Cnt *__restrict__ partial, int nchunks)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nchunks) return;
    P0Cnt run = partial[t];
    const int64_t b = (int64_t)t * P0_CHUNK, e = b + P0_CHUNK < n ? b + P0_CHUNK : n;
    for (int64_t i = b; i < e; i++) { const P0Cnt v = cnt[i]; cnt[i] = run; p0cnt_add(run, v); }
}

__global__ void k_p0_fill(int64_t n, int p0, const int32_t *__restrict__ pos, const uint8_t *__restrict__ mapq,
                          const int64_t *__restrict__ seq_off, const int64_t *__restrict__ cig_off,
                          const uint32_t *__restrict__ cigar, const uint8_t *__restrict__ inplp,
                          const int32_t *__restrict__ end, const P0Cnt *__restrict__ off, const P0Cnt *__restrict__ total,
                          Seg *__restrict__ segs, SegRec *__restrict__ sr, int32_t *__restrict__ sr_ppos,
                          int32_t *__restrict__ sr_pend, ReadRec *__restrict__ rr, int64_t *__restrict__ seg_off,
                          int32_t *__restrict__ cix, int64_t *__restrict__ slow_g, int32_t *__restrict__ slow_r,
                          int *n_slow, const uint8_t *__restrict__ mq_slot, int want_cix)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const P0Cnt o = off[r];
    seg_off[r] = o.nseg;
    if (r == n - 1) seg_off[n] = total->nseg;
    const int64_t so = seq_off[r];
    const int lq = (int)(seq_off[r + 1] - so);
    ReadRec x;
    x.pos = pos[r]; x.end = end[r]; x.so = so; x.lq = lq; x.q0 = 0; x.info = mapq[r]; x.cix = 0;
    if (!inplp[r]) { rr[r] = x; return; }
    const uint32_t *cig = cigar + cig_off[r];
    const int ncg = (int)(cig_off[r + 1] - cig_off[r]);
    /* pass 1: the op table (Seg) and the self-contained records (SegRec) */
    int64_t si = o.nseg;
    const int64_t s_first = si;
    int xr = 0, y = 0, nseg = 0;
    for (int k = 0; k < ncg; k++) {
        const int op = cig[k] & 15, l = (int)(cig[k] >> 4);
        if (p0_is_mdnx(op)) {
            if (l > 0) {
                const uint32_t ty = (op == 2) ? SEG_D : (op == 3) ? SEG_N : SEG_M;
                Seg s;
                s.roff = xr; s.qstart = y; s.len_type = (uint32_t)l | (ty << 30);
                s.indel_after = p0_indel_after(cig, ncg, k);
                segs[si] = s;
                SegRec q;
                q.pos = x.pos + xr; q.end = q.pos + l; q.so = so; q.lq = lq;
                q.qoff = (ty == SEG_M) ? y - q.pos : y;
                q.indel_after = s.indel_after;
                q.info = (uint32_t)mapq[r] | (ty << 8) | ((uint32_t)mq_slot[mapq[r]] << 16);
                if (si == s_first) { q.info |= SR_FIRST; if (y > 0) q.info |= SR_HEADCLIP; x.q0 = y; }
                sr[si] = q; sr_ppos[si] = x.pos; sr_pend[si] = x.end;
                si++; nseg++;
            }
            xr += l;
            if (op != 2 && op != 3) y += l;
        } else if (op == 1 || op == 4) y += l;
    }
    {   /* the last op carries the tail flags */
        SegRec q = sr[si - 1];
        const Seg sl = segs[si - 1];
        const int qt = (seg_type(sl) == SEG_M) ? sl.qstart + seg_len(sl) - 1 : sl.qstart;
        q.info |= SR_LAST;
        if (qt + 1 < lq) q.info |= SR_TAILCLIP;
        sr[si - 1] = q;
    }
    const Seg s0 = segs[s_first];
    const bool simple = nseg == 1 && seg_type(s0) == SEG_M && s0.indel_after == 0 && lq > 0;
    if (simple) { x.info |= RR_SIMPLE; rr[r] = x; return; }
    /* pass 2: base -> column table of a general CIGAR (replay).  Long-read batches that go through the packed replay
     * (territories, k_p0_vrecs) never look a base up: only the list of bases that D/N columns report is made -- a
     * base is reported by a run of consecutive D/N ops, so equal qstart of consecutive ops is the duplicate test */
    x.cix = (uint32_t)(o.ncix / CIX_UNIT);
    rr[r] = x;
    if (!want_cix) {
        int last = -1;
        for (int64_t k = s_first; k < si; k++) {
            const Seg sg = segs[k];
            if (seg_type(sg) != SEG_M && sg.qstart < lq && sg.qstart != last) {
                last = sg.qstart;
                const int w = atomicAdd(n_slow, 1);
                slow_g[w] = so + sg.qstart; slow_r[w] = (int32_t)r;
            }
        }
        return;
    }
    int32_t *cx = cix + o.ncix;
    for (int i = 0; i < lq; i++) cx[i] = -1;
    for (int64_t k = s_first; k < si; k++) {
        const Seg sg = segs[k];
        if (seg_type(sg) != SEG_M) continue;
        for (int i = 0; i < seg_len(sg) && sg.qstart + i < lq; i++) cx[sg.qstart + i] = (x.pos + sg.roff + i) - p0;
    }
    for (int64_t k = s_first; k < si; k++) {               /* bases that D/N columns report */
        const Seg sg = segs[k];
        if (seg_type(sg) != SEG_M && sg.qstart < lq && cx[sg.qstart] != -2) {
            cx[sg.qstart] = -2;
            const int w = atomicAdd(n_slow, 1);
            slow_g[w] = so + sg.qstart; slow_r[w] = (int32_t)r;
        }
    }
}

/* monotone search keys for the per-tile candidate ranges of reads */
__global__ void k_p0_keys(int64_t n, const int32_t *__restrict__ pos, const int32_t *__restrict__ end,
                          const uint8_t *__restrict__ inplp, int32_t *__restrict__ kpos, int32_t *__restrict__ kend)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    kpos[r] = inplp[r] ? pos[r] : INT_MIN;
    kend[r] = inplp[r] ? end[r] : INT_MIN;
}

/* For tile t = columns [c0, c0+63]: hi = first index whose (prefix-max) start key > c0+63,
 * lo = min(hi, first index whose (prefix-max) end key > c0): exactly the two-pointer walk over
 * position-sorted reads (entries between lo and hi that ended earlier are skipped by the kernels). */
__global__ void k_p0_tiles(int n_tiles, int p0, int64_t n, const int32_t *__restrict__ kpos,
                           const int32_t *__restrict__ kend, int32_t *__restrict__ tlo, int32_t *__restrict__ thi)
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
}

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

/* keep-BED cursor per tile (k_columns2): the first region whose prefix-max of `end` reaches the tile's first column --
 * every column's own cursor (bed_preserve, kernels.h) is at or behind it.  Once per batch instead of a search per tile
 * in every run (three dependent rounds of loads in the tile's epilogue: 9 % of k_columns2 on config 4). */
__global__ void k_p0_tilebed(int n_tiles, int p0, int n_bed, const int32_t *__restrict__ bed_pmax, int32_t *__restrict__ tbed)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    const int64_t c0 = (int64_t)p0 + (int64_t)t * 64;
    int lo = 0, hi = n_bed;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if ((int64_t)bed_pmax[mid] < c0) lo = mid + 1; else hi = mid; }
    tbed[t] = lo;
}

__global__ void k_p0_blkread(int64_t nblocks, int64_t n, const int64_t *__restrict__ seq_off, int32_t *__restrict__ blk_read)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    const int64_t g0 = b * 64;
    int64_t lo = 0, hi = n;                                /* first r with seq_off[r + 1] > g0 */
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (seq_off[mid + 1] <= g0) lo = mid + 1; else hi = mid; }
    blk_read[b] = (int32_t)(lo < n ? lo : n - 1);
}

/* the same per 16-base chunk, from the block table (short-read batches: k_replay_fast starts its chunk
 * without walking forward from the block's read -- one dependent load less for every wave) */
__global__ void k_p0_chunkread(int64_t nchunks, int64_t n, const int64_t *__restrict__ seq_off,
                               const int32_t *__restrict__ blk_read, int32_t *__restrict__ chunk_read)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchunks) return;
    const int64_t g0 = c * 16;
    int64_t r = blk_read[c >> 2];
    while (r + 1 < n && seq_off[r + 1] <= g0) r++;
    chunk_read[c] = (int32_t)r;
}

/* The same two tables for short-read batches, by scatter: read r owns the blocks / chunks whose first base lies in
 * [seq_off[r], seq_off[r+1]) -- a couple of blocks and ~10 chunks per 150-base read, instead of a 24-step binary
 * search per block (0.64 + 0.44 ms per upload) */
__global__ void k_p0_owner_scatter(int64_t n, int64_t nb, const int64_t *__restrict__ seq_off,
                                   int32_t *__restrict__ blk_read, int32_t *__restrict__ chunk_read)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int64_t a = seq_off[r], b = seq_off[r + 1];
    if (b <= a) return;
    for (int64_t k = (a + 63) >> 6; k * 64 < b; k++) blk_read[k] = (int32_t)r;
    for (int64_t k = (a + 15) >> 4; k * 16 < b; k++) chunk_read[k] = (int32_t)r;
    if (b == nb) {                                         /* the entries past the last base (readers clamp) */
        blk_read[(nb + 63) >> 6] = (int32_t)r;
        chunk_read[(nb + 15) >> 4] = (int32_t)r;
    }
}

/* k_replay_reads' work list: the reads with a general CIGAR and bases, in any order.  One atomic per 1024-thread
 * workgroup (ballot + prefix popcount inside the waves, the wave totals through LDS): an atomic per read or per wave
 * on the one counter costs more than the rest of the upload's kernels together. */
__global__ void __launch_bounds__(1024)
k_p0_genlist(int64_t n, const ReadRec *__restrict__ rr, int32_t *__restrict__ list, int *__restrict__ count)
{
    __shared__ int wcnt[16], wbase[16];
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    bool gen = false;
    if (r < n) { const ReadRec x = rr[r]; gen = x.lq > 0 && x.end > x.pos && !(x.info & RR_SIMPLE); }
    const unsigned long long m = __ballot(gen);
    if (lane == 0) wcnt[w] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int k = 0; k < (int)(blockDim.x >> 6); k++) { wbase[k] = tot; tot += wcnt[k]; }
        const int base = tot ? atomicAdd(count, tot) : 0;
        for (int k = 0; k < (int)(blockDim.x >> 6); k++) wbase[k] += base;
    }
    __syncthreads();
    if (gen) list[wbase[w] + __popcll(m & ((1ull << lane) - 1ull))] = (int32_t)r;
}

/* Long-read batches: the territories of every read (VRec, dev_common.h) and the chunk -> territory table, by scatter
 * like k_p0_owner_scatter.  One thread per read. */
__global__ void k_p0_vrecs(int64_t n, int64_t nb, int p0, int min_mqual, const ReadRec *__restrict__ rr,
                           const int64_t *__restrict__ seg_off, const Seg *__restrict__ segs,
                           VRec *__restrict__ vr, int32_t *__restrict__ chunk_vr)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const ReadRec x = rr[r];
    const int64_t s0 = seg_off[r], base = s0 + r;
    const int ns = (int)(seg_off[r + 1] - s0);
    const int lq = x.lq;
    const int hcol = (x.end > x.pos && (int)(x.info & 255u) <= min_mqual) ? x.pos - p0 : -1;
    for (int j = -1; j < ns; j++) {
        int qs = 0, qn = lq;
        VRec v;
        v.pos = v.end = 0; v.q0 = 0; v.hcol = hcol; v.read = (int32_t)r;
        if (j >= 0) {
            const Seg sg = segs[s0 + j];
            qs = sg.qstart < lq ? sg.qstart : lq;
            if (seg_type(sg) == SEG_M) {
                const int len = seg_len(sg) < lq - qs ? seg_len(sg) : lq - qs;     /* fence: query bases that exist */
                v.pos = x.pos + sg.roff; v.end = v.pos + (len > 0 ? len : 0);
            }
        }
        if (j + 1 < ns) { const int q2 = segs[s0 + j + 1].qstart; qn = q2 < lq ? q2 : lq; }
        if (qn < qs) qn = qs;
        v.so = x.so + qs; v.lq = qn - qs;
        const int64_t slot = base + 1 + j;
        vr[slot] = v;
        const int64_t a = v.so, b = v.so + v.lq;
        for (int64_t k = (a + 15) >> 4; k * 16 < b; k++) chunk_vr[k] = (int32_t)slot;
        if (b > a && b == nb) chunk_vr[(nb + 15) >> 4] = (int32_t)slot;           /* the entry past the last base */
    }
}

/* Work split of k_columns: an estimated cost per 64-column tile (its candidate records plus a constant
 * for the per-tile prologue/epilogue), prefix-summed, so that every wave can take a CONTIGUOUS run of
 * tiles of equal total cost -- no atomics, adjacent tiles (which share their reads) stay in one wave. */
#define TILE_COST_SPAN (1ll << 56)               /* marks a tile whose reads span too much of the base array */
__global__ void k_tile_cost(int n_tiles, const int32_t *__restrict__ slo, const int32_t *__restrict__ shi,
                            const SegRec *__restrict__ sr, const int64_t *__restrict__ hit_off, int64_t *__restrict__ cost)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    const int w = shi[t] - slo[t];
    /* long reads (hit_off): the records that reach the tile, not the candidates around them */
    int64_t c = hit_off ? (hit_off[t + 1] - hit_off[t]) + 2 : w + 2;
    /* k_columns2 addresses the bases of a tile's reads with 32-bit offsets from the first one */
    if (w > 0 && sr[shi[t] - 1].so - sr[slo[t]].so > 0xf0000000ll) c = TILE_COST_SPAN;
    cost[t] = c;
}
__global__ void k_i64scan_reduce(int64_t n, const int64_t *__restrict__ v, int64_t *__restrict__ partial, int nchunks)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nchunks) return;
    int64_t acc = 0;
    const int64_t b = (int64_t)t * P0_CHUNK, e = b + P0_CHUNK < n ? b + P0_CHUNK : n;
    for (int64_t i = b; i < e; i++) acc += v[i];
    partial[t] = acc;
}
__global__ void __launch_bounds__(1024)
k_i64scan_partials(int64_t *partial, int nchunks)
{
    __shared__ int64_t sh[1024];
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
__global__ void k_i64scan_apply(int64_t n, int64_t *__restrict__ v, const int64_t *__restrict__ partial, int nchunks)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nchunks) return;
    int64_t run = partial[t];
    const int64_t b = (int64_t)t * P0_CHUNK, e = b + P0_CHUNK < n ? b + P0_CHUNK : n;
    for (int64_t i = b; i < e; i++) { const int64_t x = v[i]; v[i] = run; run += x; }
    if (t == nchunks - 1) v[n] = run;
}

/* unit g of the cost prefix starts at tile unit_first[g] = first t with cum[t] >= g*KCOL_UNIT */
__global__ void k_unit_first(int n_units, int n_tiles, const int64_t *__restrict__ cum, int32_t *__restrict__ unit_first)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g > n_units) return;
    const int64_t c = (int64_t)g * KCOL_UNIT;
    int lo = 0, hi = n_tiles;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (cum[mid] < c) lo = mid + 1; else hi = mid; }
    unit_first[g] = g == n_units ? n_tiles : lo;
}

#endif
