/*
 * stream_kernels.h -- BAM records that stay on the device between inflate and deflate (crumble_stream, stream_host.h).
 *
 * The reference walks its records one sam_read1() at a time (pileup_callback, snp_score.c:1113-1153) and hands them back
 * through sam_write1() (flush_bam_list, snp_score.c:1081-1105).  Here the inflated BGZF payload -- the BAM record stream of
 * SAMv1 4.2: [block_size][refID pos l_read_name mapq bin n_cigar_op flag l_seq next_refID next_pos tlen][name][cigar][seq]
 * [qual][tags] -- lies in one device buffer, and four things are found or moved there:
 *
 *   where the records start.  A record's start is known only from its predecessor's block_size: a chain.  It is cut into
 *     64 KiB chunks: every chunk guesses the first record start inside it (k_st_guess: a wave tests 64 offsets at a time
 *     for a plausible header whose next two records are plausible too), walks its records from there (k_st_walk: a lane
 *     per chunk) and the chain is CHECKED afterwards -- chunk c's walk must end exactly where chunk c + 1 guessed
 *     (k_st_verify); a chunk inside one huge record has no start at all.  A failed check is reported, never papered over.
 *   where the contigs change (k_st_groups), the per-read fields and totals of a contig (k_st_fields),
 *   the engine's input arrays from the records (k_st_gather) and its output qualities back into them (k_st_scatter).
 */
#ifndef CRUMBLE_STREAM_KERNELS_H
#define CRUMBLE_STREAM_KERNELS_H

#define ST_CHUNK     65536
#define ST_MAXREC    (ST_CHUNK / 36 + 2)         /* records that can start inside one chunk (a record is at least 36 bytes) */
#define ST_NONE      (-1ll)

__device__ __forceinline__ uint32_t st_u32(const uint8_t *p) { return ((const GzU32u *)p)->v; }
__device__ __forceinline__ uint32_t st_u16(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }

/* is there a plausible record header at R[o]?  (n bytes in all, n_ref reference sequences)  Everything SAMv1 4.2 fixes:
 * the fixed part fits block_size, refID / next_refID name a reference or -1, pos / next_pos >= -1, the read name is at least
 * its NUL and ends in one, and name + cigar + seq + qual fit block_size. */
__device__ __forceinline__ bool st_plausible(const uint8_t *R, int64_t o, int64_t n, int n_ref, int64_t *next)
{
    if (o + 36 > n) return false;
    const uint32_t bs = st_u32(R + o);
    if (bs < 32u || bs > (1u << 29)) return false;
    const int32_t refid = (int32_t)st_u32(R + o + 4), pos = (int32_t)st_u32(R + o + 8);
    if (refid < -1 || refid >= n_ref || pos < -1) return false;
    const uint32_t lname = R[o + 12], ncig = st_u16(R + o + 16);
    const int32_t lseq = (int32_t)st_u32(R + o + 20), nref2 = (int32_t)st_u32(R + o + 24), npos = (int32_t)st_u32(R + o + 28);
    if (lname < 1u || lseq < 0 || nref2 < -1 || nref2 >= n_ref || npos < -1) return false;
    const uint64_t need = 32ull + lname + 4ull * ncig + ((uint64_t)lseq + 1) / 2 + (uint64_t)lseq;
    if (need > bs) return false;
    if (o + 36 + (int64_t)lname <= n && R[o + 36 + lname - 1] != 0) return false;
    if (lname > 1u && o + 37 <= n) { const uint8_t c0 = R[o + 36]; if (c0 < 33 || c0 > 126) return false; }     /* a printable name */
    *next = o + 4 + (int64_t)bs;
    return true;
}

/* per chunk c >= 1: the first offset in [start + c * ST_CHUNK, + ST_CHUNK) at which a plausible record begins whose two
 * successors (as far as the buffer goes) are plausible as well; ST_NONE when there is none */
__global__ void __launch_bounds__(64)
k_st_guess(const uint8_t *__restrict__ R, int64_t start, int64_t n, int n_ref, int n_chunks, int64_t *__restrict__ guess)
{
    const int c = blockIdx.x, lane = threadIdx.x & 63;
    if (c >= n_chunks) return;
    if (c == 0) { if (lane == 0) guess[0] = start; return; }
    const int64_t lo = start + (int64_t)c * ST_CHUNK, hi = lo + ST_CHUNK < n ? lo + ST_CHUNK : n;
    int64_t found = ST_NONE;
    for (int64_t b = lo; b < hi && found == ST_NONE; b += 64) {
        const int64_t o = b + lane;
        bool ok = false;
        if (o < hi) {
            int64_t o1, o2, o3;
            ok = st_plausible(R, o, n, n_ref, &o1);
            if (ok && o1 + 36 <= n) { ok = st_plausible(R, o1, n, n_ref, &o2); if (ok && o2 + 36 <= n) ok = st_plausible(R, o2, n, n_ref, &o3); }
        }
        const unsigned long long m = __ballot(ok);
        if (m) found = b + __builtin_ctzll(m);
    }
    if (lane == 0) guess[c] = found;
}

/* a lane per chunk: the records that start inside it, from its guess on; exit[c] = where its walk left off (the first
 * record start at or behind the chunk's end, or an incomplete record at the end of the buffer) */
__global__ void __launch_bounds__(64)
k_st_walk(const uint8_t *__restrict__ R, int64_t start, int64_t n, int n_chunks, const int64_t *__restrict__ guess,
          int64_t *__restrict__ slots, int32_t *__restrict__ count, int64_t *__restrict__ exitp)
{
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= n_chunks) return;
    int64_t o = guess[c];
    int cnt = 0;
    if (o != ST_NONE) {
        const int64_t hi = start + (int64_t)(c + 1) * ST_CHUNK;
        int64_t *sl = slots + (size_t)c * ST_MAXREC;
        while (o < hi && o + 4 <= n) {
            const int64_t nx = o + 4 + (int64_t)st_u32(R + o);
            if (nx > n || nx <= o) break;                       /* the record is not all there (or no record at all: the check below fails) */
            if (cnt < ST_MAXREC) sl[cnt] = o;
            cnt++;
            o = nx;
        }
    }
    count[c] = cnt; exitp[c] = o;
}

/* the chain, checked and REPAIRED: chunk c's first record start must be where the walk of the last chunk in front of it that
 * has one left off (if that lies inside chunk c; if it lies behind, chunk c lies inside a record and has none).  A guess can
 * be wrong -- with a few hundred reference sequences a record's own block_size passes for a refID four bytes early -- so:
 * k_st_check notes for every chunk whether its guess is what the chain says (want[c] = ST_OK) or what the chain says instead,
 * and k_st_fix replaces the guess of every inconsistent chunk WHOSE PREDECESSORS (the last chunk with a start and the ones
 * without between it and this one) ARE CONSISTENT -- a chunk behind a wrong one is inconsistent only because that one's walk
 * went astray or never happened: it is left alone and is consistent again once its predecessor is repaired.  Chunk 0 starts at a known record, so the first inconsistent chunk is always repaired; every round of walk /
 * check / fix removes at least one wrong guess and all isolated ones, and what stands when a round changes nothing is the
 * sequential walk's result.  bad: a chunk with more records than it can hold (cannot happen: 36 bytes each). */
#define ST_OK (-2ll)
__global__ void __launch_bounds__(256)
k_st_check(int n_chunks, int64_t start, const int64_t *__restrict__ guess, const int64_t *__restrict__ exitp, const int32_t *__restrict__ count,
           int64_t *__restrict__ want, int32_t *__restrict__ pred, int *bad)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n_chunks) return;
    if (count[c] > ST_MAXREC) atomicOr(bad, 4);
    if (c == 0) { want[0] = ST_OK; pred[0] = 0; return; }
    int p = c - 1;
    while (p > 0 && guess[p] == ST_NONE) p--;
    const int64_t e = exitp[p], lo = start + (int64_t)c * ST_CHUNK, hi = lo + ST_CHUNK;
    const int64_t w = e < hi ? (e >= lo ? e : ST_NONE) : ST_NONE;      /* (e < lo: chunk p's walk stopped at a record the buffer ends in) */
    want[c] = guess[c] == w ? ST_OK : w;
    pred[c] = p;
}
__global__ void __launch_bounds__(256)
k_st_fix(int n_chunks, int64_t *__restrict__ guess, const int64_t *__restrict__ want, const int32_t *__restrict__ pred, int *changed)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n_chunks || c == 0) return;
    if (want[c] == ST_OK) return;
    for (int q = pred[c]; q < c; q++) if (want[q] != ST_OK) return;      /* (the chunks without a start in between count: their "none" is a claim too) */
    guess[c] = want[c];
    atomicOr(changed, 1);
}
/* the walk's end must be the buffer's end or an incomplete record; *tail = where the unconsumed bytes start */
__global__ void k_st_tail(int n_chunks, const int64_t *__restrict__ guess, const int64_t *__restrict__ exitp, int64_t n, int64_t *tail, int *bad)
{
    int p = n_chunks - 1;
    while (p > 0 && guess[p] == ST_NONE) p--;
    const int64_t e = exitp[p];
    if (e > n || e < 0) atomicOr(bad, 2);
    *tail = e;
}

/* the chunks' record starts side by side: rec_off[off[c] + i] = slots[c][i] */
__global__ void __launch_bounds__(256)
k_st_compact(int n_chunks, const int64_t *__restrict__ slots, const int32_t *__restrict__ count, const int64_t *__restrict__ off,
             int64_t *__restrict__ rec_off)
{
    const int c = blockIdx.x;
    if (c >= n_chunks) return;
    const int n = count[c] < ST_MAXREC ? count[c] : ST_MAXREC;
    const int64_t *sl = slots + (size_t)c * ST_MAXREC;
    int64_t *d = rec_off + off[c];
    for (int i = threadIdx.x; i < n; i += 256) d[i] = sl[i];
}
__global__ void __launch_bounds__(1024)
k_st_offsets(int n, const int32_t *__restrict__ count, int64_t *__restrict__ off)
{   /* exclusive prefix sum of the chunks' counts; off[n] = the total */
    __shared__ long long part[1024];
    const int tid = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int lo = tid * per < n ? tid * per : n, hi = lo + per < n ? lo + per : n;
    long long s = 0;
    for (int i = lo; i < hi; i++) s += count[i];
    part[tid] = s;
    __syncthreads();
    if (tid == 0) { long long a = 0; for (int i = 0; i < 1024; i++) { const long long v = part[i]; part[i] = a; a += v; } off[n] = a; }
    __syncthreads();
    long long a = part[tid];
    for (int i = lo; i < hi; i++) { off[i] = a; a += count[i]; }
}

/* where the refID changes: (record index, refID of the run that starts there), unordered; every record is checked for a
 * plausible header on the way (a record that was only stepped over by block_size may still be garbage) */
__global__ void __launch_bounds__(256)
k_st_groups(const uint8_t *__restrict__ R, int64_t n_bytes, int n_ref, int64_t n_rec, const int64_t *__restrict__ rec_off,
            int64_t *__restrict__ list, int *__restrict__ n_list, int max_list, int *bad)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_rec) return;
    const int64_t o = rec_off[i];
    int64_t nx;
    if (!st_plausible(R, o, n_bytes, n_ref, &nx)) { atomicOr(bad, 8); return; }
    const int32_t tid = (int32_t)st_u32(R + o + 4);
    const int32_t prev = i ? (int32_t)st_u32(R + rec_off[i - 1] + 4) : tid - 1;
    if (tid != prev) {
        const int k = atomicAdd(n_list, 1);
        if (k < max_list) { list[2 * k] = i; list[2 * k + 1] = tid; }
    }
}

/* ---- one contig's records -> what the engine's expansion (p0_kernels.h) starts from ------------------------------ */
struct StTotals { unsigned long long n_bases, n_cigar, err_ppm, err_cnt; unsigned mq_hist[256]; unsigned unsupported; unsigned pad_; };

/* per read: pos mapq flag l_seq n_cigar_op and "qualities absent" (0xff, SAMv1 4.2.3: presented as 0, written back
 * untouched); the contig's totals, a sampled mapq histogram and error estimate (the column kernel's hints) */
__global__ void __launch_bounds__(256)
k_st_fields(const uint8_t *__restrict__ R, const int64_t *__restrict__ rec_off, int64_t n, int32_t *__restrict__ pos,
            uint8_t *__restrict__ mapq, uint16_t *__restrict__ flag, int32_t *__restrict__ lq, uint16_t *__restrict__ ncg,
            uint8_t *__restrict__ noq, StTotals *__restrict__ tot, const uint32_t *__restrict__ err_ppm, int64_t sample_step)
{
    __shared__ unsigned hist[256];
    __shared__ unsigned flags;
    __shared__ unsigned long long sums[4][4];                /* per wave: bases, cigar operations, error ppm, error samples */
    unsigned long long nb = 0, nc = 0, ep = 0, ec = 0;
    hist[threadIdx.x] = 0u;
    if (threadIdx.x == 0) flags = 0u;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const uint8_t *r = R + rec_off[i];
        const uint32_t lname = r[12], nco = st_u16(r + 16);
        const int32_t ls = (int32_t)st_u32(r + 20);
        pos[i] = (int32_t)st_u32(r + 8); mapq[i] = r[13]; flag[i] = (uint16_t)st_u16(r + 18); lq[i] = ls; ncg[i] = (uint16_t)nco;
        const uint8_t *cg = r + 36 + lname, *q = cg + 4 * nco + (ls + 1) / 2;
        const bool absent = ls > 0 && q[0] == 0xff;
        noq[i] = absent ? 1 : 0;
        nb += (unsigned long long)ls; nc += nco;
        /* a CIGAR of more than 65535 operations lives in the CG tag behind a kSmN placeholder (SAMv1 4.2.2): not taken here */
        if (nco == 2u && ls > 0) {
            const uint32_t c0 = st_u32(cg), c1 = st_u32(cg + 4);
            if ((c0 & 15u) == 4u && (int32_t)(c0 >> 4) == ls && (c1 & 15u) == 3u) atomicOr(&flags, 1u);
        }
        if (i % sample_step == 0) {
            atomicAdd(&hist[r[13]], 1u);
            if (ls > 0) { ep += err_ppm[absent ? 0 : (q[0] & 127)]; ec += 1ull; }
        }
    }
    /* the sums: wave, workgroup, then one atomic each per workgroup (a few hundred per launch: atomics on one address take
     * their turns in the L2); the histogram and the flag: one atomic per bin the workgroup touched */
    for (int d = 32; d > 0; d >>= 1) {
        nb += (unsigned long long)__shfl_xor((long long)nb, d); nc += (unsigned long long)__shfl_xor((long long)nc, d);
        ep += (unsigned long long)__shfl_xor((long long)ep, d); ec += (unsigned long long)__shfl_xor((long long)ec, d);
    }
    if ((threadIdx.x & 63) == 0) { const int w = threadIdx.x >> 6; sums[w][0] = nb; sums[w][1] = nc; sums[w][2] = ep; sums[w][3] = ec; }
    __syncthreads();
    if (threadIdx.x < 4) {
        const unsigned long long v = sums[0][threadIdx.x] + sums[1][threadIdx.x] + sums[2][threadIdx.x] + sums[3][threadIdx.x];
        if (v) atomicAdd(threadIdx.x == 0 ? &tot->n_bases : threadIdx.x == 1 ? &tot->n_cigar : threadIdx.x == 2 ? &tot->err_ppm : &tot->err_cnt, v);
    }
    if (hist[threadIdx.x]) atomicAdd(&tot->mq_hist[threadIdx.x], hist[threadIdx.x]);
    if (threadIdx.x == 0 && flags) atomicOr(&tot->unsupported, 1u);
}

/* n bytes by one wave, four per lane and step (both sides at any byte address: the device takes unaligned words), the last
 * one to three bytes singly */
struct __attribute__((packed, aligned(1))) StU32u { uint32_t v; };
__device__ __forceinline__ void st_copy_wave(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, int n, int lane)
{
    const int n4 = n >> 2;
    for (int j = lane; j < n4; j += 64) ((StU32u *)dst)[j].v = ((const StU32u *)src)[j].v;
    const int rest = n & 3;
    if (lane < rest) dst[4 * n4 + lane] = src[4 * n4 + lane];
}

/* a wave per read (sixteen reads per wave in turn): CIGAR words, the packed bases into the bases4 layout (read i at byte
 * (seq_off[i] >> 1) + i), the qualities (zeros where they are absent) */
__global__ void __launch_bounds__(256)
k_st_gather(const uint8_t *__restrict__ R, const int64_t *__restrict__ rec_off, int64_t n, const int64_t *__restrict__ seq_off,
            const int64_t *__restrict__ cig_off, const uint8_t *__restrict__ noq, uint32_t *__restrict__ cigar,
            uint8_t *__restrict__ bases4, uint8_t *__restrict__ quals)
{
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    for (int k = 0; k < 16; k++) {
        const int64_t i = w * 16 + k;
        if (i >= n) return;
        const uint8_t *r = R + rec_off[i];
        const uint32_t lname = r[12], nco = st_u16(r + 16);
        const int32_t ls = (int32_t)st_u32(r + 20);
        const uint8_t *cg = r + 36 + lname, *sq = cg + 4 * nco, *q = sq + (ls + 1) / 2;
        const int64_t so = seq_off[i], co = cig_off[i];
        for (uint32_t j = lane; j < nco; j += 64) cigar[co + j] = st_u32(cg + 4 * j);
        uint8_t *b4 = bases4 + (so >> 1) + i;
        st_copy_wave(b4, sq, (ls + 1) / 2, lane);
        if (noq[i] != 0) { for (int j = lane; j < ls; j += 64) quals[so + j] = 0; }
        else st_copy_wave(quals + so, q, ls, lane);
    }
}

/* the engine's output qualities back into the records (flush_bam_list's "replace the record's qualities",
 * snp_score.c:1081-1105); records without qualities stay as they are */
__global__ void __launch_bounds__(256)
k_st_scatter(uint8_t *__restrict__ R, const int64_t *__restrict__ rec_off, int64_t n, const int64_t *__restrict__ seq_off,
             const uint8_t *__restrict__ noq, const uint8_t *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    for (int k = 0; k < 16; k++) {
        const int64_t i = w * 16 + k;
        if (i >= n) return;
        if (noq[i]) continue;
        uint8_t *r = R + rec_off[i];
        const uint32_t lname = r[12], nco = st_u16(r + 16);
        const int32_t ls = (int32_t)st_u32(r + 20);
        uint8_t *q = r + 36 + lname + 4 * nco + (ls + 1) / 2;
        st_copy_wave(q, out + seq_off[i], ls, lane);
    }
}

/* bytes [src, src + n) of one buffer to the front of another (the unconsumed tail after a drain) */
__global__ void __launch_bounds__(256)
k_st_copy(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int64_t n)
{
    struct __attribute__((packed, aligned(1))) U128u { uint32_t a, b, c, d; };    /* sixteen bytes at any address */
    const int64_t n16 = n >> 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256)
        ((U128u *)dst)[i] = ((const U128u *)src)[i];
    for (int64_t i = (n16 << 4) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
}

#endif /* CRUMBLE_STREAM_KERNELS_H */
