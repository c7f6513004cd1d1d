/*
 * gz_kernels.h -- BGZF members inflated and deflated on the device (SURVEY.md 8 f1 "and/or GPU inflate; symmetric BGZF
 * deflate writer").  Reference crumble reads and writes its records through htslib's bgzf layer (sam_read1 inside
 * pileup_callback, snp_score.c:1116-1117; sam_write1 inside flush_bam_list, snp_score.c:1101), which inflates and deflates
 * 64 KiB members on host threads.  DEFLATE itself is RFC 1951 (zlib is not part of /root/reference): the format is restated
 * here from the RFC, the parity tests hold it against zlib / libdeflate output in both directions.
 *
 * Inflate is two kernels.  A DEFLATE stream is walked symbol by symbol, so one lane per member does the walk
 * (k_gz_tokens: Huffman tables in LDS, the compressed bytes staged through LDS by the whole wave, 16 waves per CU)
 * and writes what it found as 4-byte TOKENS -- up to three literal bytes, or (length, distance) -- never touching the output
 * window; a workgroup per member then plays the tokens into a 64 KiB LDS image of the member (k_gz_resolve: output positions
 * by a prefix sum over the tokens' lengths, copies in dependency rounds, long copies by whole waves), checks the CRC-32 and
 * stores the bytes with coalesced stores.  The serial half is bound by LDS latency per symbol, the parallel half is cheap.
 *
 * Deflate is one kernel, a workgroup per 0xff00-byte block (k_gz_deflate): candidates from a two-way hash table, a run
 * detector (distance 1: what crumble leaves in the quality strings) and the neighbour's distance, greedy parse by pointer
 * jumping, dynamic Huffman codes (exact, length-limited), bits placed by a prefix sum; stored block when that is smaller.
 */
#ifndef CRUMBLE_GZ_KERNELS_H
#define CRUMBLE_GZ_KERNELS_H

#define GZ_TB        10            /* literal/length table: bits of the primary look-up */
#define GZ_DB        8             /* distance table */
#define GZ_TOKBUF    256           /* tokens staged in LDS between two cooperative flushes */
#define GZ_RING      256           /* 32-bit words of compressed input staged in LDS (four chunks of 64) */
#define GZ_WAVES     4             /* members per workgroup in k_gz_tokens */

/* table entries: bits 0-4 the code's length, 5-7 kind, 8-12 extra bits, 16-31 literal / base length / base distance */
#define GZ_K_LIT     1u
#define GZ_K_LEN     2u            /* (distance table: a distance) */
#define GZ_K_EOB     3u
#define GZ_K_LONG    4u            /* code longer than the table's bits: decoded the canonical way */
#define GZ_ENTRY(len_, kind_, xb_, val_) ((uint32_t)(len_) | ((uint32_t)(kind_) << 5) | ((uint32_t)(xb_) << 8) | ((uint32_t)(val_) << 16))

/* error bits of a member (status[]) */
#define GZ_E_BTYPE   1u            /* reserved block type / bad stored-block length */
#define GZ_E_CODES   2u            /* over-subscribed code lengths, a repeat without a previous length, too many lengths */
#define GZ_E_SYMBOL  4u            /* an unassigned code in the stream */
#define GZ_E_DIST    8u            /* distance beyond the start of the member, or output beyond ISIZE */
#define GZ_E_INPUT   16u           /* ran past the member's compressed bytes */
#define GZ_E_SIZE    32u           /* the tokens do not add up to ISIZE */
#define GZ_E_CRC     64u           /* CRC-32 differs from the member's trailer */

/* one BGZF member: where its DEFLATE payload starts in the compressed buffer, where its bytes go (crumble_gz_block of
 * include/crumble_hip.h, same layout) */
struct GzBlock { int64_t coff, uoff; uint32_t clen, ulen, crc, pad_; };

struct __attribute__((packed, aligned(1))) GzU32u { uint32_t v; };      /* a 32-bit word at any byte address (global memory) */
#ifdef CRUMBLE_EMU
#define GZ_GLOBAL(T) T
#else
#define GZ_GLOBAL(T) __attribute__((address_space(1))) T
#endif

struct GzWave {
    uint32_t lit[1 << GZ_TB];
    uint32_t dst[1 << GZ_DB];
    uint32_t in[GZ_RING];
    uint32_t tok[GZ_TOKBUF];
    uint16_t lsym[288], dsym[32];      /* symbols in canonical order (by length, then value) */
    uint16_t lcnt[16], dcnt[16];       /* codes per length */
    uint16_t lfirst[16], dfirst[16];   /* first canonical code of each length */
    uint16_t loffs[16], doffs[16];     /* index of that code's symbol in lsym / dsym */
    uint8_t  lens[320];                /* code lengths as the block header gives them: literal/length, then distance */
    int32_t  nlit, ndist;
    int32_t  sh_wi, sh_ntok, sh_state; /* lane 0 -> the wave */
    /* the code-length code of a dynamic header (lane 0's scratch: in LDS, not in 60 registers of every lane) */
    int16_t  ccnt[8], cfirst[8], coffs[8], cnxt[8];
    uint8_t  cl[19], csym[19];
};

__device__ __forceinline__ uint32_t gz_bitrev(uint32_t v, int n)
{   /* the low n bits of v reversed (Huffman codes are packed starting from their most significant bit, RFC 1951 3.1.1) */
    uint32_t r = 0;
    for (int i = 0; i < n; i++) { r = (r << 1) | (v & 1u); v >>= 1; }
    return r;
}

__device__ __forceinline__ uint32_t gz_lit_entry(int sym, int len)
{   /* RFC 1951 3.2.5 */
    const int lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    const int lext[29]  = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    if (sym < 256) return GZ_ENTRY(len, GZ_K_LIT, 0, sym);
    if (sym == 256) return GZ_ENTRY(len, GZ_K_EOB, 0, 0);
    if (sym > 285) return 0u;                              /* 286, 287: never in a valid stream */
    return GZ_ENTRY(len, GZ_K_LEN, lext[sym - 257], lbase[sym - 257]);
}
__device__ __forceinline__ uint32_t gz_dist_entry(int sym, int len)
{
    const int dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073,
                           4097, 6145, 8193, 12289, 16385, 24577};
    const int dext[30]  = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    if (sym > 29) return 0u;
    return GZ_ENTRY(len, GZ_K_LEN, dext[sym], dbase[sym]);
}

/* One decode table from n code lengths (all lanes of the wave; lane 0 does the short serial parts).  Returns 0, or
 * GZ_E_CODES for an over-subscribed set.  Incomplete sets are taken as they are (a stream that uses a missing code
 * fails there): zlib allows the one-distance-code case, libdeflate writes it. */
__device__ __forceinline__ unsigned gz_build_table(const uint8_t *lens, int n, int tb, uint32_t *tab, uint16_t *sym, uint16_t *cnt,
                                                   uint16_t *first, uint16_t *offs, bool is_dist, int lane)
{
    for (int i = lane; i < (1 << tb); i += 64) tab[i] = 0u;
    if (lane < 16) cnt[lane] = 0;
    WAVE_SYNC();
    unsigned err = 0;
    if (lane == 0) {
        for (int s = 0; s < n; s++) cnt[lens[s]]++;
        cnt[0] = 0;
        int left = 1, code = 0, o = 0;
        for (int l = 1; l <= 15; l++) {
            left <<= 1; left -= cnt[l];
            if (left < 0) err = GZ_E_CODES;
            code = (code + (l > 1 ? cnt[l - 1] : 0)) << 1;
            first[l] = (uint16_t)code; offs[l] = (uint16_t)o; o += cnt[l];
        }
        if (!err) {
            uint16_t nxt[16];
            for (int l = 0; l < 16; l++) nxt[l] = offs[l];
            for (int s = 0; s < n; s++) if (lens[s]) sym[nxt[lens[s]]++] = (uint16_t)s;
        }
    }
    err = (unsigned)READLANE(err, 0);
    WAVE_SYNC();
    if (err) return err;
    int total = 0;
    for (int l = 1; l <= 15; l++) total += cnt[l];
    for (int i = lane; i < total; i += 64) {
        const int s = sym[i], l = lens[s];
        const uint32_t code = (uint32_t)first[l] + (uint32_t)(i - offs[l]);
        const uint32_t rev = gz_bitrev(code, l);
        if (l <= tb) {
            const uint32_t e = is_dist ? gz_dist_entry(s, l) : gz_lit_entry(s, l);
            for (uint32_t j = rev; j < (1u << tb); j += 1u << l) tab[j] = e;
        } else tab[rev & ((1u << tb) - 1u)] = GZ_ENTRY(0, GZ_K_LONG, 0, 0);        /* (every writer stores the same word) */
    }
    WAVE_SYNC();
    return 0;
}

/* a code longer than the table's bits: bit by bit against the canonical first codes (at most fifteen steps) */
__device__ __forceinline__ uint32_t gz_decode_long(uint64_t bb, const uint16_t *cnt, const uint16_t *first, const uint16_t *offs,
                                                   const uint16_t *sym, bool is_dist)
{
    uint32_t code = 0;
    for (int l = 1; l <= 15; l++) {
        code = (code << 1) | (uint32_t)(bb & 1u); bb >>= 1;
        const uint32_t c = cnt[l], f = first[l];
        if (c && code >= f && code - f < c) {
            const int s = sym[offs[l] + (code - f)];
            return is_dist ? gz_dist_entry(s, l) : gz_lit_entry(s, l);
        }
    }
    return 0u;
}

/* tokens: literal  = bytes in bits 0-23 (first byte lowest), their number 1..3 in bits 24-25
 *         match    = bit 31 | (length - 3) << 16 | (distance - 1) */
#define GZ_TOK_MATCH(len_, dist_) (0x80000000u | ((uint32_t)((len_) - 3) << 16) | (uint32_t)((dist_) - 1))

/* ---- k_gz_tokens: a wave per member ------------------------------------------------------------------------------ */
__global__ void __launch_bounds__(GZ_WAVES * 64)
#ifndef CRUMBLE_EMU
__attribute__((amdgpu_waves_per_eu(4, 4)))                   /* sixteen members per CU: the walk waits on LDS, occupancy is its throughput */
#endif
k_gz_tokens(int n_members, const GzBlock *__restrict__ blk, const uint8_t *__restrict__ comp, uint32_t *__restrict__ gtok,
            const int64_t *__restrict__ tok_off, uint32_t *__restrict__ ntok_out, uint32_t *__restrict__ status)
{
    __shared__ GzWave LW[GZ_WAVES];
    const int lane = threadIdx.x & 63, wv = (int)(threadIdx.x >> 6);
    const int mi = (int)blockIdx.x * GZ_WAVES + wv;
    if (mi >= n_members) return;                            /* (whole waves leave: no workgroup barrier below) */
    GzWave &W = LW[wv];
    const GzBlock m = blk[mi];
    const uint64_t a0 = (uint64_t)m.coff & ~(uint64_t)3;    /* aligned base of the staged words */
    const int lead = (int)((uint64_t)m.coff & 3);
    const uint32_t *src = (const uint32_t *)(comp + a0);
    const int64_t nwords = ((int64_t)lead + (int64_t)m.clen + 3) / 4;     /* words that hold this member's bytes */
    uint32_t *tout = gtok + tok_off[mi];

    enum { S_HEADER = 0, S_BUILD = 1, S_DECODE = 2, S_STORED = 3, S_DONE = 4 };
    /* lane 0's state (every lane carries the variables; only lane 0's mean anything) */
    uint64_t bb = 0; int bc = 0; int wi = 0;
    int state = S_HEADER, bfinal = 0, stored_left = 0;
    uint32_t litacc = 0; int litn = 0;
    int64_t outpos = 0;
    unsigned err = 0;
    int loaded = 0;                                          /* chunks of 64 words in the ring (wave-uniform) */
    int64_t tokpos = 0;
    bool first = true;

    for (;;) {
        /* the ring: four chunks ahead of the consumer */
        const int cons = READLANE(wi, 0) >> 6;
        while (loaded < cons + 4) { W.in[((loaded & 3) << 6) + lane] = src[(int64_t)loaded * 64 + lane]; loaded++; }
        WAVE_SYNC();
        if (state == S_BUILD) {                              /* (wave-uniform: set below from lane 0's) */
            unsigned e = gz_build_table(W.lens, W.nlit, GZ_TB, W.lit, W.lsym, W.lcnt, W.lfirst, W.loffs, false, lane);
            e |= gz_build_table(W.lens + W.nlit, W.ndist, GZ_DB, W.dst, W.dsym, W.dcnt, W.dfirst, W.doffs, true, lane);
            err |= e;
            state = e ? S_DONE : S_DECODE;
        }
        int ntok = 0;
        if (lane == 0) {
            if (first) { bb = (uint64_t)(W.in[0] >> (8 * lead)); bc = 32 - 8 * lead; wi = 1; first = false; }
            const int wlimit = (loaded << 6) - 3;            /* words the ring holds beyond any doubt */
#define GZ_REFILL() do { if (bc <= 32) { bb |= (uint64_t)W.in[wi & (GZ_RING - 1)] << bc; bc += 32; wi++; } } while (0)
#define GZ_TAKE(n_) do { bb >>= (n_); bc -= (n_); } while (0)
#define GZ_EMIT(t_) do { W.tok[ntok++] = (t_); } while (0)
#define GZ_FLUSH_LIT() do { if (litn) { GZ_EMIT(litacc | ((uint32_t)litn << 24)); litacc = 0; litn = 0; } } while (0)
            while (state != S_DONE && state != S_BUILD && wi < wlimit && ntok < GZ_TOKBUF - 2) {
                if ((int64_t)wi > nwords + 3) { err |= GZ_E_INPUT; state = S_DONE; break; }
                if (state == S_HEADER && wi + 96 > (loaded << 6)) break;    /* a dynamic header is up to ~72 words: refill first */
                GZ_REFILL();
                if (state == S_HEADER) {
                    bfinal = (int)(bb & 1u);
                    const int bt = (int)((bb >> 1) & 3u);
                    GZ_TAKE(3);
                    if (bt == 0) {                           /* stored: to the byte boundary, LEN, NLEN */
                        GZ_TAKE(bc & 7);
                        GZ_REFILL();
                        const uint32_t ln = (uint32_t)(bb & 0xffffu), nl = (uint32_t)((bb >> 16) & 0xffffu);
                        GZ_TAKE(32);
                        if ((ln ^ nl) != 0xffffu) { err |= GZ_E_BTYPE; state = S_DONE; break; }
                        stored_left = (int)ln;
                        state = S_STORED;
                    } else if (bt == 1) {                    /* fixed codes, RFC 1951 3.2.6 */
                        for (int s = 0; s < 288; s++) W.lens[s] = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8);
                        for (int s = 0; s < 30; s++) W.lens[288 + s] = 5;
                        W.nlit = 288; W.ndist = 30;
                        state = S_BUILD;
                    } else if (bt == 2) {                    /* dynamic codes, RFC 1951 3.2.7 */
                        const int order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                        const int hlit = (int)(bb & 31u) + 257, hdist = (int)((bb >> 5) & 31u) + 1, hclen = (int)((bb >> 10) & 15u) + 4;
                        GZ_TAKE(14);
                        for (int i = 0; i < 19; i++) W.cl[i] = 0;
                        for (int i = 0; i < hclen; i++) { GZ_REFILL(); W.cl[order[i]] = (uint8_t)(bb & 7u); GZ_TAKE(3); }
                        /* the code-length code: at most seven bits, decoded the canonical way */
                        for (int l = 0; l < 8; l++) W.ccnt[l] = 0;
                        for (int i = 0; i < 19; i++) W.ccnt[W.cl[i]]++;
                        W.ccnt[0] = 0;
                        {
                            int left = 1, code = 0, o = 0;
                            for (int l = 1; l <= 7; l++) {
                                left <<= 1; left -= W.ccnt[l];
                                if (left < 0) err |= GZ_E_CODES;
                                code = (code + (l > 1 ? W.ccnt[l - 1] : 0)) << 1;
                                W.cfirst[l] = (int16_t)code; W.coffs[l] = (int16_t)o; o += W.ccnt[l];
                            }
                            for (int l = 0; l < 8; l++) W.cnxt[l] = W.coffs[l];
                            for (int i = 0; i < 19; i++) if (W.cl[i]) W.csym[W.cnxt[W.cl[i]]++] = (uint8_t)i;
                        }
                        if (hlit > 286 || hdist > 30) err |= GZ_E_CODES;
                        if (err) { state = S_DONE; break; }
                        int k = 0, prev = 0;
                        const int need = hlit + hdist;
                        while (k < need) {
                            GZ_REFILL();
                            int s = -1; uint32_t code = 0; uint64_t t = bb;
                            for (int l = 1; l <= 7; l++) {
                                code = (code << 1) | (uint32_t)(t & 1u); t >>= 1;
                                const int cn = W.ccnt[l], cf = W.cfirst[l];
                                if (cn && (int)code >= cf && (int)code - cf < cn) { s = W.csym[W.coffs[l] + ((int)code - cf)]; GZ_TAKE(l); break; }
                            }
                            if (s < 0) { err |= GZ_E_SYMBOL; break; }
                            if (s < 16) { W.lens[k++] = (uint8_t)s; prev = s; }
                            else {
                                int rep, val = 0;
                                if (s == 16) { if (k == 0) { err |= GZ_E_CODES; break; } val = prev; rep = 3 + (int)(bb & 3u); GZ_TAKE(2); }
                                else if (s == 17) { rep = 3 + (int)(bb & 7u); GZ_TAKE(3); }
                                else { rep = 11 + (int)(bb & 127u); GZ_TAKE(7); }
                                if (k + rep > need) { err |= GZ_E_CODES; break; }
                                for (int r = 0; r < rep; r++) W.lens[k++] = (uint8_t)val;
                                if (s != 16) prev = 0;
                            }
                            if ((int64_t)wi > nwords + 3) { err |= GZ_E_INPUT; break; }
                        }
                        if (err) { state = S_DONE; break; }
                        if (W.lens[256] == 0) { err |= GZ_E_CODES; state = S_DONE; break; }      /* no end-of-block code */
                        W.nlit = hlit; W.ndist = hdist;
                        state = S_BUILD;
                    } else { err |= GZ_E_BTYPE; state = S_DONE; }
                    continue;
                }
                if (state == S_STORED) {
                    if (stored_left == 0) { state = bfinal ? S_DONE : S_HEADER; continue; }
                    if (outpos >= (int64_t)m.ulen) { err |= GZ_E_DIST; state = S_DONE; break; }
                    litacc |= (uint32_t)(bb & 0xffu) << (8 * litn); litn++;
                    GZ_TAKE(8); stored_left--; outpos++;
                    if (litn == 3) GZ_FLUSH_LIT();
                    continue;
                }
                /* S_DECODE: one literal/length symbol */
                uint32_t e = W.lit[bb & ((1u << GZ_TB) - 1u)];
                if (((e >> 5) & 7u) == GZ_K_LONG) e = gz_decode_long(bb, W.lcnt, W.lfirst, W.loffs, W.lsym, false);
                const uint32_t kind = (e >> 5) & 7u;
                if (kind == 0u) { err |= GZ_E_SYMBOL; state = S_DONE; break; }
                GZ_TAKE((int)(e & 31u));
                if (kind == GZ_K_LIT) {
                    if (outpos >= (int64_t)m.ulen) { err |= GZ_E_DIST; state = S_DONE; break; }
                    litacc |= ((e >> 16) & 0xffu) << (8 * litn); litn++; outpos++;
                    if (litn == 3) GZ_FLUSH_LIT();
                    continue;
                }
                if (kind == GZ_K_EOB) { state = bfinal ? S_DONE : S_HEADER; continue; }
                {   /* a length, then its distance */
                    const int xb = (int)((e >> 8) & 31u);
                    const int len = (int)(e >> 16) + (int)(bb & ((1u << xb) - 1u));
                    GZ_TAKE(xb);
                    GZ_REFILL();
                    uint32_t d = W.dst[bb & ((1u << GZ_DB) - 1u)];
                    if (((d >> 5) & 7u) == GZ_K_LONG) d = gz_decode_long(bb, W.dcnt, W.dfirst, W.doffs, W.dsym, true);
                    if (((d >> 5) & 7u) != GZ_K_LEN) { err |= GZ_E_SYMBOL; state = S_DONE; break; }
                    GZ_TAKE((int)(d & 31u));
                    const int dxb = (int)((d >> 8) & 31u);
                    const int dist = (int)(d >> 16) + (int)(bb & ((1u << dxb) - 1u));
                    GZ_TAKE(dxb);
                    if ((int64_t)dist > outpos || outpos + len > (int64_t)m.ulen) { err |= GZ_E_DIST; state = S_DONE; break; }
                    GZ_FLUSH_LIT();
                    GZ_EMIT(GZ_TOK_MATCH(len, dist));
                    outpos += len;
                }
            }
            if (state == S_DONE) {
                GZ_FLUSH_LIT();
                if (!err && outpos != (int64_t)m.ulen) err |= GZ_E_SIZE;
                /* bits consumed beyond the member's payload: the stream ran into its neighbour */
                if (!err && (int64_t)wi * 32 - bc > ((int64_t)lead + (int64_t)m.clen) * 8) err |= GZ_E_INPUT;
            }
#undef GZ_REFILL
#undef GZ_TAKE
#undef GZ_EMIT
#undef GZ_FLUSH_LIT
            W.sh_ntok = ntok; W.sh_state = state;
        }
        WAVE_SYNC();
        ntok = W.sh_ntok; state = W.sh_state;
        for (int i = lane; i < ntok; i += 64) tout[tokpos + i] = W.tok[i];
        tokpos += ntok;
        WAVE_SYNC();                                         /* (the token buffer is lane 0's again) */
        if (state == S_DONE) break;
    }
    if (lane == 0) { ntok_out[mi] = (uint32_t)tokpos; status[mi] = err; }
}

/* ---- k_gz_tokens_lanes: a LANE per member --------------------------------------------------------------------------
 * The wave-per-member walk above keeps one lane of sixty-four busy; with thousands of members in a call the other way
 * round pays: a lane walks its own member (bit buffer, position and state in its registers), the walks of a wave advance
 * in lockstep one step per loop iteration whatever each lane's step is (a literal, a match, a header field).
 *
 * What bounds such a walk is instruction issue, not memory: a wave alone on its SIMD issues an instruction every four to
 * five cycles, and every iteration executes every path that ANY of its lanes takes.  So (1) only LPW lanes of a wave walk
 * (8 or 16: eight or four waves per CU instead of one, fewer lanes to diverge in each), (2) an iteration takes one more
 * literal after its first symbol when that is what comes next, (3) nothing on a path goes to global memory except the input word (always loaded
 * one refill ahead) and the tokens: the decode tables of a member live in LDS as 16-bit entries interleaved by lane
 * (entry j of lane L at half-word j * LPW + L) -- 2^9 literal/length and 2^7 distance entries, behind them the canonical
 * decoder's counts / first codes / offsets and sorted symbols for the codes longer than that, and the code lengths of a
 * dynamic header (zeroed when the header starts, so that the zero runs, most of a header, cost no stores): 2432 bytes per
 * member.  A lane that reaches a block header reads its code lengths one symbol per iteration like everybody else's
 * symbols; the table fill is done for one lane at a time by the whole wave.  Same tokens, same status words as
 * k_gz_tokens. */
#define GZ2_TB       9             /* bits of the literal/length look-up in LDS */
#define GZ2_DB       7             /* of the distance look-up */
#ifndef GZ2_MORE
#define GZ2_MORE     1             /* literals an iteration takes beyond its first symbol (measured, 10.5 k members, walk + copy: 17.0 / 16.4 / 16.8 /
                                    * 18.0 / 19.2 ms for 0 / 1 / 2 / 3 / 5: every lane pays for a path that any lane takes) */
#endif
/* 16-bit entry: code length (bits 0-3; 0: not a code), kind (bits 4-5), value (bits 6-15: the literal, or the index of the
 * length / distance symbol -- base and extra bits follow from it by arithmetic, RFC 1951 3.2.5) */
#define GZ2_LIT      0u
#define GZ2_LEN      1u
#define GZ2_EOB      2u
#define GZ2_LONG     3u
#define GZ2_E(len_, kind_, val_) ((uint16_t)((uint32_t)(len_) | ((uint32_t)(kind_) << 4) | ((uint32_t)(val_) << 6)))

__device__ __forceinline__ uint32_t gz_brev15(uint32_t v)
{   /* the low fifteen bits, first bit of the stream on top */
#ifdef CRUMBLE_EMU
    return gz_bitrev(v & 0x7fffu, 15);
#else
    return __builtin_bitreverse32(v) >> 17;
#endif
}

/* the decode table of one member's block, by the whole wave (uniform arguments): n code lengths at lens[0], lens[st], ..
 * -> tab, sym, cnt / first / offs (all LDS, stride st) */
__device__ __forceinline__ unsigned gz2_build(const uint8_t *lens, int n, int tb, uint16_t *tab, uint16_t *sym, uint16_t *cfo, bool is_dist, int lane, int st)
{
    for (int i = lane; i < (1 << tb); i += 64) tab[i * st] = 0;
    int cnt[16];
    #pragma unroll
    for (int q = 0; q < 16; q++) cnt[q] = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int l = i0 + lane < n ? lens[(i0 + lane) * st] : 0;
        #pragma unroll
        for (int q = 1; q < 16; q++) cnt[q] += __popcll(__ballot(l == q));
    }
    int first[16], offs[16], left = 1, code = 0, o = 0;
    bool over = false;
    first[0] = 0; offs[0] = 0;
    #pragma unroll
    for (int l = 1; l < 16; l++) {
        left <<= 1; left -= cnt[l];
        if (left < 0) over = true;
        code = (code + (l > 1 ? cnt[l - 1] : 0)) << 1;
        first[l] = code; offs[l] = o; o += cnt[l];
    }
    if (over) return GZ_E_CODES;
    if (lane < 16) {
        int c = 0, f = 0, of = 0;
        #pragma unroll
        for (int q = 0; q < 16; q++) if (lane == q) { c = cnt[q]; f = first[q]; of = offs[q]; }
        cfo[lane * st] = (uint16_t)c; cfo[(16 + lane) * st] = (uint16_t)f; cfo[(32 + lane) * st] = (uint16_t)of;
    }
    WAVE_SYNC();                                             /* (the cleared entries are in LDS before the codes go in) */
    int seen[16];
    #pragma unroll
    for (int q = 0; q < 16; q++) seen[q] = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int s = i0 + lane, l = s < n ? lens[s * st] : 0;
        int rank = 0, f = 0, of = 0;
        #pragma unroll
        for (int q = 1; q < 16; q++) {
            const unsigned long long bm = __ballot(l == q);
            if (l == q) { rank = seen[q] + __popcll(bm & ((1ull << lane) - 1ull)); f = first[q]; of = offs[q]; }
            seen[q] += __popcll(bm);
        }
        if (l) {
            const uint32_t rev = gz_bitrev((uint32_t)(f + rank), l);
            sym[(of + rank) * st] = (uint16_t)s;
            if (l <= tb) {
                uint16_t e;
                if (is_dist) e = s > 29 ? (uint16_t)0 : GZ2_E(l, GZ2_LEN, s);
                else e = s < 256 ? GZ2_E(l, GZ2_LIT, s) : s == 256 ? GZ2_E(l, GZ2_EOB, 0) : s > 285 ? (uint16_t)0 : GZ2_E(l, GZ2_LEN, s - 257);
                for (uint32_t j = rev; j < (1u << tb); j += 1u << l) tab[j * st] = e;
            } else tab[(rev & ((1u << tb) - 1u)) * st] = GZ2_E(0, GZ2_LONG, 0);
        }
    }
    return 0;
}

/* a code longer than the look-up's bits, against the canonical first codes of the lengths tb + 1 .. 15: the symbol (its
 * length in *cl), or -1 */
__device__ __forceinline__ int gz2_long(uint64_t bb, const uint16_t *cfo, const uint16_t *sym, int tb, int st, int *cl)
{
    const uint32_t v = gz_brev15((uint32_t)bb);
    int s = -1;
    for (int l = tb + 1; l <= 15; l++) {
        const uint32_t c = v >> (15 - l), n = cfo[l * st], f = cfo[(16 + l) * st];
        if (s < 0 && c - f < n) { s = (int)sym[(cfo[(32 + l) * st] + (c - f)) * st]; *cl = l; }
    }
    return s;
}

template <int LPW>
__global__ void __launch_bounds__(64)
k_gz_tokens_lanes(int n_members, const GzBlock *__restrict__ blk, const uint8_t *__restrict__ comp, uint32_t *__restrict__ gtok,
                  const int64_t *__restrict__ tok_off, uint32_t *__restrict__ ntok_out, uint32_t *__restrict__ status)
{
    __shared__ uint16_t TLs[(1 << GZ2_TB) * LPW], TDs[(1 << GZ2_DB) * LPW];
    __shared__ uint16_t SYs[320 * LPW];                       /* literal/length symbols in canonical order [288], distance symbols [32] */
    __shared__ uint16_t CFs[96 * LPW];                        /* lcnt lfirst loffs dcnt dfirst doffs, sixteen each */
    __shared__ uint8_t LNs[320 * LPW];                        /* code lengths as read */
    const int lane = threadIdx.x & 63;
    const int mi = (int)blockIdx.x * LPW + lane;
    const bool valid = lane < LPW && mi < n_members;
    const int ml = valid ? lane : 0;                          /* (the idle lanes' addresses stay inside the arrays) */
    GzBlock m;
    m.coff = 0; m.uoff = 0; m.clen = 0; m.ulen = 0; m.crc = 0; m.pad_ = 0;
    if (valid) m = blk[mi];
    const uint8_t *src = comp + m.coff;
    uint32_t *tout = gtok + (valid ? tok_off[mi] : 0);
    const uint16_t *TL = TLs + ml, *TD = TDs + ml, *SY = SYs + ml, *CF = CFs + ml;
    uint8_t *lens = LNs + ml;
    enum { S_HDR = 0, S_CLEN = 1, S_LENS = 2, S_BUILD = 3, S_DECODE = 4, S_STORED = 5, S_DONE = 6 };
    int state = valid ? S_HDR : S_DONE;
    uint64_t bb = 0; int bc = 0, ip = 0;
    int bfinal = 0, stored_left = 0, k = 0, need = 0, hlit = 0, hdist = 0, hclen = 0, prev = 0;
    uint64_t clpack = 0, cnt7 = 0, first7 = 0, sortlo = 0, sorthi = 0;
    uint32_t litacc = 0; int litn = 0, ntok = 0;
    int outpos = 0;
    unsigned err = 0;
    const int ulen = (int)m.ulen, clen = (int)m.clen;
    /* the word behind the bit buffer, loaded one refill ahead (the buffer behind the last member is padded: GZ_IN_PAD) */
    uint32_t nxt = valid ? ((const GzU32u *)src)->v : 0u;

#define G2_REFILL() do { bb |= (uint64_t)nxt << bc; bc += 32; ip += 4; nxt = ((const GzU32u *)(src + ip))->v; } while (0)
#define G2_TAKE(n_) do { bb >>= (n_); bc -= (n_); } while (0)
#define G2_EMIT(t_) do { tout[ntok++] = (t_); } while (0)
#define G2_FLUSH_LIT() do { if (litn) { G2_EMIT(litacc | ((uint32_t)litn << 24)); litacc = 0; litn = 0; } } while (0)
#define G2_FAIL(e_) do { err |= (e_); state = S_DONE; } while (0)
    for (;;) {
        if (__ballot(state != S_DONE) == 0ull) break;
        if (state != S_DONE) {
            if (bc <= 32) G2_REFILL();
            if (ip > clen + 12) G2_FAIL(GZ_E_INPUT);
        }
        if (state == S_DECODE) {
            const uint32_t e = TL[((uint32_t)bb & ((1u << GZ2_TB) - 1u)) * LPW];
            int cl = (int)(e & 15u), xb = 0, val = (int)(e >> 6);
            uint32_t kind = (e >> 4) & 3u;
            bool bad = e == 0u;
            if (kind == GZ2_LONG) {
                const int sy = gz2_long(bb, CF, SY, GZ2_TB, LPW, &cl);
                bad = sy < 0 || sy > 285;
                kind = sy < 256 ? GZ2_LIT : sy == 256 ? GZ2_EOB : GZ2_LEN;
                val = sy < 256 ? sy : sy - 257;
            }
            if (kind == GZ2_LEN) {                           /* length symbol -> base, extra bits (RFC 1951 3.2.5) */
                if (val < 8) val += 3;
                else if (val == 28) val = 258;
                else { xb = (val - 4) >> 2; val = 3 + ((4 + (val & 3)) << xb); }
            }
            if (bad) G2_FAIL(GZ_E_SYMBOL);
            else {
                G2_TAKE(cl);
                if (kind == GZ2_LIT) {
                    if (outpos >= ulen) G2_FAIL(GZ_E_DIST);
                    else { litacc |= ((uint32_t)val & 0xffu) << (8 * litn); litn++; outpos++; if (litn == 3) G2_FLUSH_LIT(); }
                } else if (kind == GZ2_EOB) state = bfinal ? S_DONE : S_HDR;
                else {
                    const int len = val + (int)(bb & ((1u << xb) - 1u));
                    G2_TAKE(xb);
                    if (bc <= 32) G2_REFILL();
                    const uint32_t d = TD[((uint32_t)bb & ((1u << GZ2_DB) - 1u)) * LPW];
                    int dl = (int)(d & 15u), dxb = 0, dv = (int)(d >> 6);
                    bool dbad = ((d >> 4) & 3u) != GZ2_LEN;
                    if (((d >> 4) & 3u) == GZ2_LONG) {
                        dv = gz2_long(bb, CF + 48 * LPW, SY + 288 * LPW, GZ2_DB, LPW, &dl);
                        dbad = dv < 0 || dv > 29;
                    }
                    if (dv < 4) dv += 1;
                    else { dxb = (dv - 2) >> 1; dv = 1 + ((2 + (dv & 1)) << dxb); }
                    if (dbad) G2_FAIL(GZ_E_SYMBOL);
                    else {
                        G2_TAKE(dl);
                        const int dist = dv + (int)(bb & ((1u << dxb) - 1u));
                        G2_TAKE(dxb);
                        if (dist > outpos || outpos + len > ulen) G2_FAIL(GZ_E_DIST);
                        else { G2_FLUSH_LIT(); G2_EMIT(GZ_TOK_MATCH(len, dist)); outpos += len; }
                    }
                }
            }
            /* up to GZ2_MORE more symbols while they are literals whose bits are all in the buffer (anything else waits for the
             * next iteration's full path) */
            #pragma unroll
            for (int r = 0; r < GZ2_MORE; r++) {
                if (r == 2 && state == S_DECODE && bc <= 32) G2_REFILL();      /* (the word was asked for at the top of the iteration) */
                if (state == S_DECODE) {
                    const uint32_t e2 = TL[((uint32_t)bb & ((1u << GZ2_TB) - 1u)) * LPW];
                    const int c2 = (int)(e2 & 15u);
                    if ((e2 & 0x30u) == (GZ2_LIT << 4) && c2 != 0 && c2 <= bc && outpos < ulen) {
                        G2_TAKE(c2);
                        litacc |= (e2 >> 6) << (8 * litn); litn++; outpos++;
                        if (litn == 3) G2_FLUSH_LIT();
                    }
                }
            }
        } else if (state == S_STORED) {
            if (stored_left == 0) state = bfinal ? S_DONE : S_HDR;
            else if (outpos >= ulen) G2_FAIL(GZ_E_DIST);
            else {
                litacc |= (uint32_t)(bb & 0xffu) << (8 * litn); litn++;
                G2_TAKE(8); stored_left--; outpos++;
                if (litn == 3) G2_FLUSH_LIT();
            }
        } else if (state == S_HDR) {
            bfinal = (int)(bb & 1u);
            const int bt = (int)((bb >> 1) & 3u);
            G2_TAKE(3);
            if (bt == 0) {
                G2_TAKE(bc & 7);
                if (bc < 32) G2_REFILL();
                const uint32_t ln = (uint32_t)(bb & 0xffffu), nl = (uint32_t)((bb >> 16) & 0xffffu);
                G2_TAKE(32);
                if ((ln ^ nl) != 0xffffu) G2_FAIL(GZ_E_BTYPE);
                else { stored_left = (int)ln; state = S_STORED; }
            } else if (bt == 1) {
                for (int sy = 0; sy < 288; sy++) lens[sy * LPW] = (uint8_t)(sy < 144 ? 8 : sy < 256 ? 9 : sy < 280 ? 7 : 8);
                for (int sy = 0; sy < 30; sy++) lens[(288 + sy) * LPW] = 5;
                hlit = 288; hdist = 30;
                state = S_BUILD;
            } else if (bt == 2) {
                hlit = (int)(bb & 31u) + 257; hdist = (int)((bb >> 5) & 31u) + 1; hclen = (int)((bb >> 10) & 15u) + 4;
                G2_TAKE(14);
                if (hlit > 286 || hdist > 30) G2_FAIL(GZ_E_CODES);
                else {
                    clpack = 0; k = 0; state = S_CLEN;
                    for (int sy = 0; sy < 316; sy++) lens[sy * LPW] = 0;      /* (the zero runs of the header then only move k) */
                }
            } else G2_FAIL(GZ_E_BTYPE);
        } else if (state == S_CLEN) {
            /* one 3-bit length of the code-length code per iteration, packed by symbol into 57 bits */
            const int order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            int sy = 0;
            #pragma unroll
            for (int q = 0; q < 19; q++) if (k == q) sy = order[q];
            clpack |= (uint64_t)(bb & 7u) << (3 * sy);
            G2_TAKE(3);
            if (++k == hclen) {
                /* the canonical decoder of this code in registers: counts (5 bits each), first codes (8 bits each), the
                 * symbols in canonical order (5 bits each, nineteen of them over two words) */
                cnt7 = 0; first7 = 0; sortlo = 0; sorthi = 0;
                int c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (int i = 0; i < 19; i++) { const int l = (int)((clpack >> (3 * i)) & 7u);
                    #pragma unroll
                    for (int q = 1; q < 8; q++) c[q] += l == q; }
                int left = 1, code = 0, nx[8];
                nx[0] = 0;
                {
                    int o = 0;
                    #pragma unroll
                    for (int l = 1; l < 8; l++) {
                        left <<= 1; left -= c[l];
                        code = (code + (l > 1 ? c[l - 1] : 0)) << 1;
                        cnt7 |= (uint64_t)c[l] << (5 * (l - 1)); first7 |= (uint64_t)(code & 255) << (8 * (l - 1));
                        nx[l] = o; o += c[l];
                    }
                }
                if (left < 0) G2_FAIL(GZ_E_CODES);
                else {
                    for (int i = 0; i < 19; i++) {
                        const int l = (int)((clpack >> (3 * i)) & 7u);
                        if (!l) continue;
                        int pos = 0;
                        #pragma unroll
                        for (int q = 1; q < 8; q++) if (l == q) pos = nx[q]++;
                        if (pos < 12) sortlo |= (uint64_t)i << (5 * pos); else sorthi |= (uint64_t)i << (5 * (pos - 12));
                    }
                    k = 0; prev = 0; need = hlit + hdist;
                    state = S_LENS;
                }
            }
        } else if (state == S_LENS) {
            /* one code-length symbol per iteration */
            int sy = -1; uint32_t code = 0; uint64_t t = bb; int offs = 0;
            #pragma unroll
            for (int l = 1; l <= 7; l++) {
                code = (code << 1) | (uint32_t)(t & 1u); t >>= 1;
                const int cn = (int)((cnt7 >> (5 * (l - 1))) & 31u), cf = (int)((first7 >> (8 * (l - 1))) & 255u);
                if (sy < 0 && cn && (int)code >= cf && (int)code - cf < cn) {
                    const int pos = offs + (int)code - cf;
                    sy = (int)((pos < 12 ? sortlo >> (5 * pos) : sorthi >> (5 * (pos - 12))) & 31u);
                    G2_TAKE(l);
                }
                offs += cn;
            }
            if (sy < 0) G2_FAIL(GZ_E_SYMBOL);
            else if (sy < 16) { lens[k * LPW] = (uint8_t)sy; k++; prev = sy; }
            else {
                int rep;
                bool bad = false;
                if (sy == 16) { bad = k == 0; rep = 3 + (int)(bb & 3u); G2_TAKE(2); }
                else if (sy == 17) { rep = 3 + (int)(bb & 7u); G2_TAKE(3); }
                else { rep = 11 + (int)(bb & 127u); G2_TAKE(7); }
                if (bad || k + rep > need) G2_FAIL(GZ_E_CODES);
                else if (sy == 16) { for (int r = 0; r < rep; r++) lens[(k + r) * LPW] = (uint8_t)prev; k += rep; }
                else { k += rep; prev = 0; }                 /* (zeros are there already) */
            }
            if (state == S_LENS && k >= need) {
                if (lens[256 * LPW] == 0) G2_FAIL(GZ_E_CODES);
                else state = S_BUILD;
            }
        }
        /* ---- the table fills: one requesting lane at a time, by all sixty-four */
        unsigned long long bm = __ballot(state == S_BUILD);
        if (bm) {
            WAVE_SYNC();                                     /* the lanes' code lengths are in LDS */
            while (bm) {
                const int L = __builtin_ctzll(bm);
                bm &= bm - 1ull;
                const int nl = READLANE(hlit, L), nd = READLANE(hdist, L);
                unsigned e = gz2_build(LNs + L, nl, GZ2_TB, TLs + L, SYs + L, CFs + L, false, lane, LPW);
                e |= gz2_build(LNs + L + nl * LPW, nd, GZ2_DB, TDs + L, SYs + L + 288 * LPW, CFs + L + 48 * LPW, true, lane, LPW);
                if (lane == L) { if (e) G2_FAIL(e); else state = S_DECODE; }
            }
            WAVE_SYNC();                                     /* the tables are in LDS before anybody looks something up */
        }
    }
    if (valid) {
        G2_FLUSH_LIT();
        if (!err && outpos != ulen) err |= GZ_E_SIZE;
        if (!err && (int64_t)ip * 8 - bc > (int64_t)clen * 8) err |= GZ_E_INPUT;
        ntok_out[mi] = (uint32_t)ntok; status[mi] = err;
    }
#undef G2_REFILL
#undef G2_TAKE
#undef G2_EMIT
#undef G2_FLUSH_LIT
#undef G2_FAIL
}

/* ---- CRC-32 (the gzip one: reflected 0xEDB88320), slices combined the way zlib's crc32_combine does ----------------- */
#define GZ_POLY 0xedb88320u
HD uint32_t gz_multmodp(uint32_t a, uint32_t b)
{   /* a(x) * b(x) mod p(x), reflected representation */
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) { p ^= b; if ((a & (m - 1u)) == 0u) break; }
        m >>= 1;
        b = (b & 1u) ? (b >> 1) ^ GZ_POLY : b >> 1;
    }
    return p;
}
/* x^(n * 2^k) mod p(x); x2n[i] = x^(2^i) */
HD uint32_t gz_x2nmodp(uint64_t n, unsigned k, const uint32_t *x2n)
{
    uint32_t p = 1u << 31;
    while (n) { if (n & 1u) p = gz_multmodp(x2n[k & 31u], p); n >>= 1; k++; }
    return p;
}
/* CRC-32 of n <= 64 * nthr bytes of `img` (LDS) by a workgroup of nthr threads (a power of two).  Thread t takes the 64 bytes
 * that END (nthr - 1 - t) * 64 bytes in front of the end of the data (the first threads' slices are empty or short when n is
 * not 64 * nthr): every slice then sits a whole number of slices in front of the end, and two neighbouring partial results
 * combine by ONE shift that is the same for the whole level -- state(A || B) = state(A) * x^(8 |B|) + state0(B) (the register is
 * linear in state and data; zlib's crc32_combine is this identity on finished CRCs), |B| = 64 * 2^k bytes at level k, and
 * x^(2^(9 + k)) mod p is x2n[9 + k].  Ten levels with 8, 4, 2, 1, 1 .. waves at work; the version before shifted every
 * slice to the end by its own amount (up to sixteen polynomial products per thread, every wave, diverging): 95 k cycles per
 * block against ~15 k.  tab / x2n: LDS copies of the byte table and the powers; red: nthr words of LDS.  Returns the CRC
 * in every thread. */
__device__ __forceinline__ uint32_t gz_crc_block(const uint8_t *img, int n, const uint32_t *tab, const uint32_t *x2n, uint32_t *red,
                                                 int tid, int nthr)
{
    const int s0 = n - (nthr - tid) * 64;
    const int lo = s0 > 0 ? s0 : 0, hi = s0 + 64 > 0 ? s0 + 64 : 0;
    uint32_t c = 0;
    if (hi > lo) {
        if (lo == 0) c = 0xffffffffu;                        /* (the register's start value, where the data starts) */
        for (int i = lo; i < hi; i++) c = tab[(c ^ img[i]) & 0xffu] ^ (c >> 8);
    }
    red[tid] = c;
    __syncthreads();
    int k = 0;
    for (int half = nthr >> 1; half > 0; half >>= 1, k++) {
        uint32_t v = 0;
        if (tid < half) v = gz_multmodp(x2n[(9 + k) & 31], red[2 * tid]) ^ red[2 * tid + 1];
        __syncthreads();
        if (tid < half) red[tid] = v;
        __syncthreads();
    }
    const uint32_t r = n > 0 ? red[0] ^ 0xffffffffu : 0u;
    __syncthreads();
    return r;
}

/* a workgroup barrier for what went through LDS only: __syncthreads() also waits for the wave's outstanding GLOBAL stores
 * (vmcnt(0): a microsecond per barrier when every step of a loop leaves a store behind), this one for LDS traffic alone.
 * Whatever was stored to global memory in front of it is NOT ordered by it: a __syncthreads() must follow before another
 * wave reads those bytes. */
#ifdef CRUMBLE_EMU
#define GZ_LDS_BARRIER() __syncthreads()
#else
#define GZ_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif

/* does any thread of the workgroup say so?  (flag: a word of LDS) */
__device__ __forceinline__ bool gz_any(bool p, int32_t *flag, int tid)
{
    if (tid == 0) *flag = 0;
    __syncthreads();
    if (p) *flag = 1;
    __syncthreads();
    const bool r = *flag != 0;
    __syncthreads();
    return r;
}

/* ---- k_gz_resolve: a workgroup per member plays its tokens into LDS, checks the CRC, stores the bytes ------------------
 * No copy waits for another: every output byte gets a REFERENCE -- itself for a literal, position - distance for a byte of a
 * match -- and the references are shortened by pointer jumping (ref[p] = ref[ref[p]]) until each one names a literal; a
 * run written as "distance 1, length 258" is a chain of 258 that nine rounds resolve.  The member is done in two halves of
 * 32 KiB (a distance never exceeds 32 KiB, so the second half's chains end in the first half's finished bytes or in its own
 * literals): 64 KiB of image + 64 KiB of 16-bit references fit the LDS of one CU. */
#define GZ_RT        1024          /* threads: one workgroup per CU */
#define GZ_HALF      32768
struct GzResolve {
    uint8_t  img[65536 + 16];
    uint16_t ref[GZ_HALF];
    uint32_t crc_tab[256];
    uint32_t x2n[32];
    uint32_t red[GZ_RT];
    int32_t  scan[GZ_RT / 64 + 1];
    int32_t  anyflag, resume_chunk, resume_base;
};
__global__ void __launch_bounds__(GZ_RT)
k_gz_resolve(int n_members, const GzBlock *__restrict__ blk, const uint32_t *__restrict__ gtok, const int64_t *__restrict__ tok_off,
             const uint32_t *__restrict__ ntok_in, uint32_t *__restrict__ status, const uint32_t *__restrict__ crc_tab,
             const uint32_t *__restrict__ x2n, uint8_t *__restrict__ raw, int check_crc)
{
    __shared__ GzResolve R;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int mi = blockIdx.x; mi < n_members; mi += gridDim.x) {
    const GzBlock m = blk[mi];
    __syncthreads();                                         /* (the previous member's stores out of LDS are done) */
    if (status[mi] != 0u || m.ulen == 0u) continue;          /* (uniform: every thread reads the same word) */
    if (tid < 256) R.crc_tab[tid] = crc_tab[tid];
    if (tid < 32) R.x2n[tid] = x2n[tid];
    const uint32_t *tk = gtok + tok_off[mi];
    const int ntok = (int)ntok_in[mi];
    const int ulen = (int)m.ulen;
    unsigned err = 0;
    int total_out = 0;
    if (tid == 0) { R.resume_chunk = 0; R.resume_base = 0; }
    __syncthreads();
    for (int half = 0; half < 2; half++) {
        const int lo = half * GZ_HALF, hi = ulen < lo + GZ_HALF ? ulen : lo + GZ_HALF;
        if (lo >= ulen) break;
        /* ---- references (and literal bytes) of the positions [lo, hi): the tokens from where the previous half stopped */
        int outbase = R.resume_base;
        const int c0 = R.resume_chunk;
        __syncthreads();
        for (int c = c0; c < ntok; c += GZ_RT) {
            const bool valid = c + tid < ntok;
            const uint32_t t = valid ? tk[c + tid] : 0u;
            const bool is_match = (t >> 31) != 0u;
            const int len = !valid ? 0 : is_match ? (int)((t >> 16) & 0xffu) + 3 : (int)((t >> 24) & 3u);
            int incl = len;
            for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
            if (lane == 63) R.scan[wv] = incl;
            __syncthreads();
            int wbase = 0, total = 0;
            for (int w = 0; w < GZ_RT / 64; w++) { const int sc = R.scan[w]; if (w < wv) wbase += sc; total += sc; }
            const int pos = outbase + wbase + incl - len;
            const int dist = (int)(t & 0x7fffu) + 1;
            if (valid && (pos + len > ulen || (is_match && dist > pos) || (!is_match && len == 0))) err |= GZ_E_DIST;
#ifdef GZ_EXP_NOTOKW
            else if (false) {
#else
            else if (valid && pos < hi && pos + len > lo) {
#endif
                if (!is_match) {
                    for (int k = 0; k < len; k++) { const int p = pos + k; if (p >= lo && p < hi) { R.img[p] = (uint8_t)(t >> (8 * k)); R.ref[p - lo] = (uint16_t)p; } }
                } else {
                    int k0 = lo - pos; if (k0 < 0) k0 = 0;
                    int k1 = hi - pos; if (k1 > len) k1 = len;
                    for (int k = k0; k < k1; k++) R.ref[pos + k - lo] = (uint16_t)(pos + k - dist);
                }
            }
            /* the chunk in which the next half begins (the first one that reaches hi) */
            if (tid == 0 && outbase < hi && outbase + total >= hi) { R.resume_chunk = c; R.resume_base = outbase; }
            outbase += total;
            __syncthreads();
            if (outbase >= hi && half == 0) break;           /* (uniform) the rest belongs to the second half */
        }
        if (half == 1 || hi == ulen) total_out = outbase;
        if (gz_any(err != 0u, &R.anyflag, tid)) { err |= GZ_E_DIST; break; }
        /* ---- pointer jumping: until every reference names a literal of this half (ref[a] == a) or a byte below lo */
#ifdef GZ_EXP_NOJUMP                                          /* (diagnostic builds: wrong bytes, the kernel's time without this part) */
        for (int round = 0; round < 0; round++) {
#else
        for (int round = 0; round < 17; round++) {
#endif
            bool changed = false;
            /* eight positions at a time: their references first, then the references' references (two dependent LDS round
             * trips per eight positions instead of per position; a stale or a fresh value of another thread's reference is an
             * ancestor either way) */
            for (int p0 = lo + tid; p0 < hi; p0 += 8 * GZ_RT) {
                int a[8], b[8];
                #pragma unroll
                for (int u = 0; u < 8; u++) { const int p = p0 + u * GZ_RT; a[u] = p < hi ? (int)R.ref[p - lo] : p; }
                #pragma unroll
                for (int u = 0; u < 8; u++) { const int p = p0 + u * GZ_RT; b[u] = (a[u] != p && a[u] >= lo) ? (int)R.ref[a[u] - lo] : a[u]; }
                #pragma unroll
                for (int u = 0; u < 8; u++) { const int p = p0 + u * GZ_RT; if (b[u] != a[u]) { R.ref[p - lo] = (uint16_t)b[u]; changed = true; } }
            }
            if (!gz_any(changed, &R.anyflag, tid)) break;
        }
        for (int p0 = lo + tid; p0 < hi; p0 += 8 * GZ_RT) {
            int a[8]; uint8_t v[8];
            #pragma unroll
            for (int u = 0; u < 8; u++) { const int p = p0 + u * GZ_RT; a[u] = p < hi ? (int)R.ref[p - lo] : p; }
            #pragma unroll
            for (int u = 0; u < 8; u++) { const int p = p0 + u * GZ_RT; v[u] = a[u] != p ? R.img[a[u]] : (uint8_t)0; }
            #pragma unroll
            for (int u = 0; u < 8; u++) { const int p = p0 + u * GZ_RT; if (a[u] != p) R.img[p] = v[u]; }
        }
        __syncthreads();
    }
    if (!err && total_out != ulen) err |= GZ_E_SIZE;
    if (!gz_any(err != 0u, &R.anyflag, tid) && check_crc) {
        const uint32_t crc = gz_crc_block(R.img, ulen, R.crc_tab, R.x2n, R.red, tid, GZ_RT);
        if (crc != m.crc) err |= GZ_E_CRC;
    }
    if (gz_any(err != 0u, &R.anyflag, tid)) { if (err) atomicOr(&status[mi], err); continue; }
    uint8_t *dst = raw + m.uoff;
    const int n4 = ulen >> 2;
    for (int i = tid; i < n4; i += GZ_RT) { uint32_t v; memcpy(&v, &R.img[4 * i], 4); ((GzU32u *)dst)[i].v = v; }
    for (int i = (n4 << 2) + tid; i < ulen; i += GZ_RT) dst[i] = R.img[i];
    }
}

/* ==== deflate ======================================================================================================== */
#define GZ_DT        1024          /* threads of k_gz_deflate: one workgroup (and one block) per CU */
#define GZ_SUB       256           /* positions looked at together (four candidate lanes each) */
#define GZ_HB        13            /* bits of each of the two hash tables */
#define GZ_MAXBLK    65280         /* 0xff00: bytes per block (htslib's BGZF_BLOCK_SIZE) */
#define GZ_SLOT      65536         /* bytes of a member's output slot (header + payload + trailer never exceed it) */
#define GZ_RANGE     4096          /* positions parsed by one wave (sixteen ranges per block) */
#ifndef GZ_CMPMAX
#define GZ_CMPMAX    258           /* bytes a hash / repeat-distance candidate is compared over */
#endif
#define GZ_NICE      16            /* a run at least this long: the hash candidates are not looked at */

struct GzHuff {                    /* scratch of one code construction (a wave each) */
    uint32_t w[2 * 288];           /* weights: leaves in ascending order, then the internal nodes as they are made */
    uint16_t parent[2 * 288];
    uint16_t order[288];           /* symbols by (frequency, value) ascending */
    uint32_t blcount[18];
};
struct GzDeflate {
    uint8_t  data[65536 + 288];    /* the block (zeros behind it) */
    uint32_t h4[1 << GZ_HB];       /* position + 1 of the latest 4-byte / 6-byte context with this hash (0: none) */
    uint32_t h6[1 << GZ_HB];       /* (both become the output bit buffer once the tokens are made) */
    uint32_t eq[2048 + 8];         /* bit p: data[p] == data[p - 1] */
    uint16_t hk[2 * GZ_SUB];       /* the sub-block's hash keys: 4-byte contexts, then 6-byte ones */
    uint32_t best[GZ_SUB];         /* per position of the sub-block: the best candidate, length << 16 | ~(distance - 1) */
    uint32_t hist_l[288], hist_d[32], hist_b[19];
    uint8_t  len_l[288], len_d[32], len_b[19];
    uint16_t code_l[288], code_d[32], code_b[19];
    uint8_t  rl_sym[320], rl_ext[320];     /* the code lengths in their run-length form (RFC 1951 3.2.7) */
    GzHuff   hf[2];
    uint32_t crc_tab[256], x2n[32], red[GZ_DT];
    int32_t  scan[GZ_DT / 64 + 1];
    int32_t  lastm[2], anyflag, n_rl, hlit, hdist, hclen, use_stored;
    int32_t  rcount[GZ_DT / 64];   /* tokens of each range */
    uint32_t bitpos;               /* bits written so far (header, by lane 0) */
};

/* tokens of the deflate side: literal = byte; match = bit 31 | (length - 3) << 16 | (distance - 1) */

__device__ __forceinline__ int gz_len_sym(int len, int *xb, int *xv)
{   /* length 3..258 -> symbol 257..285 and its extra bits (RFC 1951 3.2.5), by arithmetic: from 11 on every doubling of
     * length - 3 brings four symbols with one more extra bit */
    if (len >= 258) { *xb = 0; *xv = 0; return 285; }
    const int x = len - 3;
    if (x < 8) { *xb = 0; *xv = 0; return 257 + x; }
    const int nb = 31 - __clz(x);                            /* 3..7 */
    const int e = nb - 2;
    *xb = e; *xv = x & ((1 << e) - 1);
    return 257 + 4 * e + 4 + ((x >> e) & 3);
}
__device__ __forceinline__ int gz_dist_sym(int dist, int *xb, int *xv)
{   /* distance 1..32768 -> symbol 0..29: from 5 on every doubling of distance - 1 brings two symbols with one more extra bit */
    const int x = dist - 1;
    if (x < 4) { *xb = 0; *xv = 0; return x; }
    const int nb = 31 - __clz(x);                            /* 2..14 */
    const int e = nb - 1;
    *xb = e; *xv = x & ((1 << e) - 1);
    return 2 * nb + ((x >> e) & 1);
}

/* Optimal prefix-code lengths for n symbols with frequencies freq[] (zeros get length 0), no code longer than maxlen, and the
 * canonical codes in the bit order of the stream.  One wave: the sort by rank over all lanes, the tree and the lengths by
 * lane 0 (two queues; the length limit the way miniz / zlib repair an over-long tree: move leaves down from the deepest
 * level that has room). */
/* the sort: symbol i's rank among the used symbols by (frequency, value) -> H.order / H.w (one thread per symbol: the
 * caller spreads the symbols over as many threads as it has) */
__device__ __forceinline__ void gz_huff_rank(const uint32_t *freq, int n, int i, GzHuff &H)
{
    const uint32_t f = freq[i];
    if (!f) return;
    int r = 0;
    #pragma unroll 8
    for (int j = 0; j < n; j++) { const uint32_t g = freq[j]; r += (g != 0u && (g < f || (g == f && j < i))) ? 1 : 0; }
    H.order[r] = (uint16_t)i; H.w[r] = f;
}
/* ranked = the caller has run gz_huff_rank for every symbol (and synchronised); otherwise this wave ranks them itself */
__device__ __forceinline__ void gz_huff_build(const uint32_t *freq, int n, int maxlen, uint8_t *lens, uint16_t *codes, GzHuff &H, int lane,
                                              bool ranked = false)
{
    for (int i = lane; i < n; i += 64) lens[i] = 0;
    if (lane < 17) H.blcount[lane] = 0;
    WAVE_SYNC();
    int used = 0;
    for (int i0 = 0; i0 < n; i0 += 64) used += __popcll(__ballot(i0 + lane < n && freq[i0 + lane] != 0u));
    if (!ranked) for (int i = lane; i < n; i += 64) gz_huff_rank(freq, n, i, H);
    WAVE_SYNC();
    if (used == 1) { if (lane == 0) lens[H.order[0]] = 1; }
    else if (used >= 2) {
        const int nn = 2 * used - 1;                         /* nodes: leaves 0..used-1 (ascending), then the internal ones as made */
        if (lane == 0) {
            /* two queues: the lighter of the next unused leaf and the next unused internal node, twice, makes a node; the heads'
             * weights stay in registers */
            int la = 0, ia = used, mk = used;
            uint32_t wl = H.w[0], wi = 0xffffffffu;
            for (int k = 0; k < used - 1; k++) {
                uint32_t wsum = 0;
                #pragma unroll
                for (int pick = 0; pick < 2; pick++) {
                    if (la < used && (ia >= mk || wl <= wi)) { wsum += wl; H.parent[la] = (uint16_t)mk; la++; wl = la < used ? H.w[la] : 0xffffffffu; }
                    else { wsum += wi; H.parent[ia] = (uint16_t)mk; ia++; wi = ia < mk ? H.w[ia] : 0xffffffffu; }
                }
                H.w[mk] = wsum;
                if (ia == mk) wi = wsum;                     /* (the node just made is the internal queue's head) */
                mk++;
            }
            H.parent[nn - 1] = (uint16_t)(nn - 1);            /* the root */
        }
        WAVE_SYNC();
        /* every leaf's depth: up its chain of parents (all lanes; a chain is as long as the code, ~20 at the very most) */
        for (int k = lane; k < used; k += 64) {
            int d = 0, x = k;
            while (x != nn - 1) { x = H.parent[x]; d++; }
            if (d > maxlen) d = maxlen;                      /* (counted at maxlen: the repair below makes room) */
            atomicAdd(&H.blcount[d], 1u);
        }
        WAVE_SYNC();
        if (lane == 0) {
            /* the repair: while the code is over-subscribed at maxlen, take a leaf from the deepest level above it that has
             * one and give it two children one level down (miniz tdefl_huffman_enforce_max_code_size) */
            uint32_t total = 0;
            for (int l = maxlen; l > 0; l--) total += (uint32_t)H.blcount[l] << (maxlen - l);
            while (total > (1u << maxlen)) {
                H.blcount[maxlen]--;
                for (int l = maxlen - 1; l > 0; l--) if (H.blcount[l]) { H.blcount[l]--; H.blcount[l + 1] += 2; break; }
                total--;
            }
        }
        WAVE_SYNC();
        /* lengths to symbols: the rarest get the longest -- rank k takes the length whose cumulative count (from maxlen down)
         * first exceeds k */
        for (int k = lane; k < used; k += 64) {
            int acc = 0, l = maxlen;
            for (; l >= 1; l--) { acc += (int)H.blcount[l]; if (k < acc) break; }
            lens[H.order[k]] = (uint8_t)(l < 1 ? 1 : l);
        }
    }
    WAVE_SYNC();
    /* canonical codes (RFC 1951 3.2.2), bit-reversed for the LSB-first stream: a symbol's code = the first code of its length
     * + the number of smaller symbols with that length (ballots over chunks of 64 symbols, the running counts uniform) */
    {
        int cnt[16];
        #pragma unroll
        for (int l = 0; l < 16; l++) cnt[l] = 0;
        for (int i0 = 0; i0 < n; i0 += 64) {
            const int l = i0 + lane < n ? lens[i0 + lane] : 0;
            #pragma unroll
            for (int q = 1; q < 16; q++) cnt[q] += __popcll(__ballot(l == q));
        }
        int first[16];
        int code = 0;
        first[0] = 0;
        #pragma unroll
        for (int l = 1; l < 16; l++) { code = (code + (l > 1 ? cnt[l - 1] : 0)) << 1; first[l] = code; }
        int seen[16];
        #pragma unroll
        for (int l = 0; l < 16; l++) seen[l] = 0;
        for (int i0 = 0; i0 < n; i0 += 64) {
            const int l = i0 + lane < n ? lens[i0 + lane] : 0;
            int mycode = 0;
            #pragma unroll
            for (int q = 1; q < 16; q++) {
                const unsigned long long bm = __ballot(l == q);
                if (l == q) mycode = first[q] + seen[q] + __popcll(bm & ((1ull << lane) - 1ull));
                seen[q] += __popcll(bm);
            }
            if (i0 + lane < n) codes[i0 + lane] = l ? (uint16_t)gz_bitrev((uint32_t)mycode, l) : 0;
        }
    }
    WAVE_SYNC();
}

/* appends nb bits (lane 0 of wave 0 only: the block header) */
__device__ __forceinline__ void gz_putbits(uint32_t *buf, uint32_t &bitpos, uint32_t v, int nb)
{
    if (!nb) return;
    const uint32_t w = bitpos >> 5, sh = bitpos & 31u;
    buf[w] |= v << sh;
    if (sh + (uint32_t)nb > 32u) buf[w + 1] |= v >> (32u - sh);
    bitpos += (uint32_t)nb;
}

#if defined(GZ_STAMP) && !defined(CRUMBLE_EMU)      /* diagnostic build: shader cycles per phase of k_gz_deflate, summed over workgroups */
__device__ unsigned long long gz_stamp[16];
#define GZ_STAMP_AT(k_) do { if (tid == 0) { const long long t_ = (long long)__builtin_amdgcn_s_memtime(); atomicAdd(&gz_stamp[k_], (unsigned long long)(t_ - tlast)); tlast = t_; } } while (0)
#define GZ_STAMP_DECL long long tlast = (long long)__builtin_amdgcn_s_memtime()
#define GZ_STAMP_PB(k_) do { if (wv == 0 && lane == 0) { const long long t_ = (long long)__builtin_amdgcn_s_memtime(); atomicAdd(&gz_stamp[k_], (unsigned long long)(t_ - tpb)); tpb = t_; } } while (0)
#define GZ_STAMP_PB_DECL long long tpb = (long long)__builtin_amdgcn_s_memtime()
#else
#define GZ_STAMP_AT(k_) do { } while (0)
#define GZ_STAMP_DECL do { } while (0)
#define GZ_STAMP_PB(k_) do { } while (0)
#define GZ_STAMP_PB_DECL do { } while (0)
#endif
/* Pass B: the greedy parse, sixteen ranges of 4096 positions at once, a wave each.  A match is cut at the end of its
 * range, so every range's parse starts at its first position and needs nothing from its neighbour (the price: at most
 * fifteen shortened matches per block).  Inside a range the wave goes sub-block by sub-block: lane l holds the four
 * positions l, l + 64, l + 128, l + 192; which of them have a match is four 64-bit ballots; the walk -- literals up to
 * the next position with a match, that match, on to its end -- is scalar code, a match's length fetched by readlane.
 * Tokens go to the range's own 4096 slots of tk[], their number to D.rcount[]. */
__device__ __noinline__ void gz_parse_ranges(GzDeflate &D, uint32_t *__restrict__ tk, const uint32_t *__restrict__ cand, int N, int level, int lane, int wv)
{   /* (a function of its own, not inlined: its sixty-four-bit masks and per-word loops would otherwise raise the register
     * pressure of the candidate loop in front of it, which runs 255 times per block) */
    if (level > 0 && N > 0) {
        const int r0 = wv * GZ_RANGE, r1 = r0 + GZ_RANGE < N ? r0 + GZ_RANGE : N;
        int sp_abs = r0, nt = 0;                            /* next uncovered position, tokens of the range so far (uniform) */
        /* (global pointers by their address space: a function that is not inlined would go through flat instructions, which
         * wait on the LDS counter as well) */
        GZ_GLOBAL(uint32_t) *rtk = (GZ_GLOBAL(uint32_t) *)(tk + r0);
        GZ_GLOBAL(const uint32_t) *gcand = (GZ_GLOBAL(const uint32_t) *)cand;
        GZ_STAMP_PB_DECL;
        /* the candidates of the next sub-block are loaded while this one is walked (they come from HBM: pass A wrote 256 KiB
         * per block and CU; measured 12 k cycles per sub-block waiting for them, more than the walk takes) */
        uint32_t cn[4];
        #pragma unroll
        for (int q = 0; q < 4; q++) { const int p = r0 + lane + 64 * q; cn[q] = p < r1 ? gcand[p] : 0u; }
        for (int base = r0; base < r1; base += GZ_SUB) {
            int ml[4], md[4];
            unsigned long long mm[4], sel[4] = {0ull, 0ull, 0ull, 0ull};
            const int nend = r1 - base < GZ_SUB ? r1 - base : GZ_SUB;
            uint32_t cc[4];
            #pragma unroll
            for (int q = 0; q < 4; q++) cc[q] = cn[q];
            #pragma unroll
            for (int q = 0; q < 4; q++) { const int p = base + GZ_SUB + lane + 64 * q; cn[q] = p < r1 ? gcand[p] : 0u; }
            #pragma unroll
            for (int q = 0; q < 4; q++) {
                const int p = base + lane + 64 * q;
                const uint32_t c = cc[q];
                int l = (int)(c >> 16);
                if (l > r1 - p) l = r1 - p;                  /* cut at the end of the range */
                ml[q] = l >= 3 ? l : 0; md[q] = (int)(c & 0xffffu);
                mm[q] = __ballot(ml[q] >= 3);
            }
            GZ_STAMP_PB(8);
            int sp = sp_abs - base;
            if (sp < 0) sp = 0;
            #pragma unroll
            for (int w = 0; w < 4; w++) {
                const int wend = nend - 64 * w < 64 ? nend - 64 * w : 64;        /* valid positions of this word */
                int lo = sp - 64 * w;
                if (lo < 0) lo = 0;
                while (lo < wend) {
                    const unsigned long long m = mm[w] & (~0ull << lo);
                    if (m == 0ull) {                                               /* literals to the end of the word */
                        sel[w] |= (wend == 64 ? ~0ull : ((1ull << wend) - 1ull)) & (~0ull << lo);
                        lo = wend;
                        break;
                    }
                    const int b = __builtin_ctzll(m);
                    sel[w] |= ((2ull << b) - 1ull) & (~0ull << lo);              /* the literals in front of it, and the match */
                    lo = b + READLANE(ml[w], b);
                }
                if (64 * w + lo > sp) sp = 64 * w + lo;
            }
            if (base + sp > sp_abs) sp_abs = base + sp;
            GZ_STAMP_PB(9);
            /* the tokens in position order: word 0's positions, then word 1's ... (position = 64 * word + lane) */
            const unsigned long long below = (1ull << lane) - 1ull;
            #pragma unroll
            for (int q = 0; q < 4; q++) {
                if ((sel[q] >> lane) & 1ull) {
                    const int p = base + lane + 64 * q;
                    uint32_t tokv;
                    if (ml[q] >= 3) {
                        const int d = md[q] + 1;
                        tokv = 0x80000000u | ((uint32_t)(ml[q] - 3) << 16) | (uint32_t)(d - 1);
                        int xb, xv;
                        atomicAdd(&D.hist_l[gz_len_sym(ml[q], &xb, &xv)], 1u);
                        atomicAdd(&D.hist_d[gz_dist_sym(d, &xb, &xv)], 1u);
                    } else {
                        tokv = D.data[p];
                        atomicAdd(&D.hist_l[tokv], 1u);
                    }
                    rtk[nt + __popcll(sel[q] & below)] = tokv;
                }
                nt += __popcll(sel[q]);
            }
            GZ_STAMP_PB(10);
        }
        if (lane == 0) D.rcount[wv] = r0 < N ? nt : 0;
    } else if (lane == 0) D.rcount[wv] = 0;
}

__global__ void __launch_bounds__(GZ_DT)
k_gz_deflate(int n_blocks, const uint8_t *__restrict__ raw, int64_t raw_len, int level, uint32_t *__restrict__ gtok, uint32_t *__restrict__ gcand,
             uint8_t *__restrict__ slots, uint32_t *__restrict__ sizes, const uint32_t *__restrict__ crc_tab, const uint32_t *__restrict__ x2n)
{
    __shared__ GzDeflate D;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int bi = blockIdx.x; bi < n_blocks; bi += gridDim.x) {
    const int64_t b0 = (int64_t)bi * GZ_MAXBLK;
    const int N = (int)(raw_len - b0 < GZ_MAXBLK ? raw_len - b0 : GZ_MAXBLK);
    const uint8_t *src = raw + b0;
    uint32_t *tk = gtok + (size_t)bi * 65536;
    uint32_t *cand = gcand + (size_t)bi * 65536;            /* per position: best match length << 16 | distance - 1 */
    uint8_t *slot = slots + (size_t)bi * GZ_SLOT;
    __syncthreads();                                          /* (the previous block's stores out of LDS are done) */
    GZ_STAMP_DECL;
    /* ---- the block into LDS, the tables cleared */
    for (int i = tid; i < (65536 + 288) / 4; i += GZ_DT) {
        uint32_t v = 0;
        const int p = 4 * i;
        if (p + 4 <= N) v = ((const GzU32u *)(src + p))->v;
        else for (int k = 0; k < 4; k++) if (p + k < N) v |= (uint32_t)src[p + k] << (8 * k);
        ((uint32_t *)D.data)[i] = v;
    }
    for (int i = tid; i < (1 << GZ_HB); i += GZ_DT) { D.h4[i] = 0u; D.h6[i] = 0u; }
    for (int i = tid; i < 288; i += GZ_DT) D.hist_l[i] = 0u;
    if (tid < 32) D.hist_d[tid] = 0u;
    if (tid < 19) D.hist_b[tid] = 0u;
    if (tid < GZ_SUB) D.best[tid] = 0u;
    if (tid < 256) D.crc_tab[tid] = crc_tab[tid];
    if (tid < 32) D.x2n[tid] = x2n[tid];
    if (tid == 0) { D.lastm[0] = -1; D.lastm[1] = -1; D.bitpos = 0; D.use_stored = 0; }
    __syncthreads();
    const uint32_t crc = gz_crc_block(D.data, N, D.crc_tab, D.x2n, D.red, tid, GZ_DT);
    /* eq[]: bit p set when byte p repeats byte p - 1 */
    for (int i = tid; i < 2048 + 8; i += GZ_DT) {
        uint32_t m = 0;
        for (int k = 0; k < 32; k++) { const int p = 32 * i + k; if (p >= 1 && p < N && D.data[p] == D.data[p - 1]) m |= 1u << k; }
        D.eq[i] = m;
    }
    __syncthreads();
    GZ_STAMP_AT(0);
    int last_dist = 0;                                        /* the distance of a recent match (uniform): the "same distance again" candidate */
    /* Pass A, two workgroup barriers per sub-block of 256 positions: every thread one candidate of one position (four lanes per
     * position), the best of the four into cand[] (global: 256 KiB per block, read again by pass B); then the sub-block's
     * positions into the hash tables.  Nothing here depends on which matches the parse will take. */
#ifdef GZ_EXP_NOA                                             /* (diagnostic builds, tools/build_variant.sh: what a phase costs is what the kernel gains without it) */
    for (int i = tid; i < N; i += GZ_DT) cand[i] = 0u;
    if (false)
#else
    if (level > 0 && N > 0)
#endif
    for (int base = 0; base < N; base += GZ_SUB) {
        /* a WAVE per candidate kind and 64 positions (wave w: kind w & 3, positions 64 * (w >> 2) + lane): no lane computes
         * what only another kind needs; the four kinds of a position meet in an LDS word by atomicMax on
         * length << 16 | ~(distance - 1) -- the longest wins, then the nearest */
        const int cnd = wv & 3, j = ((wv >> 2) << 6) + lane;
        const int P = base + j;
        const int maxl = N - P < 258 ? N - P : 258;
        if (P < N) {
            const uint32_t *dw = (const uint32_t *)D.data;
            if (cnd == 2) {
                /* how far the byte in front of P repeats from P on: the consecutive ones of eq[] from bit P */
                int run = 0, p = P;
                while (run < maxl) {
                    const uint32_t w = ~(D.eq[p >> 5] >> (p & 31));         /* first zero = end of the run */
                    const int avail = 32 - (p & 31);
                    const int z = w ? __builtin_ctz(w) : 32;
                    if (z < avail) { run += z; break; }
                    run += avail; p += avail;
                }
                if (run > maxl) run = maxl;
                if (run >= 3 && P >= 1) atomicMax(&D.best[j], ((uint32_t)run << 16) | 0xffffu);
            } else {
                /* inside a run of GZ_NICE or more the run's own match is good enough: sixteen one bits of eq[] from bit P */
                const uint32_t e0 = D.eq[P >> 5], e1 = D.eq[(P >> 5) + 1];
                const uint64_t eqw = (((uint64_t)e1 << 32) | e0) >> (P & 31);
                const bool inrun = ((uint32_t)eqw & ((1u << GZ_NICE) - 1u)) == ((1u << GZ_NICE) - 1u);
                const int pw = P >> 2, psh = 8 * (P & 3);
                const uint32_t w0 = dw[pw], w1 = dw[pw + 1], w2 = dw[pw + 2];
                const uint64_t two = ((uint64_t)w1 << 32) | w0;
                const uint64_t b8 = psh ? (two >> psh) | ((uint64_t)w2 << (64 - psh)) : two;      /* the eight bytes from P on */
                int cand_p = -1;
                if (cnd == 0) {
                    const uint32_t k4 = ((uint32_t)b8 * 2654435761u) >> (32 - GZ_HB);
                    D.hk[j] = (uint16_t)(k4 | (inrun ? 0x8000u : 0u));
                    if (!inrun && maxl >= 4) cand_p = (int)D.h4[k4] - 1;
                } else if (cnd == 1) {
                    const uint32_t k6 = (uint32_t)(((b8 & 0xffffffffffffull) * 0x9E3779B97F4A7C15ull) >> (64 - GZ_HB));
                    D.hk[GZ_SUB + j] = (uint16_t)(k6 | (inrun ? 0x8000u : 0u));
                    if (!inrun && maxl >= 6) cand_p = (int)D.h6[k6] - 1;
                } else if (!inrun && last_dist > 1 && P - last_dist >= 0) cand_p = P - last_dist;
#ifdef GZ_EXP_NOCMP
                cand_p = -1;
#endif
                if (cand_p >= 0 && P - cand_p <= 32768) {
                    /* common prefix of the bytes at cand_p and at P, eight at a time (aligned words, shifted into place) */
                    const int cw = cand_p >> 2, csh = 8 * (cand_p & 3);
                    const int cmax = maxl < GZ_CMPMAX ? maxl : GZ_CMPMAX;
                    int l = 0;
                    uint64_t pv = b8;
                    for (int it = 0; l < cmax; it++) {
                        const uint32_t c0 = dw[cw + 2 * it], c1 = dw[cw + 2 * it + 1], c2 = dw[cw + 2 * it + 2];
                        const uint64_t ct = ((uint64_t)c1 << 32) | c0;
                        const uint64_t cv = csh ? (ct >> csh) | ((uint64_t)c2 << (64 - csh)) : ct;
                        if (it) {
                            const uint32_t q0 = dw[pw + 2 * it], q1 = dw[pw + 2 * it + 1], q2 = dw[pw + 2 * it + 2];
                            const uint64_t qt = ((uint64_t)q1 << 32) | q0;
                            pv = psh ? (qt >> psh) | ((uint64_t)q2 << (64 - psh)) : qt;
                        }
                        const uint64_t x = cv ^ pv;
                        if (x) { l += __builtin_ctzll(x) >> 3; break; }
                        l += 8;
                    }
                    if (l > cmax) l = cmax;
                    if (l >= 4) atomicMax(&D.best[j], ((uint32_t)l << 16) | (0xffffu - (uint32_t)(P - cand_p - 1)));
                }
            }
        }
#ifndef GZ_EXP_NOBAR1
        GZ_LDS_BARRIER();                                                                   /* ---- 1 */
#endif
        GZ_STAMP_AT(1);
        {
            if (tid < GZ_SUB) {                              /* the winners: to cand[] (pass B reads them), the latest one's distance on */
                const uint32_t v = D.best[tid];
                D.best[tid] = 0u;
                const int bl = (int)(v >> 16), bd1 = (int)(0xffffu - (v & 0xffffu));        /* length, distance - 1 */
                if (base + tid < N) cand[base + tid] = bl >= 3 ? ((uint32_t)bl << 16) | (uint32_t)bd1 : 0u;
                /* the latest position with a match, and its distance: one atomic per wave (its highest such lane), not one per
                 * lane on one word */
                const unsigned long long hm = __ballot(bl >= 3);
                if (hm && lane == 63 - __builtin_clzll(hm)) atomicMax(&D.lastm[(base / GZ_SUB) & 1], (tid << 16) | bd1);
            } else if (tid < 3 * GZ_SUB) {
                /* the hash tables take the sub-block's positions (the largest position wins: deterministic) */
                const int t = tid - GZ_SUB, jj = t >> 1, PP = base + jj;
                /* (not the positions inside a run: they all hash alike -- sixty-four atomics on one word -- and nothing looks
                 * a run's inside up: bit 15 of the key) */
                const uint32_t hkv = D.hk[(t & 1) ? GZ_SUB + jj : jj];
                if (!(hkv & 0x8000u)) {
                    if (t & 1) { if (PP + 6 <= N) atomicMax(&D.h6[hkv], (uint32_t)PP + 1u); }
                    else       { if (PP + 4 <= N) atomicMax(&D.h4[hkv], (uint32_t)PP + 1u); }
                }
            }
            if (tid == 0) D.lastm[((base / GZ_SUB) + 1) & 1] = -1;      /* (the other one: the next sub-block's, untouched since its readers two barriers ago) */
        }
#ifndef GZ_EXP_NOBAR2
        GZ_LDS_BARRIER();                                                                   /* ---- 2 */
#endif
        GZ_STAMP_AT(2);
        {
            const int lm = D.lastm[(base / GZ_SUB) & 1];
            if (lm >= 0) last_dist = (lm & 0xffff) + 1;
        }
    }
    __syncthreads();
#ifdef GZ_EXP_NOPARSE
    if (lane == 0) D.rcount[wv] = 0;
#else
    gz_parse_ranges(D, tk, cand, N, level, lane, wv);
#endif
    __syncthreads();
    /* ---- the codes */
    if (tid == 0) D.hist_l[256] = 1u;                        /* end of block */
    __syncthreads();
    if (level > 0) {                                          /* the two sorts by every thread that has a symbol, the trees by a wave each */
        if (tid < 286) gz_huff_rank(D.hist_l, 286, tid, D.hf[0]);
        else if (tid >= 320 && tid < 350) gz_huff_rank(D.hist_d, 30, tid - 320, D.hf[1]);
    }
    __syncthreads();
#ifdef GZ_EXP_NOHUFF                                          /* (diagnostic: flat 9- and 5-bit codes instead of the two trees) */
    if (level > 0) {
        for (int i = tid; i < 286; i += GZ_DT) { D.len_l[i] = 9; D.code_l[i] = (uint16_t)gz_bitrev((uint32_t)i, 9); }
        if (tid < 30) { D.len_d[tid] = 5; D.code_d[tid] = (uint16_t)gz_bitrev((uint32_t)tid, 5); }
    }
#else
    if (level > 0) {
        if (wv == 0) gz_huff_build(D.hist_l, 286, 15, D.len_l, D.code_l, D.hf[0], lane, true);
        if (wv == 1) gz_huff_build(D.hist_d, 30, 15, D.len_d, D.code_d, D.hf[1], lane, true);
    }
#endif
    __syncthreads();
    GZ_STAMP_AT(4);
    /* the output bit buffer: the hash tables' 64 KiB */
    uint32_t *obuf = D.h4;
    for (int i = tid; i < 2 * (1 << GZ_HB); i += GZ_DT) obuf[i] = 0u;
    __syncthreads();
    if (level > 0 && wv == 0) {
        {   /* HLIT / HDIST: the last used length and distance symbol (257 and 1 at least) */
            const unsigned long long ul = __ballot(lane < 29 && D.len_l[257 + (lane < 29 ? lane : 0)] != 0);
            const unsigned long long ud = __ballot(lane < 30 && D.len_d[lane < 30 ? lane : 0] != 0);
            const int hlit = 257 + (ul ? 64 - __builtin_clzll(ul) : 0);
            int hdist = ud ? 64 - __builtin_clzll(ud) : 1;
            /* at least one distance code must be given a length (zlib refuses a block that has matches and none; one unused
             * code of one bit is what zlib itself writes for a block without matches) */
            if (lane == 0) {
                if (!ud) { D.len_d[0] = 1; D.code_d[0] = 0; hdist = 1; }
                D.hlit = hlit; D.hdist = hdist;
            }
            WAVE_SYNC();
        }
        {   /* the lengths as runs: 16 = repeat the previous length 3-6 times, 17 = 3-10 zeros, 18 = 11-138 zeros.  Every lane
             * looks at five of the (at most 316) lengths; a lane whose length differs from the one in front of it starts a run,
             * learns the run's length from the next start, counts the symbols the run becomes and writes them at the prefix sum
             * of the counts (the serial loop this replaces took as long as pass B) */
            const int hlit = D.hlit, tot = D.hlit + D.hdist;
            uint16_t *spos = (uint16_t *)D.red;              /* run starts in order (scratch: the CRC's reduction words) */
            uint8_t *sval = (uint8_t *)(spos + 320);
            int before = 0;
            for (int c = 0; c < 5; c++) {
                const int i = 64 * c + lane;
                const int v = i < tot ? (i < hlit ? D.len_l[i] : D.len_d[i - hlit]) : -1;
                const int pv = i == 0 || i >= tot ? -2 : (i - 1 < hlit ? D.len_l[i - 1] : D.len_d[i - 1 - hlit]);
                const bool st = i < tot && v != pv;
                const unsigned long long sm = __ballot(st);
                if (st) { const int k = before + __popcll(sm & ((1ull << lane) - 1ull)); spos[k] = (uint16_t)i; sval[k] = (uint8_t)v; }
                before += __popcll(sm);
            }
            const int ns = before;
            WAVE_SYNC();
            int obase = 0;
            for (int k0 = 0; k0 < ns; k0 += 64) {
                const int k = k0 + lane;
                int v = 0, r = 0, cnt = 0;
                if (k < ns) {
                    v = sval[k];
                    r = (k + 1 < ns ? (int)spos[k + 1] : tot) - (int)spos[k];
                    if (v == 0 && r >= 3) { int left = r; while (left >= 3) { left -= left > 138 ? 138 : left; cnt++; } cnt += left; }
                    else if (v != 0 && r >= 4) { int left = r - 1; cnt = 1; while (left >= 3) { left -= left > 6 ? 6 : left; cnt++; } cnt += left; }
                    else cnt = r;
                }
                int incl = cnt;
                for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
                int o = obase + incl - cnt;
#define GZ_RL(sym_, ext_) do { D.rl_sym[o] = (uint8_t)(sym_); D.rl_ext[o] = (uint8_t)(ext_); atomicAdd(&D.hist_b[sym_], 1u); o++; } while (0)
                if (k < ns) {
                    if (v == 0 && r >= 3) {
                        int left = r;
                        while (left >= 3) { const int take = left > 138 ? 138 : left; if (take >= 11) GZ_RL(18, take - 11); else GZ_RL(17, take - 3); left -= take; }
                        while (left-- > 0) GZ_RL(0, 0);
                    } else if (v != 0 && r >= 4) {
                        GZ_RL(v, 0);
                        int left = r - 1;
                        while (left >= 3) { const int take = left > 6 ? 6 : left; GZ_RL(16, take - 3); left -= take; }
                        while (left-- > 0) GZ_RL(v, 0);
                    } else for (int q = 0; q < r; q++) GZ_RL(v, 0);
                }
#undef GZ_RL
                obase += READLANE(incl, 63);
            }
            if (lane == 0) D.n_rl = obase;
        }
        WAVE_SYNC();
        gz_huff_build(D.hist_b, 19, 7, D.len_b, D.code_b, D.hf[0], lane);
        {
            const int order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            /* HCLEN: the code-length code's lengths in their fixed order, trailing zeros dropped (at least four stay) */
            const unsigned long long nz = __ballot(lane < 19 && D.len_b[order[lane < 19 ? lane : 0]] != 0);
            int hclen = nz ? 64 - __builtin_clzll(nz) : 0;
            if (hclen < 4) hclen = 4;
            const int n_rl = D.n_rl;
            /* the size of the dynamic block against a stored one: every lane adds up its share of the symbols */
            uint32_t bits = 0;
            for (int k = lane; k < n_rl; k += 64) { const int sy = D.rl_sym[k]; bits += D.len_b[sy] + (sy == 16 ? 2u : sy == 17 ? 3u : sy == 18 ? 7u : 0u); }
            for (int sy = lane; sy < 286; sy += 64) {
                const int xb = (sy < 265 || sy == 285) ? 0 : (sy - 265) / 4 + 1;      /* extra bits of a length symbol (RFC 1951 3.2.5) */
                bits += D.hist_l[sy] * (uint32_t)(D.len_l[sy] + xb);
            }
            if (lane < 30) bits += D.hist_d[lane] * (uint32_t)(D.len_d[lane] + (lane < 4 ? 0 : (lane >> 1) - 1));
            for (int d = 32; d > 0; d >>= 1) bits += (uint32_t)__shfl_xor((int)bits, d);
            bits += 3u + 14u + 3u * (uint32_t)hclen;
            const bool stored = (bits + 7u) / 8u >= (uint32_t)N + 5u;
            if (lane == 0) { D.hclen = hclen; D.use_stored = stored ? 1 : 0; }
            if (!stored) {
                /* the header's bits, each piece placed by its own lane: the fixed 17 bits, the HCLEN 3-bit lengths, then the
                 * run-length symbols at the prefix sum of their widths */
                if (lane == 0) { uint32_t bp = 0; gz_putbits(obuf, bp, 1u | (2u << 1), 3); gz_putbits(obuf, bp, (uint32_t)(D.hlit - 257), 5);
                                 gz_putbits(obuf, bp, (uint32_t)(D.hdist - 1), 5); gz_putbits(obuf, bp, (uint32_t)(hclen - 4), 4); }
                WAVE_SYNC();
                if (lane < hclen) {
                    const uint32_t bp = 17u + 3u * (uint32_t)lane, v = D.len_b[order[lane]];
                    atomicOr(&obuf[bp >> 5], v << (bp & 31u));
                    if ((bp & 31u) + 3u > 32u) atomicOr(&obuf[(bp >> 5) + 1], v >> (32u - (bp & 31u)));
                }
                uint32_t bitbase = 17u + 3u * (uint32_t)hclen;
                for (int k0 = 0; k0 < n_rl; k0 += 64) {
                    const int k = k0 + lane;
                    uint32_t v = 0; int nb = 0;
                    if (k < n_rl) {
                        const int sy = D.rl_sym[k];
                        v = D.code_b[sy]; nb = D.len_b[sy];
                        if (sy >= 16) { v |= (uint32_t)D.rl_ext[k] << nb; nb += sy == 16 ? 2 : sy == 17 ? 3 : 7; }
                    }
                    int incl = nb;
                    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
                    if (nb) {
                        const uint32_t bp = bitbase + (uint32_t)(incl - nb);
                        atomicOr(&obuf[bp >> 5], v << (bp & 31u));
                        if ((bp & 31u) + (uint32_t)nb > 32u) atomicOr(&obuf[(bp >> 5) + 1], v >> (32u - (bp & 31u)));
                    }
                    bitbase += (uint32_t)READLANE(incl, 63);
                }
                if (lane == 0) D.bitpos = bitbase;
            }
        }
    }
    if (level <= 0 && tid == 0) D.use_stored = 1;
    __syncthreads();
    GZ_STAMP_AT(5);
    uint32_t payload;                                         /* bytes of the DEFLATE stream */
    if (D.use_stored) {
        /* one stored block: 01 LEN NLEN bytes */
        payload = 5u + (uint32_t)N;
        if (tid == 0) {
            uint8_t *q = slot + 18;
            q[0] = 1; q[1] = (uint8_t)N; q[2] = (uint8_t)(N >> 8); q[3] = (uint8_t)~N; q[4] = (uint8_t)(~N >> 8);
        }
        for (int i = tid; i < N; i += GZ_DT) slot[18 + 5 + i] = D.data[i];
    } else {
        /* ---- every token's bits at its place: prefix sum of the bit counts, OR into the buffer */
        uint32_t bitbase = D.bitpos;
#ifdef GZ_EXP_NOENC
        for (int rg = GZ_DT / 64; rg <= GZ_DT / 64; rg++) {
#else
        for (int rg = 0; rg <= GZ_DT / 64; rg++) {           /* the ranges in order; one more round for the end-of-block code */
#endif
            const bool last = rg == GZ_DT / 64;
            const int rn = last ? 1 : D.rcount[rg];
            const uint32_t *rtk = tk + (last ? 0 : rg * GZ_RANGE);
            for (int c = 0; c < rn; c += GZ_DT) {
                const int i = c + tid;
                uint64_t v = 0; int nb = 0;
                if (last) { if (tid == 0) { v = D.code_l[256]; nb = D.len_l[256]; } }
                else if (i < rn) {
                    const uint32_t t = rtk[i];
                    if (t >> 31) {
                        const int l = (int)((t >> 16) & 0xffu) + 3, d = (int)(t & 0x7fffu) + 1;
                        int xb, xv;
                        const int ls = gz_len_sym(l, &xb, &xv);
                        v = D.code_l[ls]; nb = D.len_l[ls];
                        v |= (uint64_t)xv << nb; nb += xb;
                        const int ds = gz_dist_sym(d, &xb, &xv);
                        v |= (uint64_t)D.code_d[ds] << nb; nb += D.len_d[ds];
                        v |= (uint64_t)xv << nb; nb += xb;
                    } else { v = D.code_l[t & 0xffu]; nb = D.len_l[t & 0xffu]; }
                }
                int incl = nb;
                for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
                if (lane == 63) D.scan[wv] = incl;
                __syncthreads();
                uint32_t wbase = 0, total = 0;
                for (int w = 0; w < GZ_DT / 64; w++) { const uint32_t sc = (uint32_t)D.scan[w]; if (w < wv) wbase += sc; total += sc; }
                if (nb) {
                    const uint32_t bp = bitbase + wbase + (uint32_t)(incl - nb);
                    const uint32_t w = bp >> 5, sh = bp & 31u;
                    atomicOr(&obuf[w], (uint32_t)(v << sh));
                    if (sh + (uint32_t)nb > 32u) atomicOr(&obuf[w + 1], (uint32_t)(v >> (32u - sh)));
                    if (sh + (uint32_t)nb > 64u) atomicOr(&obuf[w + 2], (uint32_t)(v >> (64u - sh)));
                }
                bitbase += total;
                __syncthreads();
            }
        }
        payload = (bitbase + 7u) >> 3;
        for (uint32_t i = tid; i < payload; i += GZ_DT) slot[18 + i] = ((const uint8_t *)obuf)[i];
    }
    GZ_STAMP_AT(6);
    if (tid == 0) {
        const uint32_t total = 18u + payload + 8u;
        const uint8_t hdr[18] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, (uint8_t)((total - 1u) & 0xffu), (uint8_t)((total - 1u) >> 8)};
        for (int k = 0; k < 18; k++) slot[k] = hdr[k];
        uint8_t *q = slot + 18 + payload;
        for (int k = 0; k < 4; k++) { q[k] = (uint8_t)(crc >> (8 * k)); q[4 + k] = (uint8_t)((uint32_t)N >> (8 * k)); }
        sizes[bi] = total;
    }
    }
}

/* the members of a deflate call side by side: out[off[b] ...] = slot b's bytes */
__global__ void __launch_bounds__(256)
k_gz_gather(int n_blocks, const uint8_t *__restrict__ slots, const uint32_t *__restrict__ sizes, const int64_t *__restrict__ off,
            uint8_t *__restrict__ out)
{
    const int b = blockIdx.x;
    if (b >= n_blocks) return;
    const uint8_t *s = slots + (size_t)b * GZ_SLOT;
    uint8_t *d = out + off[b];
    const uint32_t n = sizes[b], n4 = n >> 2;
    for (uint32_t i = threadIdx.x; i < n4; i += 256) ((GzU32u *)d)[i].v = ((const uint32_t *)s)[i];
    for (uint32_t i = (n4 << 2) + threadIdx.x; i < n; i += 256) d[i] = s[i];
}
/* exclusive prefix sum of the members' sizes (one workgroup; n up to a few hundred thousand) */
__global__ void __launch_bounds__(1024)
k_gz_offsets(int n, const uint32_t *__restrict__ sizes, int64_t *__restrict__ off)
{
    __shared__ long long part[1024];
    const int tid = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int lo = tid * per < n ? tid * per : n, hi = lo + per < n ? lo + per : n;
    long long s = 0;
    for (int i = lo; i < hi; i++) s += sizes[i];
    part[tid] = s;
    __syncthreads();
    if (tid == 0) { long long a = 0; for (int i = 0; i < 1024; i++) { const long long v = part[i]; part[i] = a; a += v; } off[n] = a; }
    __syncthreads();
    long long a = part[tid];
    for (int i = lo; i < hi; i++) { off[i] = a; a += sizes[i]; }
}

#endif /* CRUMBLE_GZ_KERNELS_H */
