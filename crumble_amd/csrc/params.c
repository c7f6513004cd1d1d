/*
 * params.c -- host-side parameter block of the engine: defaults, presets,
 * -y technology profiles, -k/-K quality lists, init_bins and BED collapsing.
 * Mirrors the option handling of the reference's main()
 * (/root/reference/snp_score.c:2144-2542) and bed.c:20-40 so that a caller who
 * switches from crumble's getopt loop to this ABI gets identical settings.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <unistd.h>
#include <sched.h>
#include "../../include/crumble_hip.h"
#include "team.h"

/* main()'s initialiser, snp_score.c:2152-2192 (== preset -9) */
void crumble_params_default(crumble_params *p)
{
    memset(p, 0, sizeof(*p));
    p->reduce_qual   = 1;      /* REDUCE_QUAL  snp:103 */
    p->binary_qual   = 0;      /* BINARY_QUAL  snp:104 */
    p->iSTR_mul      = 1.0;    /* I_STR_MUL    snp:128 */
    p->iSTR_add      = 2;      /* I_STR_ADD    snp:129 */
    p->sSTR_mul      = 0.0;    /* S_STR_MUL    snp:125 */
    p->sSTR_add      = 0;      /* S_STR_ADD    snp:126 */
    p->qlow          = 5;      /* QL           snp:94  */
    p->qcutoff       = 25;     /* QM           snp:95  */
    p->qhigh         = 40;     /* QH           snp:96  */
    p->qcap          = 60;     /* QCAP         snp:97  */
    p->min_mqual     = 0;      /* MIN_MQUAL    snp:100 */
    p->min_qual_A    = 0;      /* MIN_QUAL_A   snp:110 */
    p->min_indel_A   = 50;     /* MIN_INDEL_A  snp:111 */
    p->min_discrep_A = 2.0;    /* MIN_DISCREP_A snp:112 */
    p->min_qual_B    = 70;     /* MIN_QUAL_B   snp:115 */
    p->min_indel_B   = 125;    /* MIN_INDEL_B  snp:116 */
    p->min_discrep_B = 1.5;    /* MIN_DISCREP_B snp:117 */
    p->indel_fract   = 0.0;    /* INDEL_FRACT  snp:107 */
    p->clip_perc     = 0.2;    /* CLIP_PERC    snp:139 */
    p->low_mqual_perc= 1.0;    /* LOW_MQUAL_PERC snp:133 */
    p->ins_len_perc  = 1.0;    /* INS_LEN_PERC snp:136 */
    p->over_depth    = 999.0;  /* OVER_DEPTH   snp:142 */
    p->indel_ov_perc = 0.0;    /* INDEL_OVERLAP_PERC snp:145 */
    p->pblock        = 8;      /* PBLOCK       snp:147 */
    p->softclip      = 0;
    p->perfect_col   = 0;
}

/* The 13 fields every preset overwrites (snp_score.c:2380-2482); columns are
 * p Q D X M Z V P s_mul s_add i_mul i_add m. */
int crumble_params_preset(crumble_params *p, int level)
{
    struct row { int lvl, pb, Q, D; double X, M, Z, V, P, smul; int sadd; double imul; int iadd, m; };
    static const struct row rows[] = {
        {9, 8, 70, 125, 1.5, 1.0, 1.0, 0.0, 999.0, 0.0, 0, 1.0, 2, 0},
        {8, 0, 70, 125, 1.5, 1.0, 1.0, 0.0, 999.0, 0.0, 0, 1.0, 2, 0},
        {7, 0, 75, 150, 1.0, 1.0, 1.0, 0.0, 999.0, 0.0, 0, 1.1, 2, 0},
        {5, 0, 75, 150, 1.0, 0.5, 0.1, 0.5,   3.0, 0.0, 0, 1.1, 2, 0},
        {3, 0, 75, 150, 1.0, 0.5, 0.1, 0.5,   3.0, 1.0, 0, 1.1, 2, 0},
        {1, 0, 75, 150, 1.0, 0.5, 0.1, 0.5,   3.0, 1.0, 5, 2.0, 1, 5},
    };
    size_t i;
    for (i = 0; i < sizeof(rows) / sizeof(rows[0]); i++) {
        const struct row *r = &rows[i];
        if (r->lvl != level) continue;
        p->pblock = r->pb;  p->min_qual_B = r->Q;  p->min_indel_B = r->D;
        p->min_discrep_B = r->X;  p->low_mqual_perc = r->M;  p->ins_len_perc = r->Z;
        p->indel_ov_perc = r->V;  p->over_depth = r->P;
        p->sSTR_mul = r->smul;  p->sSTR_add = r->sadd;
        p->iSTR_mul = r->imul;  p->iSTR_add = r->iadd;  p->min_mqual = r->m;
        return CRUMBLE_OK;
    }
    return CRUMBLE_EINVAL;   /* -2 -4 -6 do not exist (getopt string snp:2199) */
}

/* -y (snp_score.c:2316-2328).  The code, not the man page, is authoritative:
 * pbccs sets qcutoff (not min_mqual) to 40. */
int crumble_params_tech(crumble_params *p, const char *tech)
{
    if (!tech) return CRUMBLE_EINVAL;
    if (strcasecmp(tech, "illumina") == 0) return CRUMBLE_OK;
    if (strcasecmp(tech, "pbccs") == 0) {
        p->indel_fract = 0.1;
        p->min_discrep_B = 0.8;
        p->qcutoff = 40;
        p->qhigh = 60;
        p->pblock = 16;
        p->perfect_col = 1;
        p->preserve_qual[93] = 1;
        return CRUMBLE_OK;
    }
    return CRUMBLE_OK;       /* unknown names are silently ignored, as upstream */
}

/* -k / -K  "q1[-q2][,...]" (snp_score.c:2362-2375) */
int crumble_params_keep_quals(crumble_params *p, const char *list, int really)
{
    char *endp;
    if (!list) return CRUMBLE_EINVAL;
    endp = (char *)list;
    do {
        long q1 = strtol(endp, &endp, 10), q2 = q1;
        if (*endp == '-') q2 = strtol(endp + 1, &endp, 10);
        do {
            long q = q1 < 0 ? 0 : (q1 > 255 ? 255 : q1);
            p->preserve_qual[q] = 1 + (really != 0);
        } while (++q1 <= q2);
    } while (*endp++ == ',');
    return CRUMBLE_OK;
}

/* init_bins (snp_score.c:234-247) */
void crumble_init_bins(const crumble_params *p, uint8_t bin2[256])
{
    int i;
    for (i = 0; i < p->qcutoff && i < 256; i++) bin2[i] = (uint8_t)p->qlow;
    for (; i < 256; i++) bin2[i] = (uint8_t)p->qhigh;
    for (i = 0; i < 256; i++)
        if (p->preserve_qual[i] > 1) bin2[i] = (uint8_t)i;
}

/* bed_collapse (bed.c:20-40): sort by (tid,start); a region starts a new entry
 * when it begins beyond the end of the PREVIOUS INPUT region (not of the merged
 * one) -- reproduced literally, because the monotone cursor of the column loop
 * (snp_score.c:1443-1463) observes the difference.  The reference then copies
 * one element past the input as an extra region (uninitialised memory,
 * SURVEY.md Q15); that phantom entry is not reproduced. */
typedef struct { int32_t tid, start, end; } bed3;
static int bed3_cmp(const void *a, const void *b)
{
    const bed3 *x = a, *y = b;
    if (x->tid != y->tid) return x->tid - y->tid;
    return x->start - y->start;
}
int crumble_bed_collapse(int32_t *tid, int32_t *start, int32_t *end, int n)
{
    bed3 *r;
    int i, j, last_tid = -1, last_end = -1;
    if (n <= 0) return 0;
    r = malloc(n * sizeof(*r));
    if (!r) return CRUMBLE_ENOMEM;
    for (i = 0; i < n; i++) { r[i].tid = tid[i]; r[i].start = start[i]; r[i].end = end[i]; }
    qsort(r, n, sizeof(*r), bed3_cmp);
    for (i = j = 0; i < n; i++) {
        if (r[i].tid > last_tid || r[i].start > last_end) r[j++] = r[i];
        else if (r[i].end > r[j - 1].end) r[j - 1].end = r[i].end;
        last_tid = r[i].tid;
        last_end = r[i].end;
    }
    for (i = 0; i < j; i++) { tid[i] = r[i].tid; start[i] = r[i].start; end[i] = r[i].end; }
    free(r);
    return j;
}

const char *crumble_hip_strerror(int err)
{
    switch (err) {
    case CRUMBLE_OK:        return "ok";
    case CRUMBLE_EINVAL:    return "invalid argument or unsupported parameter";
    case CRUMBLE_ENOMEM:    return "out of memory";
    case CRUMBLE_EHIP:      return "HIP runtime error";
    case CRUMBLE_EUNSORTED: return "reads are not coordinate sorted";
    case CRUMBLE_EOVERFLOW: return "internal work list overflow";
    case CRUMBLE_ENODEVICE: return "no usable HIP device (there is no CPU fallback)";
    case CRUMBLE_ERCCL:     return "RCCL error";
    default:                return "unknown error";
    }
}

int crumble_hip_abi_version(void) { return CRUMBLE_HIP_ABI; }
size_t crumble_hip_struct_size(int which)
{
    switch (which) {
    case 0: return sizeof(crumble_batch);
    case 1: return sizeof(crumble_params);
    case 2: return sizeof(crumble_coded);
    case 3: return sizeof(crumble_window);
    case 4: return sizeof(crumble_carry);
    case 5: return sizeof(crumble_column);
    default: return 0;
    }
}

/* ---- BGZF member walk (SAMv1 4.1; RFC 1952 header with the BC extra subfield): host side of crumble_gz_inflate ---- */
int64_t crumble_bgzf_scan(const uint8_t *comp, int64_t comp_len, crumble_gz_block *blk, int64_t max_blocks,
                          int64_t *consumed, int64_t *raw_total)
{
    int64_t off = 0, nb = 0, utot = 0;
    while (off + 18 <= comp_len && (!blk || nb < max_blocks)) {
        const uint8_t *h = comp + off;
        if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) return -1;
        const unsigned xlen = (unsigned)h[10] | (unsigned)h[11] << 8;
        if (off + 12 + (int64_t)xlen > comp_len) break;
        int bsize = -1;
        for (unsigned x = 0; x + 4 <= xlen; ) {
            const uint8_t *sf = h + 12 + x;
            const unsigned slen = (unsigned)sf[2] | (unsigned)sf[3] << 8;
            if (sf[0] == 'B' && sf[1] == 'C' && slen == 2 && x + 6 <= xlen) bsize = sf[4] | sf[5] << 8;
            x += 4 + slen;
        }
        if (bsize < 0) return -1;
        const int64_t total = (int64_t)bsize + 1;
        if (total < 12 + (int64_t)xlen + 8) return -1;
        if (off + total > comp_len) break;
        const uint8_t *t = h + total - 8;
        const uint32_t crc = (uint32_t)t[0] | (uint32_t)t[1] << 8 | (uint32_t)t[2] << 16 | (uint32_t)t[3] << 24;
        const uint32_t isz = (uint32_t)t[4] | (uint32_t)t[5] << 8 | (uint32_t)t[6] << 16 | (uint32_t)t[7] << 24;
        if (isz > 65536u) return -1;
        if (blk) {
            blk[nb].coff = off + 12 + xlen; blk[nb].clen = (uint32_t)(total - 12 - xlen - 8);
            blk[nb].uoff = utot; blk[nb].ulen = isz; blk[nb].crc = crc; blk[nb].pad_ = 0;
        }
        utot += isz; nb++; off += total;
    }
    if (consumed) *consumed = off;
    if (raw_total) *raw_total = utot;
    return nb;
}

static int host_threads_(void);
/* one nt16 code per byte -> BAM's packing in the crumble_batch.bases4 layout (include/crumble_hip.h) */
typedef struct { const int64_t *seq_off; const uint8_t *bases; uint8_t *out; } pack4_job;
static void pack4_reads(void *ctx, int64_t lo, int64_t hi, int tid)
{
    const pack4_job *j = ctx;
    (void)tid;
    for (int64_t i = lo; i < hi; i++) {
        const uint8_t *b = j->bases + j->seq_off[i];
        const int64_t lq = j->seq_off[i + 1] - j->seq_off[i];
        uint8_t *o = j->out + CRUMBLE_BASES4_OFF(j->seq_off[i], i);
        int64_t k = 0;
        for (; k + 1 < lq; k += 2) *o++ = (uint8_t)((b[k] & 15) << 4 | (b[k + 1] & 15));
        if (k < lq) *o = (uint8_t)((b[k] & 15) << 4);
    }
}
void crumble_pack_bases4(int64_t n_reads, const int64_t *seq_off, const uint8_t *bases, uint8_t *out)
{
    pack4_job j = { seq_off, bases, out };
    team_run_once(host_threads_(), n_reads, 4096, 0, pack4_reads, &j);
}

/* ---- coded upload, host side (include/crumble_hip.h, crumble_batch.qualsp): quality bytes -> 2- or 4-bit codes --- */
typedef struct { const uint8_t *q; int64_t n; unsigned char (*seen)[256]; } qseen_job;
static void qseen_piece(void *ctx, int64_t lo, int64_t hi, int tid)
{
    qseen_job *j = (qseen_job *)ctx;
    unsigned char *s = j->seen[tid];
    const uint8_t *q = j->q;
    int64_t i = lo;
    for (; i + 8 <= hi; i += 8) {                                      /* which byte values occur: a store per byte, no branch */
        s[q[i]] = 1; s[q[i + 1]] = 1; s[q[i + 2]] = 1; s[q[i + 3]] = 1;
        s[q[i + 4]] = 1; s[q[i + 5]] = 1; s[q[i + 6]] = 1; s[q[i + 7]] = 1;
    }
    for (; i < hi; i++) s[q[i]] = 1;
}
/* two quality bytes -> their two codes (code of the first | code of the second << bits): one look-up per pair */
typedef struct { const uint8_t *q; uint8_t *out; int64_t n; int bits; uint8_t map[256]; uint8_t *pair; } qpack_job;
static void qpack_piece(void *ctx, int64_t lo, int64_t hi, int tid)      /* [lo, hi): output BYTES */
{
    qpack_job *j = (qpack_job *)ctx;
    (void)tid;
    const uint8_t *q = j->q, *pair = j->pair;
    const int per = 8 / j->bits;                                          /* bases per output byte */
    int64_t full = j->n / per;
    if (full > hi) full = hi;
    if (j->bits == 2)
        for (int64_t o = lo; o < full; o++) {
            const uint8_t *x = q + 4 * o;
            j->out[o] = (uint8_t)(pair[x[0] | ((unsigned)x[1] << 8)] | (pair[x[2] | ((unsigned)x[3] << 8)] << 4));
        }
    else
        for (int64_t o = lo; o < full; o++) {
            const uint8_t *x = q + 2 * o;
            j->out[o] = pair[x[0] | ((unsigned)x[1] << 8)];
        }
    for (int64_t o = full > lo ? full : lo; o < hi; o++) {                /* the last, partial byte */
        unsigned b = 0;
        for (int k = 0; k < per && per * o + k < j->n; k++) b |= (unsigned)j->map[q[per * o + k]] << (j->bits * k);
        j->out[o] = (uint8_t)b;
    }
}
int crumble_pack_quals(int64_t n_bases, const uint8_t *quals, uint8_t *out, uint8_t lut[16])
{
    memset(lut, 0, 16);
    if (n_bases <= 0) { if (out) memset(out, 0, 16); return 2; }
    const int nt = host_threads_();
    unsigned char (*seen)[256] = calloc((size_t)(nt > 0 ? nt : 1) + 1, 256);
    if (!seen) return 0;
    qseen_job sj = { quals, n_bases, seen };
    team_run_once(nt, n_bases, 1 << 16, 1 << 20, qseen_piece, &sj);
    int nv = 0;
    qpack_job pj;
    memset(pj.map, 0, sizeof pj.map);
    for (int v = 0; v < 256; v++) {
        int any = 0;
        for (int t = 0; t <= (nt > 0 ? nt : 1); t++) any |= seen[t][v];
        if (any) { if (nv < 16) { lut[nv] = (uint8_t)v; pj.map[v] = (uint8_t)nv; } nv++; }
    }
    free(seen);
    if (nv > 16) return 0;
    if (!out) return nv <= 4 ? 2 : 4;                                  /* the probe: would this batch take codes, and how wide */
    pj.q = quals; pj.out = out; pj.n = n_bases; pj.bits = nv <= 4 ? 2 : 4;
    pj.pair = malloc(65536);
    if (!pj.pair) return 0;
    for (int b1 = 0; b1 < nv; b1++)                                    /* (only pairs of values that occur are ever looked up) */
        for (int b0 = 0; b0 < nv; b0++) pj.pair[lut[b0] | ((unsigned)lut[b1] << 8)] = (uint8_t)(b0 | (b1 << pj.bits));
    const int64_t nbytes = (n_bases * pj.bits + 7) / 8;
    team_run_once(nt, nbytes, 1 << 16, 1 << 20, qpack_piece, &pj);
    free(pj.pair);
    memset(out + nbytes, 0, 16);
    return pj.bits;
}

/* ---- coded download, host side (include/crumble_hip.h): 2 bits per base -> quality bytes -------------------- */
void crumble_decode_seek(const crumble_coded *c, int64_t base, crumble_decode_cursor *cur)
{
    cur->base = base;
    cur->esc_pos = 0;
    if (base >= c->n_bases) return;
    const int64_t blk = base / CRUMBLE_CODED_BLOCK, b0 = blk * CRUMBLE_CODED_BLOCK;
    int64_t e = c->esc_off[blk];
    const uint32_t *w = c->codes + (b0 >> 4);
    const int64_t nfull = (base - b0) >> 4;
    for (int64_t k = 0; k < nfull; k++) e += __builtin_popcount(w[k] & (w[k] >> 1) & 0x55555555u);
    const int rem = (int)((base - b0) & 15);
    if (rem) { const uint32_t x = w[nfull] & ((1u << (2 * rem)) - 1u); e += __builtin_popcount(x & (x >> 1) & 0x55555555u); }
    cur->esc_pos = e;
}

void crumble_decode_run(const crumble_coded *c, crumble_decode_cursor *cur, int64_t n, const uint8_t *qin, uint8_t *qout)
{
    int64_t b = cur->base, e = cur->esc_pos;
    const int64_t end = b + n;
    const uint8_t va = c->value[0], vb = c->value[1];
    /* a byte of codes (four bases, none of them code 2 or 3) -> four quality bytes */
    const uint32_t va4 = va * 0x01010101u, d4 = (uint32_t)(va ^ vb);
    int64_t k = 0;                                          /* index into qin / qout */
    while (b < end) {
        if ((b & (CRUMBLE_CODED_BLOCK - 1)) == 0) e = c->esc_off[b / CRUMBLE_CODED_BLOCK];   /* a block's escapes start where it says */
        const int sh = (int)(b & 15);
        uint32_t w = c->codes[b >> 4] >> (2 * sh);
        int m = 16 - sh;
        if ((int64_t)m > end - b) m = (int)(end - b);
        if (!qout) {                                        /* skipped bases: only the escape cursor moves */
            const uint32_t x = m < 16 ? (w & ((1u << (2 * m)) - 1u)) : w;
            e += __builtin_popcount(x & (x >> 1) & 0x55555555u);
        } else if (m == 16 && !(w & 0xaaaaaaaau)) {         /* sixteen bases of the two values: the bulk of every batch */
            uint32_t o[4] = { va4, va4, va4, va4 };
            if (w)
                for (int i = 0; i < 4; i++, w >>= 8) {
                    /* bits 0, 2, 4, 6 of the code byte -> 0x00 / 0x01 in bytes 0..3, times (va ^ vb), xor va */
                    const uint32_t s = (w & 1u) | ((w & 4u) << 6) | ((w & 16u) << 12) | ((w & 64u) << 18);
                    o[i] = va4 ^ (s * d4);
                }
            memcpy(qout + k, o, 16);
        } else
            for (int i = 0; i < m; i++, w >>= 2) {
                const unsigned code = w & 3u;
                qout[k + i] = code == 0 ? va : code == 1 ? vb : code == 2 ? c->cap[qin[k + i]] : c->esc[e++];
            }
        b += m; k += m;
    }
    cur->base = b; cur->esc_pos = e;
}

typedef struct { const crumble_coded *c; const uint8_t *qin; uint8_t *qout; } decode_job;
static void decode_blocks(void *ctx, int64_t lo, int64_t hi, int tid)
{
    const decode_job *j = ctx;
    (void)tid;
    crumble_decode_cursor cur;
    const int64_t b0 = lo * CRUMBLE_CODED_BLOCK;
    int64_t b1 = hi * CRUMBLE_CODED_BLOCK;
    if (b1 > j->c->n_bases) b1 = j->c->n_bases;
    crumble_decode_seek(j->c, b0, &cur);
    crumble_decode_run(j->c, &cur, b1 - b0, j->qin + b0, j->qout + b0);
}
/* threads for the host helpers: the CPUs this process may use -- on line, in its affinity mask, within its cgroup's cpu.max
 * quota (a container that sees 256 CPUs and may use 16 runs 32 threads slower than 16: the streamed leg's uploader thread was
 * the one that paid, 160 -> 186 ms of copies) -- the rule of bamio.c's default_threads() */
static int g_host_threads;      /* crumble_set_host_threads(): the caller's budget (-@), 0 = all the process may use */
void crumble_set_host_threads(int n) { __atomic_store_n(&g_host_threads, n > 0 ? n : 0, __ATOMIC_RELAXED); }
static int host_threads_(void)
{
    const int budget = __atomic_load_n(&g_host_threads, __ATOMIC_RELAXED);
    if (budget > 0) return budget > 32 ? 32 : budget;
    long nt = sysconf(_SC_NPROCESSORS_ONLN);
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0 && CPU_COUNT(&set) < nt) nt = CPU_COUNT(&set);
    FILE *fp = fopen("/sys/fs/cgroup/cpu.max", "r");
    if (fp) {
        long long period = 0;
        char q[32];
        if (fscanf(fp, "%31s %lld", q, &period) == 2 && strcmp(q, "max") && period > 0) {
            const long long c = (atoll(q) + period - 1) / period;
            if (c > 0 && c < nt) nt = (long)c;
        }
        fclose(fp);
    }
    return (int)(nt > 32 ? 32 : nt < 1 ? 1 : nt);
}
void crumble_decode_quals(const crumble_coded *c, const uint8_t *quals_in, uint8_t *qual_out, int threads)
{
    decode_job j = { c, quals_in, qual_out };
    team_run_once(threads > 0 ? threads : host_threads_(), c->n_blocks, 64, 16, decode_blocks, &j);
}
