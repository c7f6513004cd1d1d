/*
 * bamio.c -- BGZF + BAM ingest / egress for the column engine (include/crumble_bamio.h).
 *
 * Written from the SAMv1 specification (section 4: BGZF 4.1, BAM 4.2); replaces what reference
 * crumble gets from htslib's sam_read1 / sam_write1 (snp_score.c:1116-1117, 1101).
 *
 * Layout decisions (host side of an HBM-resident engine):
 *  - a contig's records are inflated straight into ONE growing byte buffer (`raw`), never copied
 *    again: the SoA views handed to the engine are built from it by a team of host threads (team.h), the engine's
 *    quality bytes are scattered back into it, and the same buffer is deflated for output;
 *  - BGZF blocks are independent deflate streams, so both directions are a parallel-for over blocks
 *    with a prefix sum over the (known) block sizes; file reads/writes stay sequential.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <errno.h>
#include <stdarg.h>
#include <zlib.h>
#include <unistd.h>
#include <sys/mman.h>
#include <pthread.h>
#include "../../include/crumble_bamio.h"
#include "team.h"

static char g_err[512];          /* process-wide: the record scan may run on any thread of the team */
const char *crumble_bam_last_error(void) { return g_err; }
static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

/* coarse phase timers (seconds), reported by crumble_bam_timing(): where host time goes */
#include <time.h>
static double T_fread, T_inflate, T_scan, T_prefix, T_unpack, T_scatter, T_deflate, T_fwrite;
static inline double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
int crumble_bam_timing(char *buf, size_t len)
{
    return snprintf(buf, len, "bamio: fread %.3f inflate %.3f record-scan %.3f prefix %.3f unpack %.3f | scatter %.3f deflate %.3f fwrite %.3f s",
                    T_fread, T_inflate, T_scan, T_prefix, T_unpack, T_scatter, T_deflate, T_fwrite);
}

/* ---- libdeflate, when the host has it (resolved at run time like htslib below; zlib otherwise) ----------------
 * BGZF blocks are whole raw-deflate streams of known size: exactly libdeflate's one-shot interface, two to three times
 * zlib's speed in both directions, and its crc32 uses the carry-less multiply.  Only entry points whose signatures
 * have not changed since libdeflate 1.0 are used; the prototypes are declared here, as the htslib backend declares its
 * own.  CRUMBLE_BAM_ZLIB=1 keeps zlib (the tests run both). */
#include <dlfcn.h>
static struct {
    void *lib;
    void *(*alloc_decompressor)(void);
    int (*deflate_decompress)(void *, const void *, size_t, void *, size_t, size_t *);
    void (*free_decompressor)(void *);
    void *(*alloc_compressor)(int);
    size_t (*deflate_compress)(void *, const void *, size_t, void *, size_t);
    void (*free_compressor)(void *);
    uint32_t (*crc32)(uint32_t, const void *, size_t);
    int ok;
} LD;
static pthread_once_t ld_once = PTHREAD_ONCE_INIT;
static void ld_load(void)
{
    const char *z = getenv("CRUMBLE_BAM_ZLIB");
    if (z && *z && *z != '0') return;
    LD.lib = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
    if (!LD.lib) LD.lib = dlopen("libdeflate.so", RTLD_NOW | RTLD_LOCAL);
    if (!LD.lib) return;
#define LDSYM(field, name) do { *(void **)&LD.field = dlsym(LD.lib, name); if (!LD.field) return; } while (0)
    LDSYM(alloc_decompressor, "libdeflate_alloc_decompressor"); LDSYM(deflate_decompress, "libdeflate_deflate_decompress");
    LDSYM(free_decompressor, "libdeflate_free_decompressor"); LDSYM(alloc_compressor, "libdeflate_alloc_compressor");
    LDSYM(deflate_compress, "libdeflate_deflate_compress"); LDSYM(free_compressor, "libdeflate_free_compressor");
    LDSYM(crc32, "libdeflate_crc32");
#undef LDSYM
    LD.ok = 1;
}
static int ld_ready(void) { pthread_once(&ld_once, ld_load); return LD.ok; }
/* "libdeflate" or "zlib": what inflates and deflates the BGZF blocks of this process */
const char *crumble_bam_codec(void) { return ld_ready() ? "libdeflate" : "zlib"; }

static inline uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
static inline uint16_t le16(const uint8_t *p) { return (uint16_t)(p[0] | p[1] << 8); }
static inline void put32(uint8_t *p, uint32_t v) { p[0] = v; p[1] = v >> 8; p[2] = v >> 16; p[3] = v >> 24; }
static inline void put16(uint8_t *p, uint16_t v) { p[0] = v; p[1] = v >> 8; }

/* Multi-GB buffers come straight from mmap with MADV_HUGEPAGE: first touch of 4 KiB pages by a
 * large team of threads is otherwise the dominant host cost (one page fault per 4 KiB, serialised on
 * the address-space lock).  Growing keeps the contents through mremap (no copy). */
/* Sizes are rounded up to size classes (2^k and 1.5 * 2^k, at least 2 MiB) and freed buffers of a class wait in a
 * small pool: a contig pipeline (crumble-hip: several groups in flight, a quality buffer per group) would otherwise
 * map, fault in and unmap gigabytes per contig.  The class is a pure function of the requested size, so callers
 * keep passing the size they asked for. */
#include <pthread.h>
static size_t big_class(size_t n)
{
    size_t c = (size_t)2 << 20;
    while (c < n) { if (c + c / 2 >= n) return c + c / 2; c *= 2; }
    return c;
}
#define BIG_POOL 12
static struct { void *p; size_t c; } big_pool[BIG_POOL];
static pthread_mutex_t big_mu = PTHREAD_MUTEX_INITIALIZER;
static void *big_alloc(size_t n)
{
    if (!n) n = 1;
    const size_t c = big_class(n);
    pthread_mutex_lock(&big_mu);
    for (int i = 0; i < BIG_POOL; i++)
        if (big_pool[i].p && big_pool[i].c == c) { void *p = big_pool[i].p; big_pool[i].p = NULL; pthread_mutex_unlock(&big_mu); return p; }
    pthread_mutex_unlock(&big_mu);
    void *p = mmap(NULL, c, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) return NULL;
    madvise(p, c, MADV_HUGEPAGE);
    return p;
}
static void big_free(void *p, size_t n)
{
    if (!p || !n) return;
    const size_t c = big_class(n);
    pthread_mutex_lock(&big_mu);
    int slot = -1;
    for (int i = 0; i < BIG_POOL; i++) if (!big_pool[i].p) { slot = i; break; }
    if (slot < 0) {                                         /* pool full: the smallest parked buffer goes */
        int sm = 0;
        for (int i = 1; i < BIG_POOL; i++) if (big_pool[i].c < big_pool[sm].c) sm = i;
        if (big_pool[sm].c < c) { munmap(big_pool[sm].p, big_pool[sm].c); slot = sm; }
    }
    if (slot >= 0) { big_pool[slot].p = p; big_pool[slot].c = c; p = NULL; }
    pthread_mutex_unlock(&big_mu);
    if (p) munmap(p, c);
}
static void *big_grow(void *p, size_t old, size_t n)
{
    if (!p || !old) return big_alloc(n);
    const size_t co = big_class(old), cn = big_class(n);
    if (cn == co) return p;
    void *q = mremap(p, co, cn, MREMAP_MAYMOVE);
    if (q == MAP_FAILED) return NULL;
    madvise(q, cn, MADV_HUGEPAGE);
    return q;
}
void *crumble_bam_big_alloc(size_t n) { return big_alloc(n); }
void  crumble_bam_big_free(void *p, size_t n) { big_free(p, n); }

/* Default team size: the CPUs this process may actually use (affinity mask and cgroup v2 quota),
 * at most 64 -- beyond that the per-block zlib work no longer scales and an over-subscribed
 * container (more visible CPUs than quota) collapses. */
#include <sched.h>
static int default_threads(void)
{
    int n = (int)sysconf(_SC_NPROCESSORS_ONLN);
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0 && c < n) n = c; }
    FILE *fp = fopen("/sys/fs/cgroup/cpu.max", "r");
    if (fp) {
        long long quota = 0, period = 0;
        char q[32];
        if (fscanf(fp, "%31s %lld", q, &period) == 2 && strcmp(q, "max") && period > 0) {
            quota = atoll(q);
            const int c = (int)((quota + period - 1) / period);
            if (c > 0 && c < n) n = c;
        }
        fclose(fp);
    }
    if (n > 64) n = 64;
    return n < 1 ? 1 : n;
}

int crumble_bam_default_threads(void) { return default_threads(); }

/* ------------------------------------------------------------------ BGZF in */
#define CHUNK_IN   (8u << 20)       /* compressed bytes read per refill: what is inflated beyond a contig's last record is
                                     * carried over to the next window (crumble_bam_detach_group copies it), so not too much at once */
#define RAW_MIN_CAP ((size_t)1 << 30) /* address space of a record buffer (see raw_reserve) */
#define MAX_BLOCKS 16384            /* per refill (>= CHUNK_IN / smallest sane block, grown on demand) */

/* the record buffer is shared between the reader and the groups detached from it (crumble_bam_detach_group): a group
 * owns the bytes of its records, the reader goes on behind them in the same mapping -- nothing is copied at a detach */
typedef struct { atomic_int refs; uint8_t *p; size_t cap; } rawbuf;
static rawbuf *rawbuf_new(size_t cap)
{
    rawbuf *b = malloc(sizeof *b);
    if (!b) return NULL;
    b->p = big_alloc(cap);
    if (!b->p) { free(b); return NULL; }
    b->cap = cap; atomic_init(&b->refs, 1);
    return b;
}
static void rawbuf_release(rawbuf *b)
{
    if (b && atomic_fetch_sub(&b->refs, 1) == 1) { big_free(b->p, b->cap); free(b); }
}

struct crumble_bam_reader {
    FILE *fp;
    int threads;
    rawbuf *rb;                               /* raw / raw_cap below are its p / cap */
    crumble_bam_coder codec; int has_codec;   /* crumble_bam_reader_set_codec: BGZF members inflated by the caller's codec (the device) */
    size_t chunk_in;                          /* compressed bytes read per refill */
    int64_t hdr_bytes;                        /* size of the BAM header in the inflated stream */
    team *tm;                                 /* the reader's own team (idle workers sleep: team.h); NULL in a detached group */
    uint8_t *cbuf; size_t clen, ccap;        /* compressed bytes not yet inflated  */
    int eof_file;
    /* the uncompressed stream window: [0, raw_len) valid, scan = parse position     */
    uint8_t *raw; size_t raw_len, raw_cap, scan;
    size_t group_end;                         /* end of the current group inside raw */
    size_t head;                              /* bytes in front of it that are done with (dropped lazily) */
    /* header */
    char *text; int64_t l_text;
    int32_t n_ref; char **ref_name; int64_t *ref_len;
    /* current group */
    int64_t n, cap_n;
    size_t *rec_off;                          /* [n] offset of block_size field      */
    int32_t tid;
    int32_t *pos; uint8_t *mapq; uint16_t *flag; int64_t *seq_off, *cig_off;
    int32_t *lq; uint16_t *ncg;               /* BAM's own l_seq / n_cigar_op per read (crumble_batch.l_qseq / n_cigar) */
    int long_cig;                             /* the group holds a CG-tag CIGAR (> 65535 ops): no n_cigar form */
    uint32_t *cigar; uint8_t *bases, *quals;
    int packed4;                    /* bases[] holds BAM's packing in the crumble_batch.bases4 layout (crumble_bam_set_packed_bases) */
    size_t cap_reads, cap_bases, cap_cig;
    /* record scan (runs beside the inflate team) */
    int have_tid, group_done, scan_err;
    int64_t acc_nb, acc_nc;
};

typedef struct { size_t coff, clen, uoff, ulen; uint32_t crc; } blk_t;

static int raw_reserve(crumble_bam_reader *r, size_t need)
{
    if (need <= r->raw_cap) return 0;
    /* (address space, not memory: pages that are never touched cost nothing, and a mapping that need not grow while detached
     * groups still read its front is never copied) */
    size_t nc = r->raw_cap ? r->raw_cap : RAW_MIN_CAP;
    while (nc < need) nc += nc / 2 + (16u << 20);
    nc = (nc + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    if (r->rb && atomic_load(&r->rb->refs) == 1) {          /* nobody else looks at it: grow in place (may move) */
        uint8_t *p = big_grow(r->raw, r->raw_cap, nc);
        if (!p) return fail(CRUMBLE_BAM_ENOMEM, "out of memory growing record buffer to %zu bytes", nc);
        r->rb->p = p; r->rb->cap = nc;
        r->raw = p; r->raw_cap = nc;
        return 0;
    }
    /* detached groups still read the front of it: a new mapping, the live bytes at the SAME offsets (the pages in front of
     * them are never touched), the old one left to its groups */
    rawbuf *nb = rawbuf_new(nc);
    if (!nb) return fail(CRUMBLE_BAM_ENOMEM, "out of memory growing record buffer to %zu bytes", nc);
    if (r->rb) { memcpy(nb->p + r->head, r->raw + r->head, r->raw_len - r->head); rawbuf_release(r->rb); }
    r->rb = nb; r->raw = nb->p; r->raw_cap = nb->cap;
    return 0;
}

/* Appends more uncompressed bytes to raw.  Returns 1 if bytes were added, 0 at EOF, < 0 on error. */
static void scan_records(crumble_bam_reader *r);

/* one piece of a refill: BGZF blocks [lo, hi) inflated and CRC-checked into the record buffer.  Every thread of the
 * team keeps one inflate state (team slot), reset per block. */
typedef struct { z_stream zs; int zok; void *ldd; } inflate_tls;
static void inflate_tls_free(void *p_)
{
    inflate_tls *p = p_;
    if (p->ldd) LD.free_decompressor(p->ldd);
    if (p->zok) inflateEnd(&p->zs);
    free(p);
}
typedef struct {
    crumble_bam_reader *r; const blk_t *blk; uint8_t *dst; int use_ld, side;
    atomic_int side_taken_done;     /* `side`: the first thread to arrive walks the records that are already there */
    atomic_int bad;
} inflate_job;
static void inflate_blocks(void *ctx, int64_t lo, int64_t hi, int tid)
{
    inflate_job *j = ctx;
    crumble_bam_reader *r = j->r;
    const blk_t *blk = j->blk;
    uint8_t *dst = j->dst;
    int bad = 0;
    if (j->side && !atomic_exchange(&j->side_taken_done, 1)) {
        /* the sequential walk over the records inflated so far costs about as much as one round of inflate: one
         * thread does it while the others start on the blocks */
        scan_records(r);
    }
    team *t = r->tm;                                    /* (every reader has one) */
    inflate_tls *tl = t->slot[tid];
    if (!tl) {
        tl = calloc(1, sizeof *tl);
        if (!tl) { atomic_fetch_or(&j->bad, 1); return; }
        tl->ldd = j->use_ld ? LD.alloc_decompressor() : NULL;
        tl->zok = tl->ldd ? 0 : inflateInit2(&tl->zs, -15) == Z_OK;
        t->slot[tid] = tl;
    }
    for (int64_t i = lo; i < hi; i++) {
        if (blk[i].ulen == 0) continue;
        if (tl->ldd) {
            size_t got = 0;
            if (LD.deflate_decompress(tl->ldd, r->cbuf + blk[i].coff, blk[i].clen, dst + blk[i].uoff, blk[i].ulen, &got) != 0 ||
                got != blk[i].ulen) bad |= 2;
            if (LD.crc32(0, dst + blk[i].uoff, blk[i].ulen) != blk[i].crc) bad |= 4;
            continue;
        }
        if (!tl->zok || inflateReset(&tl->zs) != Z_OK) { bad |= 1; continue; }
        tl->zs.next_in = r->cbuf + blk[i].coff; tl->zs.avail_in = (uInt)blk[i].clen;
        tl->zs.next_out = dst + blk[i].uoff;    tl->zs.avail_out = (uInt)blk[i].ulen;
        const int zr = inflate(&tl->zs, Z_FINISH);
        if (zr != Z_STREAM_END || tl->zs.avail_out != 0) bad |= 2;
        if (crc32(crc32(0L, Z_NULL, 0), dst + blk[i].uoff, (uInt)blk[i].ulen) != blk[i].crc) bad |= 4;
    }
    if (bad) atomic_fetch_or(&j->bad, bad);
}

/* `side`: run scan_records() on the bytes that are already there while the team inflates the new chunk */
static int refill(crumble_bam_reader *r, int side)
{
    for (;;) {
        if (!r->eof_file) {
            const size_t chunk = r->chunk_in ? r->chunk_in : CHUNK_IN;
            if (r->ccap < r->clen + chunk) {
                uint8_t *p = realloc(r->cbuf, r->clen + chunk);
                if (!p) return fail(CRUMBLE_BAM_ENOMEM, "out of memory");
                r->cbuf = p; r->ccap = r->clen + chunk;
            }
            const double tf = now_s();
            size_t got = fread(r->cbuf + r->clen, 1, chunk, r->fp);
            T_fread += now_s() - tf;
            if (got < chunk) {
                if (ferror(r->fp)) return fail(CRUMBLE_BAM_EIO, "read error: %s", strerror(errno));
                r->eof_file = 1;
            }
            r->clen += got;
        }
        if (r->clen == 0) { if (side) scan_records(r); return 0; }
        /* delimit whole blocks */
        blk_t *blk = NULL; size_t nb = 0, capb = 0, off = 0, utot = 0;
        while (off + 18 <= r->clen) {
            const uint8_t *h = r->cbuf + off;
            if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) { free(blk); return fail(CRUMBLE_BAM_EFORMAT, "not a BGZF block (bad gzip header)"); }
            const unsigned xlen = le16(h + 10);
            if (off + 12 + xlen > r->clen) break;
            int bsize = -1;
            for (unsigned x = 0; x + 4 <= xlen; ) {
                const uint8_t *sf = h + 12 + x;
                const unsigned slen = le16(sf + 2);
                if (sf[0] == 'B' && sf[1] == 'C' && slen == 2) bsize = le16(sf + 4);
                x += 4 + slen;
            }
            if (bsize < 0) { free(blk); return fail(CRUMBLE_BAM_EFORMAT, "gzip member without BGZF BC field"); }
            const size_t total = (size_t)bsize + 1;
            if (total < 12 + xlen + 8) { free(blk); return fail(CRUMBLE_BAM_EFORMAT, "corrupt BGZF block size"); }
            if (off + total > r->clen) break;
            if (nb == capb) { capb = capb ? capb * 2 : 1024; blk = realloc(blk, capb * sizeof *blk); if (!blk) return fail(CRUMBLE_BAM_ENOMEM, "out of memory"); }
            blk[nb].coff = off + 12 + xlen;
            blk[nb].clen = total - 12 - xlen - 8;
            blk[nb].ulen = le32(h + total - 4);
            blk[nb].crc  = le32(h + total - 8);
            blk[nb].uoff = utot;
            if (blk[nb].ulen > 65536) { free(blk); return fail(CRUMBLE_BAM_EFORMAT, "BGZF ISIZE > 64 KiB"); }
            utot += blk[nb].ulen;
            nb++;
            off += total;
        }
        if (nb == 0) {
            free(blk);
            if (r->eof_file) return fail(CRUMBLE_BAM_EFORMAT, "truncated BGZF block at end of file");
            continue;                                       /* need more compressed bytes */
        }
        int rc = raw_reserve(r, r->raw_len + utot);
        if (rc) { free(blk); return rc; }
        uint8_t *dst = r->raw + r->raw_len;
        int bad = 0;
        const double ti = now_s();
        if (r->has_codec) {
            /* the caller's codec (crumble-hip: the device, crumble_gz_inflate) takes the whole round */
            crumble_bam_block *gb = malloc(nb * sizeof *gb);
            if (!gb) { free(blk); return fail(CRUMBLE_BAM_ENOMEM, "out of memory"); }
            for (size_t i = 0; i < nb; i++) {
                gb[i].coff = (int64_t)blk[i].coff; gb[i].uoff = (int64_t)blk[i].uoff;
                gb[i].clen = (uint32_t)blk[i].clen; gb[i].ulen = (uint32_t)blk[i].ulen; gb[i].crc = blk[i].crc; gb[i].pad_ = 0;
            }
            const int crc = r->codec.inflate(r->codec.ctx, r->cbuf, (int64_t)off, gb, (int64_t)nb, dst, (int64_t)utot);
            free(gb); free(blk);
            T_inflate += now_s() - ti;
            if (crc) return fail(CRUMBLE_BAM_EFORMAT, "BGZF inflate/CRC failure in the codec (code %d)", crc);
            if (side) scan_records(r);
        } else {
        const int use_ld = ld_ready();
        inflate_job job = { r, blk, dst, use_ld, side, 0 };
        atomic_init(&job.bad, 0);
        team_run(r->tm, (int64_t)nb, 4, inflate_blocks, &job);
        if (side && !atomic_load(&job.side_taken_done)) scan_records(r);     /* (no block at all: nobody took the walk) */
        bad = atomic_load(&job.bad);
        free(blk);
        T_inflate += now_s() - ti;
        if (bad) return fail(CRUMBLE_BAM_EFORMAT, "BGZF inflate/CRC failure (mask %d)", bad);
        }
        memmove(r->cbuf, r->cbuf + off, r->clen - off);
        r->clen -= off;
        r->raw_len += utot;
        side = 0;                                           /* at most once per call */
        if (utot) return 1;
        if (r->eof_file && r->clen == 0) return 0;          /* only empty (EOF marker) blocks */
    }
}

/* makes sure [scan, scan+need) is in raw; returns 1 ok, 0 clean EOF before any byte, < 0 error */
static int need_bytes(crumble_bam_reader *r, size_t need)
{
    while (r->raw_len - r->scan < need) {
        const int rc = refill(r, 0);
        if (rc < 0) return rc;
        if (rc == 0) {
            if (r->raw_len == r->scan) return 0;
            return fail(CRUMBLE_BAM_EFORMAT, "truncated BAM record at end of file");
        }
    }
    return 1;
}

crumble_bam_reader *crumble_bam_open(const char *path, int threads, int *err)
{
    int e = 0;
    crumble_bam_reader *r = calloc(1, sizeof *r);
    if (!r) { e = fail(CRUMBLE_BAM_ENOMEM, "out of memory"); goto bad; }
    r->threads = threads > 0 ? threads : default_threads();
    r->tm = team_create(r->threads);
    if (!r->tm) { e = fail(CRUMBLE_BAM_ENOMEM, "out of memory"); goto bad; }
    r->tm->slot_free = inflate_tls_free;
    r->fp = fopen(path, "rb");
    if (!r->fp) { e = fail(CRUMBLE_BAM_EIO, "%s: %s", path, strerror(errno)); goto bad; }
    int rc;
    if ((rc = need_bytes(r, 12)) <= 0) { e = rc ? rc : fail(CRUMBLE_BAM_EFORMAT, "%s: empty file", path); goto bad; }
    if (memcmp(r->raw, "BAM\1", 4)) { e = fail(CRUMBLE_BAM_EFORMAT, "%s: no BAM magic", path); goto bad; }
    r->l_text = (int32_t)le32(r->raw + 4);
    if (r->l_text < 0) { e = fail(CRUMBLE_BAM_EFORMAT, "negative l_text"); goto bad; }
    r->scan = 8;
    if ((rc = need_bytes(r, (size_t)r->l_text + 4)) <= 0) { e = rc ? rc : fail(CRUMBLE_BAM_EFORMAT, "truncated header"); goto bad; }
    r->text = malloc((size_t)r->l_text + 1);
    if (!r->text) { e = fail(CRUMBLE_BAM_ENOMEM, "out of memory"); goto bad; }
    memcpy(r->text, r->raw + r->scan, (size_t)r->l_text); r->text[r->l_text] = 0;
    r->l_text = (int64_t)strlen(r->text);                   /* header text may be NUL padded */
    r->scan += (size_t)le32(r->raw + 4);
    r->n_ref = (int32_t)le32(r->raw + r->scan); r->scan += 4;
    if (r->n_ref < 0) { e = fail(CRUMBLE_BAM_EFORMAT, "negative n_ref"); goto bad; }
    r->ref_name = calloc((size_t)r->n_ref + 1, sizeof *r->ref_name);
    r->ref_len = calloc((size_t)r->n_ref + 1, sizeof *r->ref_len);
    if (!r->ref_name || !r->ref_len) { e = fail(CRUMBLE_BAM_ENOMEM, "out of memory"); goto bad; }
    for (int32_t i = 0; i < r->n_ref; i++) {
        if ((rc = need_bytes(r, 4)) <= 0) { e = rc ? rc : fail(CRUMBLE_BAM_EFORMAT, "truncated reference list"); goto bad; }
        const int32_t ln = (int32_t)le32(r->raw + r->scan);
        if (ln <= 0) { e = fail(CRUMBLE_BAM_EFORMAT, "bad reference name length"); goto bad; }
        if ((rc = need_bytes(r, (size_t)ln + 8)) <= 0) { e = rc ? rc : fail(CRUMBLE_BAM_EFORMAT, "truncated reference list"); goto bad; }
        r->ref_name[i] = malloc((size_t)ln + 1);
        if (!r->ref_name[i]) { e = fail(CRUMBLE_BAM_ENOMEM, "out of memory"); goto bad; }
        memcpy(r->ref_name[i], r->raw + r->scan + 4, (size_t)ln); r->ref_name[i][ln] = 0;
        r->ref_len[i] = (int32_t)le32(r->raw + r->scan + 4 + ln);
        r->scan += (size_t)ln + 8;
    }
    r->group_end = r->scan;                                 /* header bytes are dropped on the first call */
    r->hdr_bytes = (int64_t)r->scan;
    if (err) *err = 0;
    return r;
bad:
    if (err) *err = e;
    crumble_bam_close(r);
    return NULL;
}

void crumble_bam_close(crumble_bam_reader *r)
{
    if (!r) return;
    if (r->fp) fclose(r->fp);
    team_destroy(r->tm);
    free(r->cbuf); rawbuf_release(r->rb); free(r->text);
    if (r->ref_name) for (int32_t i = 0; i < r->n_ref; i++) free(r->ref_name[i]);
    free(r->ref_name); free(r->ref_len); free(r->rec_off);
    free(r->pos); free(r->mapq); free(r->flag); free(r->seq_off); free(r->cig_off); free(r->lq); free(r->ncg);
    free(r->cigar); big_free(r->bases, r->cap_bases); big_free(r->quals, r->cap_bases);
    free(r);
}

int32_t crumble_bam_n_ref(const crumble_bam_reader *r) { return r->n_ref; }
const char *crumble_bam_ref_name(const crumble_bam_reader *r, int32_t tid) { return tid >= 0 && tid < r->n_ref ? r->ref_name[tid] : "*"; }
int64_t crumble_bam_ref_len(const crumble_bam_reader *r, int32_t tid) { return tid >= 0 && tid < r->n_ref ? r->ref_len[tid] : 0; }
const char *crumble_bam_header_text(const crumble_bam_reader *r, int64_t *len) { if (len) *len = r->l_text; return r->text; }
int64_t crumble_bam_header_bytes(const crumble_bam_reader *r) { return r->hdr_bytes; }

/* fixed part of a BAM alignment record, offsets from the block_size field */
enum { F_REFID = 4, F_POS = 8, F_LNAME = 12, F_MAPQ = 13, F_NCIG = 16, F_FLAG = 18, F_LSEQ = 20, F_FIXED = 36 };

static const uint8_t *aux_end(const uint8_t *s, const uint8_t *lim);

/* SAMv1 4.2.2: a CIGAR of more than 65535 ops is stored in the CG:B,I tag and the record carries the
 * placeholder "<l_seq>S<reflen>N".  Returns the tag's uint32 array (and its length) or NULL. */
static const uint8_t *long_cigar(const uint8_t *p, uint32_t *n_ops)
{
    const uint32_t bs = le32(p), lname = p[F_LNAME], ncig = le16(p + F_NCIG);
    const int32_t lseq = (int32_t)le32(p + F_LSEQ);
    if (ncig != 2 || lseq <= 0) return NULL;
    const uint8_t *cg = p + F_FIXED + lname;
    const uint32_t c0 = le32(cg), c1 = le32(cg + 4);
    if ((c0 & 15) != 4 || (c0 >> 4) != (uint32_t)lseq || (c1 & 15) != 3) return NULL;
    const uint8_t *lim = p + 4 + bs;
    for (const uint8_t *t = cg + 8 + ((size_t)lseq + 1) / 2 + (size_t)lseq; t + 3 <= lim; ) {
        const uint8_t *e = aux_end(t + 2, lim);
        if (!e) return NULL;
        if (t[0] == 'C' && t[1] == 'G' && t[2] == 'B' && t[3] == 'I') { *n_ops = le32(t + 4); return t + 8; }
        t = e;
    }
    return NULL;
}

/* Walks the complete records in [scan, raw_len): group membership, offsets of the record and of its
 * bases / CIGAR words in the SoA arrays.  Stops at the first record of another refID (group_done)
 * or at the first incomplete record.  Errors are left in scan_err / g_err. */
static void scan_records(crumble_bam_reader *r)
{
    const double t0 = now_s();
    while (!r->group_done && !r->scan_err) {
        if (r->raw_len - r->scan < 4) break;
        const uint32_t bs = le32(r->raw + r->scan);
        if (bs < 32 || bs > (1u << 30)) { r->scan_err = fail(CRUMBLE_BAM_EFORMAT, "implausible BAM record size %u", bs); break; }
        if (r->raw_len - r->scan < 4 + (size_t)bs) break;
        const uint8_t *p = r->raw + r->scan;
        const int32_t tid = (int32_t)le32(p + F_REFID);
        if (!r->have_tid) { r->tid = tid; r->have_tid = 1; }
        else if (tid != r->tid) { r->group_done = 1; break; }
        const uint32_t lname = p[F_LNAME], ncig = le16(p + F_NCIG);
        const int32_t lseq = (int32_t)le32(p + F_LSEQ);
        if (lseq < 0 || (size_t)F_FIXED - 4 + lname + 4u * ncig + ((size_t)lseq + 1) / 2 + (size_t)lseq > bs) {
            r->scan_err = fail(CRUMBLE_BAM_EFORMAT, "BAM record fields exceed block_size"); break;
        }
        if (r->n + 1 >= r->cap_n) {
            const int64_t nc = r->cap_n ? r->cap_n * 2 : (1 << 20);
            size_t *q = realloc(r->rec_off, (size_t)nc * sizeof *q);
            int64_t *so = realloc(r->seq_off, (size_t)nc * sizeof *so), *co = realloc(r->cig_off, (size_t)nc * sizeof *co);
            if (q) r->rec_off = q;
            if (so) r->seq_off = so;
            if (co) r->cig_off = co;
            if (!q || !so || !co) { r->scan_err = fail(CRUMBLE_BAM_ENOMEM, "out of memory"); break; }
            r->cap_n = nc;
        }
        r->rec_off[r->n] = r->scan;
        r->seq_off[r->n] = r->acc_nb; r->cig_off[r->n] = r->acc_nc;
        uint32_t nops = ncig;
        if (ncig == 2) (void)long_cigar(p, &nops);            /* real op count of a CG-tag CIGAR */
        if (nops > 65535) r->long_cig = 1;
        r->acc_nb += lseq; r->acc_nc += nops;
        r->n++;
        r->scan += 4 + (size_t)bs;
    }
    T_scan += now_s() - t0;
}

/* records [lo, hi) of the current group -> the SoA arrays */
typedef struct { crumble_bam_reader *r; atomic_int unsupported; } unpack_job;
static void unpack_records(void *ctx, int64_t lo, int64_t hi, int tid)
{
    unpack_job *uj = ctx;
    crumble_bam_reader *r = uj->r;
    int unsupported = 0;
    (void)tid;
    for (int64_t i = lo; i < hi; i++) {
        const uint8_t *p = r->raw + r->rec_off[i];
        const uint32_t lname = p[F_LNAME], ncig = le16(p + F_NCIG);
        const int32_t lseq = (int32_t)le32(p + F_LSEQ);
        r->pos[i] = (int32_t)le32(p + F_POS);
        r->mapq[i] = p[F_MAPQ];
        r->flag[i] = le16(p + F_FLAG);
        r->lq[i] = lseq;
        r->ncg[i] = (uint16_t)(r->cig_off[i + 1] - r->cig_off[i]);      /* (a CG-tag CIGAR: the group keeps the offset form) */
        const uint8_t *cg = p + F_FIXED + lname;
        uint32_t *co = r->cigar + r->cig_off[i];
        uint32_t nlong = 0;
        const uint8_t *lc = ncig == 2 ? long_cigar(p, &nlong) : NULL;
        if (lc) { if ((int64_t)nlong != r->cig_off[i + 1] - r->cig_off[i]) unsupported |= 1; else for (uint32_t k = 0; k < nlong; k++) co[k] = le32(lc + 4 * k); }
        else for (uint32_t k = 0; k < ncig; k++) co[k] = le32(cg + 4 * k);
        const uint8_t *sq = cg + 4 * (size_t)ncig;
        if (r->packed4) memcpy(r->bases + CRUMBLE_BASES4_OFF(r->seq_off[i], i), sq, ((size_t)lseq + 1) / 2);   /* as they are */
        else {
            uint8_t *bo = r->bases + r->seq_off[i];
            for (int32_t k = 0; k + 1 < lseq; k += 2) { const uint8_t v = sq[k >> 1]; bo[k] = v >> 4; bo[k + 1] = v & 15; }
            if (lseq & 1) bo[lseq - 1] = sq[lseq >> 1] >> 4;
        }
        const uint8_t *ql = sq + ((size_t)lseq + 1) / 2;
        uint8_t *qo = r->quals + r->seq_off[i];
        if (lseq > 0 && ql[0] == 0xff) memset(qo, 0, (size_t)lseq);     /* absent qualities */
        else memcpy(qo, ql, (size_t)lseq);
    }
    if (unsupported) atomic_fetch_or(&uj->unsupported, unsupported);
}

int crumble_bam_next_contig(crumble_bam_reader *r, crumble_batch *b)
{
    /* drop the previous group (or the header) from the window: only the offset moves; the bytes are
     * compacted once the dead prefix is both large and most of the window (a file of thousands of
     * small contigs used to memmove the whole read-ahead window per contig) */
    if (r->group_end) { r->head = r->group_end; r->group_end = 0; }
    if (r->head > ((size_t)64 << 20) && r->head * 2 > r->raw_len) {
        if (atomic_load(&r->rb->refs) == 1) memmove(r->raw, r->raw + r->head, r->raw_len - r->head);
        else {                                              /* groups in flight hold the front: the live rest into a mapping of its own */
            const size_t live = r->raw_len - r->head;
            size_t nc = RAW_MIN_CAP;
            while (nc < live + ((size_t)256 << 20)) nc += nc / 2;
            rawbuf *nb = rawbuf_new(nc);
            if (!nb) return fail(CRUMBLE_BAM_ENOMEM, "out of memory");
            memcpy(nb->p, r->raw + r->head, live);
            rawbuf_release(r->rb);
            r->rb = nb; r->raw = nb->p; r->raw_cap = nb->cap;
        }
        r->raw_len -= r->head; r->scan -= r->head; r->head = 0;
    }
    r->n = 0; r->have_tid = 0; r->group_done = 0; r->scan_err = 0; r->acc_nb = r->acc_nc = 0; r->long_cig = 0;
    /* the resident bytes may already hold the whole next group: walk them before reading ahead */
    scan_records(r);
    if (r->scan_err) return r->scan_err;
    while (!r->group_done) {
        /* inflate the next chunk while one thread walks the records that are already there */
        const int rc = refill(r, 1);
        if (rc < 0) return rc;
        if (r->scan_err) return r->scan_err;
        if (r->group_done) break;
        if (rc == 0) {                                      /* end of file: everything left has been walked */
            scan_records(r);                                /* (bytes of the last round, if any) */
            if (r->scan_err) return r->scan_err;
            if (!r->group_done && r->raw_len != r->scan) return fail(CRUMBLE_BAM_EFORMAT, "truncated BAM record at end of file");
            break;
        }
    }
    r->group_end = r->scan;
    if (r->n == 0) return 0;
    const double tp0 = now_s();

    const int64_t n = r->n;
    if (r->cap_reads < (size_t)n + 1) {
        const size_t c = (size_t)n + (size_t)n / 4 + 64;
        free(r->pos); free(r->mapq); free(r->flag); free(r->lq); free(r->ncg);
        r->pos = malloc(c * sizeof *r->pos); r->mapq = malloc(c); r->flag = malloc(c * sizeof *r->flag);
        r->lq = malloc(c * sizeof *r->lq); r->ncg = malloc(c * sizeof *r->ncg);
        r->cap_reads = (r->pos && r->mapq && r->flag && r->lq && r->ncg) ? c : 0;
        if (!r->cap_reads) return fail(CRUMBLE_BAM_ENOMEM, "out of memory");
    }
    const int64_t nb = r->acc_nb, nc = r->acc_nc;
    r->seq_off[n] = nb; r->cig_off[n] = nc;
    const size_t need4 = r->packed4 ? (size_t)CRUMBLE_BASES4_SIZE(nb, n) : 0;
    if (r->cap_bases < (size_t)nb + 16 || r->cap_bases < need4 + 16) {
        const size_t want = (size_t)nb > need4 ? (size_t)nb : need4;
        const size_t c = (want + want / 8 + ((size_t)2 << 20)) & ~(((size_t)2 << 20) - 1);
        big_free(r->bases, r->cap_bases); big_free(r->quals, r->cap_bases);
        r->bases = big_alloc(c); r->quals = big_alloc(c);
        if (!r->bases || !r->quals) { big_free(r->bases, c); big_free(r->quals, c); r->bases = r->quals = NULL; }
        r->cap_bases = (r->bases && r->quals) ? c : 0;
        if (!r->cap_bases) return fail(CRUMBLE_BAM_ENOMEM, "out of memory");
    }
    if (r->cap_cig < (size_t)nc + 1) {
        const size_t c = (size_t)nc + (size_t)nc / 8 + 64;
        free(r->cigar);
        r->cigar = malloc(c * sizeof *r->cigar);
        r->cap_cig = r->cigar ? c : 0;
        if (!r->cap_cig) return fail(CRUMBLE_BAM_ENOMEM, "out of memory");
    }
    int unsupported = 0;
    const double tu0 = now_s();
    T_prefix += tu0 - tp0;
    unpack_job uj = { r, 0 };
    atomic_init(&uj.unsupported, 0);
    team_run(r->tm, n, 0, unpack_records, &uj);
    unsupported = atomic_load(&uj.unsupported);
    T_unpack += now_s() - tu0;
    if (unsupported) return fail(CRUMBLE_BAM_EFORMAT, "inconsistent CG tag");
    memset(b, 0, sizeof *b);
    b->tid = r->tid; b->n_reads = n;
    b->pos = r->pos; b->mapq = r->mapq; b->flag = r->flag;
    b->seq_off = r->seq_off; b->cig_off = r->cig_off; b->cigar = r->cigar;
    b->quals = r->quals;
    if (r->packed4) b->bases4 = r->bases; else b->bases = r->bases;
    if (!r->long_cig) { b->l_qseq = r->lq; b->n_cigar = r->ncg; b->n_bases = nb; b->n_cigar_ops = nc; }   /* 6 bytes per read to the device instead of 16 */
    return 1;
}

/* The reader hands the bases on as BAM packs them (crumble_batch.bases4, bases == NULL): no nibble expansion on the
 * host, half the bytes to the device.  Takes effect at the next crumble_bam_next_contig(). */
void crumble_bam_reader_set_codec(crumble_bam_reader *r, const crumble_bam_coder *c)
{
    if (!r) return;
    r->has_codec = c && c->inflate;
    if (r->has_codec) { r->codec = *c; r->chunk_in = (size_t)256 << 20; }    /* a device wants tens of thousands of members per call */
    else r->chunk_in = 0;
}
void crumble_bam_set_packed_bases(crumble_bam_reader *r, int on) { if (r) r->packed4 = on ? 1 : 0; }

/* A detached group: the reader-shaped shell that owns one contig's record bytes and SoA arrays, so that the
 * reader can go on to the next contig while this one is on a GPU or being written (crumble-hip --gpus N). */
crumble_bam_group *crumble_bam_detach_group(crumble_bam_reader *r)
{
    crumble_bam_reader *g = calloc(1, sizeof *g);
    if (!g) { fail(CRUMBLE_BAM_ENOMEM, "out of memory"); return NULL; }
    /* the group shares the reader's record buffer (its records lie in front of group_end, the reader goes on behind them) */
    atomic_fetch_add(&r->rb->refs, 1);
    g->rb = r->rb;
    g->threads = r->threads;
    g->raw = r->raw; g->raw_cap = r->raw_cap; g->raw_len = r->group_end; g->group_end = r->group_end; g->head = r->head;
    g->n = r->n; g->cap_n = r->cap_n; g->rec_off = r->rec_off; g->tid = r->tid;
    g->pos = r->pos; g->mapq = r->mapq; g->flag = r->flag; g->seq_off = r->seq_off; g->cig_off = r->cig_off;
    g->lq = r->lq; g->ncg = r->ncg;
    g->cigar = r->cigar; g->bases = r->bases; g->quals = r->quals;
    g->cap_reads = r->cap_reads; g->cap_bases = r->cap_bases; g->cap_cig = r->cap_cig;
    r->head = r->group_end; r->group_end = 0;
    r->n = 0; r->cap_n = 0; r->rec_off = NULL; r->seq_off = NULL; r->cig_off = NULL;
    r->pos = NULL; r->mapq = NULL; r->flag = NULL; r->cigar = NULL; r->bases = NULL; r->quals = NULL; r->lq = NULL; r->ncg = NULL;
    r->cap_reads = r->cap_bases = r->cap_cig = 0;
    return (crumble_bam_group *)g;
}
int crumble_bam_write_detached(crumble_bam_writer *w, crumble_bam_group *g, const uint8_t *qual_out)
{
    return crumble_bam_write_group(w, (crumble_bam_reader *)g, qual_out);
}
void crumble_bam_group_free(crumble_bam_group *gg)
{
    crumble_bam_reader *g = (crumble_bam_reader *)gg;
    if (!g) return;
    rawbuf_release(g->rb); free(g->rec_off); free(g->seq_off); free(g->cig_off);
    free(g->pos); free(g->mapq); free(g->flag); free(g->cigar); free(g->lq); free(g->ncg);
    big_free(g->bases, g->cap_bases); big_free(g->quals, g->cap_bases);
    free(g);
}

/* ----------------------------------------------------------------- BGZF out */
#define BLK_IN 0xff00u                       /* uncompressed bytes per block (htslib's choice too) */
struct crumble_bam_writer {
    FILE *fp; int level, threads; uint8_t *obuf; size_t ocap;
    crumble_bam_coder codec; int has_codec;   /* crumble_bam_writer_set_codec: blocks deflated by the caller's codec (the device) */
    team *tm;                          /* the writer's own team (team.h); NULL inside a crumble_hts (purge settings only) */
    /* purge_tags (snp_score.c:989-1054) */
    int aux_mode;                      /* 0 none, 1 whitelist (-t), 2 blacklist (-T) */
    uint8_t *aux_set;                  /* 65536-bit set of two-character tags */
    int bd[3], bi[3];                  /* BD / BI low, mid, high (already +33); all 0 = off */
    uint8_t *raw2; size_t raw2_cap;    /* compacted records when tags are dropped */
    size_t *noff; size_t noff_cap;
};

void crumble_bam_writer_set_codec(crumble_bam_writer *w, const crumble_bam_coder *c)
{
    if (!w) return;
    w->has_codec = c && c->deflate && c->deflate_bound;
    if (w->has_codec) w->codec = *c;
}

static const uint8_t BGZF_EOF[28] = { 0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0 };

/* one piece of a round of output: blocks [lo, hi) of the round deflated into their slots.  Every thread of the team
 * keeps one compressor (team slot). */
typedef struct { z_stream zs; int zok; void *ldc; } deflate_tls;
static void deflate_tls_free(void *p_)
{
    deflate_tls *p = p_;
    if (p->ldc) LD.free_compressor(p->ldc);
    if (p->zok) deflateEnd(&p->zs);
    free(p);
}
typedef struct { crumble_bam_writer *w; const uint8_t *data; size_t len, b0; size_t *osz; int use_ld; atomic_int bad; } deflate_job;
static void deflate_blocks(void *ctx, int64_t lo, int64_t hi, int tid)
{
    deflate_job *j = ctx;
    crumble_bam_writer *w = j->w;
    const uint8_t *data = j->data;
    const size_t len = j->len, SLOT = 65536;
    deflate_tls *tl = w->tm->slot[tid];
    if (!tl) {
        tl = calloc(1, sizeof *tl);
        if (!tl) { atomic_store(&j->bad, 1); for (int64_t i = lo; i < hi; i++) j->osz[i] = 0; return; }
        /* (zlib level 0 = stored blocks: libdeflate's level 0 is the same; the levels above it mean about the same effort) */
        tl->ldc = j->use_ld ? LD.alloc_compressor(w->level < 0 ? 6 : w->level > 12 ? 12 : w->level) : NULL;
        w->tm->slot[tid] = tl;
    }
    for (int64_t i = lo; i < hi; i++) {
        const size_t off = (j->b0 + (size_t)i) * BLK_IN, ul = len - off < BLK_IN ? len - off : BLK_IN;
        uint8_t *o = w->obuf + (size_t)i * SLOT;
        j->osz[i] = 0;
        size_t cl = tl->ldc ? LD.deflate_compress(tl->ldc, data + off, ul, o + 18, SLOT - 18 - 8) : 0;
        if (!cl) {                                      /* zlib (no libdeflate, or the one block it could not fit) */
            if (!tl->zok) { if (deflateInit2(&tl->zs, w->level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { atomic_store(&j->bad, 1); continue; } tl->zok = 1; }
            if (deflateReset(&tl->zs) != Z_OK) { atomic_store(&j->bad, 1); continue; }
            tl->zs.next_in = (Bytef *)(data + off); tl->zs.avail_in = (uInt)ul;
            tl->zs.next_out = o + 18; tl->zs.avail_out = (uInt)(SLOT - 18 - 8);
            if (deflate(&tl->zs, Z_FINISH) != Z_STREAM_END) { atomic_store(&j->bad, 1); continue; }
            cl = tl->zs.total_out;
        }
        static const uint8_t hdr[16] = { 0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0 };
        memcpy(o, hdr, 16);
        put16(o + 16, (uint16_t)(18 + cl + 8 - 1));
        put32(o + 18 + cl, tl->ldc ? LD.crc32(0, data + off, ul) : (uint32_t)crc32(crc32(0L, Z_NULL, 0), data + off, (uInt)ul));
        put32(o + 18 + cl + 4, (uint32_t)ul);
        j->osz[i] = 18 + cl + 8;
    }
}

/* deflates data[0..len) as consecutive BGZF blocks and writes them */
static int bgzf_write(crumble_bam_writer *w, const uint8_t *data, size_t len)
{
    const size_t SLOT = 65536, BATCH = 4096;                 /* 256 MiB of input per round */
    if (w->has_codec && len > 0) {
        /* the caller's codec (crumble-hip: the device, crumble_gz_deflate): rounds of whole blocks, one write each */
        const size_t round = (size_t)BATCH * BLK_IN;
        for (size_t o = 0; o < len; o += round) {
            const size_t n = len - o < round ? len - o : round;
            const size_t cap = (size_t)w->codec.deflate_bound((int64_t)n);
            if (w->ocap < cap) {
                big_free(w->obuf, w->ocap);
                w->obuf = big_alloc(cap);
                w->ocap = w->obuf ? cap : 0;
                if (!w->obuf) return fail(CRUMBLE_BAM_ENOMEM, "out of memory");
            }
            const double td = now_s();
            int64_t got = 0;
            const int rc = w->codec.deflate(w->codec.ctx, data + o, (int64_t)n, w->level, w->obuf, (int64_t)w->ocap, &got);
            T_deflate += now_s() - td;
            if (rc || got <= 0) return fail(CRUMBLE_BAM_EIO, "deflate failed in the codec (code %d)", rc);
            const double tw = now_s();
            if (fwrite(w->obuf, 1, (size_t)got, w->fp) != (size_t)got) return fail(CRUMBLE_BAM_EIO, "write error: %s", strerror(errno));
            T_fwrite += now_s() - tw;
        }
        return 0;
    }
    if (w->ocap < SLOT * BATCH) {
        big_free(w->obuf, w->ocap);
        uint8_t *p = big_alloc(SLOT * BATCH);
        if (!p) { w->obuf = NULL; w->ocap = 0; return fail(CRUMBLE_BAM_ENOMEM, "out of memory"); }
        w->obuf = p; w->ocap = SLOT * BATCH;
    }
    size_t *osz = malloc(BATCH * sizeof *osz);
    if (!osz) return fail(CRUMBLE_BAM_ENOMEM, "out of memory");
    const size_t nblk = (len + BLK_IN - 1) / BLK_IN;
    for (size_t b0 = 0; b0 < nblk; b0 += BATCH) {
        const size_t nb = nblk - b0 < BATCH ? nblk - b0 : BATCH;
        int bad = 0;
        const double td = now_s();
        const int use_ld = ld_ready();
        deflate_job job = { w, data, len, b0, osz, use_ld };
        atomic_init(&job.bad, 0);
        team_run(w->tm, (int64_t)nb, 4, deflate_blocks, &job);
        bad = atomic_load(&job.bad);
        T_deflate += now_s() - td;
        if (bad) { free(osz); return fail(CRUMBLE_BAM_EIO, "deflate failed"); }
        const double tw = now_s();
        /* compact the slots so the file sees one write per round */
        size_t tot = osz[0];
        for (size_t i = 1; i < nb; i++) { memmove(w->obuf + tot, w->obuf + i * SLOT, osz[i]); tot += osz[i]; }
        if (fwrite(w->obuf, 1, tot, w->fp) != tot) { free(osz); return fail(CRUMBLE_BAM_EIO, "write error: %s", strerror(errno)); }
        T_fwrite += now_s() - tw;
    }
    free(osz);
    return 0;
}

static int write_header(crumble_bam_writer *w, const char *text, int64_t l_text, const char *pg_line,
                        int32_t n_ref, const char *const *names, const int64_t *lens)
{
    size_t extra = pg_line ? strlen(pg_line) + 1 : 0, need = 12 + (size_t)l_text + extra + 1;
    for (int32_t i = 0; i < n_ref; i++) need += 8 + strlen(names[i]) + 1;
    uint8_t *h = malloc(need), *p = h;
    if (!h) return fail(CRUMBLE_BAM_ENOMEM, "out of memory");
    memcpy(p, "BAM\1", 4); p += 4;
    const int addnl = l_text > 0 && text[l_text - 1] != '\n';
    put32(p, (uint32_t)((size_t)l_text + addnl + extra)); p += 4;
    memcpy(p, text, (size_t)l_text); p += l_text;
    if (addnl) *p++ = '\n';
    if (pg_line) { memcpy(p, pg_line, extra - 1); p += extra - 1; *p++ = '\n'; }
    put32(p, (uint32_t)n_ref); p += 4;
    for (int32_t i = 0; i < n_ref; i++) {
        const size_t ln = strlen(names[i]) + 1;
        put32(p, (uint32_t)ln); p += 4;
        memcpy(p, names[i], ln); p += ln;
        put32(p, (uint32_t)lens[i]); p += 4;
    }
    const int rc = bgzf_write(w, h, (size_t)(p - h));
    free(h);
    return rc;
}

static crumble_bam_writer *writer_open(const char *path, int level, int threads, int *err)
{
    crumble_bam_writer *w = calloc(1, sizeof *w);
    if (!w) { if (err) *err = fail(CRUMBLE_BAM_ENOMEM, "out of memory"); return NULL; }
    w->level = level < 0 ? Z_DEFAULT_COMPRESSION : (level > 9 ? 9 : level);
    w->threads = threads > 0 ? threads : default_threads();
    w->tm = team_create(w->threads);
    if (!w->tm) { if (err) *err = fail(CRUMBLE_BAM_ENOMEM, "out of memory"); free(w); return NULL; }
    w->tm->slot_free = deflate_tls_free;
    w->fp = fopen(path, "wb");
    if (!w->fp) { if (err) *err = fail(CRUMBLE_BAM_EIO, "%s: %s", path, strerror(errno)); team_destroy(w->tm); free(w); return NULL; }
    return w;
}

crumble_bam_writer *crumble_bam_create(const char *path, const crumble_bam_reader *r, const char *pg_line,
                                       int level, int threads, int *err)
{
    crumble_bam_writer *w = writer_open(path, level, threads, err);
    if (!w) return NULL;
    const int rc = write_header(w, r->text, r->l_text, pg_line, r->n_ref, (const char *const *)r->ref_name, r->ref_len);
    if (rc) { if (err) *err = rc; fclose(w->fp); team_destroy(w->tm); big_free(w->obuf, w->ocap); free(w); return NULL; }
    if (err) *err = 0;
    return w;
}

/* -t / -T lists: "XX,YY,ZZ" exactly as parse_aux_list (snp_score.c:2031-2054) reads them */
static int parse_aux_list(uint8_t *set, const char *arg)
{
    while (strlen(arg) >= 2) {
        const unsigned x = (unsigned)(uint8_t)arg[0] << 8 | (uint8_t)arg[1];
        set[x >> 3] |= (uint8_t)(1u << (x & 7));
        arg += 2;
        if (*arg == ',') arg++;
        else if (*arg != 0) break;
    }
    return strlen(arg) ? -1 : 0;
}

int crumble_bam_set_aux_filter(crumble_bam_writer *w, const char *whitelist, const char *blacklist,
                               const int bd[3], const int bi[3])
{
    w->aux_mode = 0;
    if (whitelist || blacklist) {
        if (!w->aux_set && !(w->aux_set = calloc(8192, 1))) return fail(CRUMBLE_BAM_ENOMEM, "out of memory");
        memset(w->aux_set, 0, 8192);
        /* a white list wins over a black list (the else-if of snp_score.c:990,1007) */
        if (parse_aux_list(w->aux_set, whitelist ? whitelist : blacklist))
            return fail(CRUMBLE_EINVAL, "auxiliary tags should be exactly two characters long");
        w->aux_mode = whitelist ? 1 : 2;
    }
    for (int k = 0; k < 3; k++) { w->bd[k] = bd ? bd[k] : 0; w->bi[k] = bi ? bi[k] : 0; }
    /* the reference would write NUL bytes into the tag (and then abort in skip_aux) when a bin
     * value is left at 0: refuse instead */
    if ((w->bd[0] || w->bd[1] || w->bd[2]) && (!w->bd[0] || !w->bd[2])) return fail(CRUMBLE_EINVAL, "BD binning needs both -e and -g");
    if ((w->bi[0] || w->bi[1] || w->bi[2]) && (!w->bi[0] || !w->bi[2])) return fail(CRUMBLE_EINVAL, "BI binning needs both -E and -G");
    for (int k = 0; k < 3; k++)
        if (w->bd[k] < 0 || w->bd[k] > 255 || w->bi[k] < 0 || w->bi[k] > 255) return fail(CRUMBLE_EINVAL, "BD/BI bin value out of range");
    return 0;
}

/* end of the aux field that starts at s (after its two tag characters), or NULL when malformed */
static const uint8_t *aux_end(const uint8_t *s, const uint8_t *lim)
{
    if (s >= lim) return NULL;
    const int t = *s++;
    size_t sz;
    switch (t) {
    case 'A': case 'c': case 'C': sz = 1; break;
    case 's': case 'S': sz = 2; break;
    case 'i': case 'I': case 'f': sz = 4; break;
    case 'd': sz = 8; break;
    case 'Z': case 'H':
        while (s < lim && *s) s++;
        return s < lim ? s + 1 : NULL;
    case 'B': {
        if (s + 5 > lim) return NULL;
        const int st = *s++;
        const size_t es = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : (st == 'i' || st == 'I' || st == 'f') ? 4 : 0;
        if (!es) return NULL;
        const uint32_t n = le32(s); s += 4;
        return ((size_t)(lim - s) >= es * (size_t)n) ? s + es * (size_t)n : NULL;
    }
    default: return NULL;
    }
    return ((size_t)(lim - s) >= sz) ? s + sz : NULL;
}

static inline int aux_keep(const crumble_bam_writer *w, const uint8_t *t)
{
    if (!w->aux_mode) return 1;
    const unsigned x = (unsigned)t[0] << 8 | t[1];
    const int in = (w->aux_set[x >> 3] >> (x & 7)) & 1;
    return w->aux_mode == 1 ? in : !in;
}

/* BD / BI quantisation of one kept Z tag in place (snp_score.c:1025-1052) */
static inline void aux_bin(const crumble_bam_writer *w, uint8_t *t)
{
    const int *b = (t[0] == 'B' && t[1] == 'D' && (w->bd[0] | w->bd[1] | w->bd[2])) ? w->bd
                 : (t[0] == 'B' && t[1] == 'I' && (w->bi[0] | w->bi[1] | w->bi[2])) ? w->bi : NULL;
    if (!b || t[2] != 'Z') return;
    for (uint8_t *c = t + 3; *c; c++) *c = (uint8_t)((*c >= b[1]) ? b[2] : b[0]);
}

/* the pieces of crumble_bam_write_group(), records [lo, hi) each */
typedef struct { crumble_bam_writer *w; crumble_bam_reader *r; const uint8_t *qual_out; int binning; atomic_int bad; const crumble_coded *coded; } wg_job;
static void wg_scatter(void *ctx, int64_t lo, int64_t hi, int tid)
{
    wg_job *j = ctx; crumble_bam_reader *r = j->r;
    (void)tid;
    if (j->coded) {
        /* the coded result (include/crumble_hip.h): expanded straight into the records.  A record's quality field still
         * holds the input qualities, which is what code 2 refers to; the cursor walks the reads of this piece in base order */
        crumble_decode_cursor cur;
        crumble_decode_seek(j->coded, r->seq_off[lo], &cur);
        for (int64_t i = lo; i < hi; i++) {
            uint8_t *p = r->raw + r->rec_off[i];
            const int32_t lseq = (int32_t)le32(p + F_LSEQ);
            uint8_t *ql = p + F_FIXED + p[F_LNAME] + 4 * (size_t)le16(p + F_NCIG) + ((size_t)lseq + 1) / 2;
            if (lseq > 0) crumble_decode_run(j->coded, &cur, lseq, ql, ql[0] != 0xff ? ql : NULL);
        }
        return;
    }
    for (int64_t i = lo; i < hi; i++) {
        uint8_t *p = r->raw + r->rec_off[i];
        const int32_t lseq = (int32_t)le32(p + F_LSEQ);
        uint8_t *ql = p + F_FIXED + p[F_LNAME] + 4 * (size_t)le16(p + F_NCIG) + ((size_t)lseq + 1) / 2;
        if (lseq > 0 && ql[0] != 0xff) memcpy(ql, j->qual_out + r->seq_off[i], (size_t)lseq);
    }
}
static void wg_aux_sizes(void *ctx, int64_t lo, int64_t hi, int tid)
{
    wg_job *j = ctx; crumble_bam_reader *r = j->r; crumble_bam_writer *w = j->w;
    (void)tid;
    for (int64_t i = lo; i < hi; i++) {
        const uint8_t *p = r->raw + r->rec_off[i];
        const uint32_t bs = le32(p);
        const int32_t lseq = (int32_t)le32(p + F_LSEQ);
        const uint8_t *aux = p + F_FIXED + p[F_LNAME] + 4 * (size_t)le16(p + F_NCIG) + ((size_t)lseq + 1) / 2 + (size_t)lseq;
        const uint8_t *lim = p + 4 + bs;
        size_t keep = (size_t)(aux - p);
        for (const uint8_t *t = aux; t < lim; ) {
            const uint8_t *e = (t + 3 <= lim) ? aux_end(t + 2, lim) : NULL;
            if (!e) { atomic_store(&j->bad, 1); break; }
            if (aux_keep(w, t)) keep += (size_t)(e - t);
            t = e;
        }
        w->noff[i] = keep;
    }
}
static void wg_aux_copy(void *ctx, int64_t lo, int64_t hi, int tid)
{
    wg_job *j = ctx; crumble_bam_reader *r = j->r; crumble_bam_writer *w = j->w;
    (void)tid;
    for (int64_t i = lo; i < hi; i++) {
        const uint8_t *p = r->raw + r->rec_off[i];
        const uint32_t bs = le32(p);
        const int32_t lseq = (int32_t)le32(p + F_LSEQ);
        const uint8_t *aux = p + F_FIXED + p[F_LNAME] + 4 * (size_t)le16(p + F_NCIG) + ((size_t)lseq + 1) / 2 + (size_t)lseq;
        const uint8_t *lim = p + 4 + bs;
        uint8_t *o = w->raw2 + w->noff[i], *q = o;
        memcpy(q, p, (size_t)(aux - p)); q += aux - p;
        for (const uint8_t *t = aux; t < lim; ) {
            const uint8_t *e = aux_end(t + 2, lim);
            if (aux_keep(w, t)) { memcpy(q, t, (size_t)(e - t)); if (j->binning) aux_bin(w, q); q += e - t; }
            t = e;
        }
        put32(o, (uint32_t)(q - o - 4));
    }
}

static int write_group(crumble_bam_writer *w, crumble_bam_reader *r, const uint8_t *qual_out, const crumble_coded *coded);
int crumble_bam_write_group(crumble_bam_writer *w, crumble_bam_reader *r, const uint8_t *qual_out) { return write_group(w, r, qual_out, NULL); }
int crumble_bam_write_group_coded(crumble_bam_writer *w, crumble_bam_reader *r, const crumble_coded *c)
{
    if (r->n && (!c || c->n_bases != r->seq_off[r->n])) return fail(CRUMBLE_EINVAL, "crumble_bam_write_group_coded: the coded result is not this group's");
    return write_group(w, r, NULL, c);
}
int crumble_bam_write_detached_coded(crumble_bam_writer *w, crumble_bam_group *g, const crumble_coded *c)
{
    return crumble_bam_write_group_coded(w, (crumble_bam_reader *)g, c);
}

static int write_group(crumble_bam_writer *w, crumble_bam_reader *r, const uint8_t *qual_out, const crumble_coded *coded)
{
    if (r->n == 0) return 0;
    const double tsc = now_s();
    const int binning = w->bd[0] | w->bd[1] | w->bd[2] | w->bi[0] | w->bi[1] | w->bi[2];
    wg_job job = { w, r, qual_out, binning, 0, coded };
    atomic_init(&job.bad, 0);
    if (qual_out || coded) team_run(w->tm, r->n, 0, wg_scatter, &job);
    const size_t first = r->rec_off[0];
    if (!w->aux_mode && !binning) {
        T_scatter += now_s() - tsc;
        return bgzf_write(w, r->raw + first, r->group_end - first);
    }
    /* purge_tags: records shrink, so they are re-packed into a second buffer (sizes first, then a
     * prefix sum, then every record is copied by its own thread) */
    const int64_t n = r->n;
    if (w->noff_cap < (size_t)n + 1) {
        free(w->noff);
        w->noff = malloc(((size_t)n + 1) * sizeof *w->noff);
        w->noff_cap = w->noff ? (size_t)n + 1 : 0;
        if (!w->noff) return fail(CRUMBLE_BAM_ENOMEM, "out of memory");
    }
    team_run(w->tm, n, 0, wg_aux_sizes, &job);
    if (atomic_load(&job.bad)) return fail(CRUMBLE_BAM_EFORMAT, "malformed auxiliary field");
    size_t tot = 0;
    for (int64_t i = 0; i < n; i++) { const size_t k = w->noff[i]; w->noff[i] = tot; tot += k; }
    w->noff[n] = tot;
    if (w->raw2_cap < tot + 64) {
        big_free(w->raw2, w->raw2_cap);
        const size_t c = (tot + tot / 8 + ((size_t)2 << 20)) & ~(((size_t)2 << 20) - 1);
        w->raw2 = big_alloc(c);
        w->raw2_cap = w->raw2 ? c : 0;
        if (!w->raw2) return fail(CRUMBLE_BAM_ENOMEM, "out of memory");
    }
    team_run(w->tm, n, 0, wg_aux_copy, &job);
    T_scatter += now_s() - tsc;
    return bgzf_write(w, w->raw2, tot);
}

/* members that are BGZF already (the device pipeline's output: crumble_stream_drain) */
int crumble_bam_write_members(crumble_bam_writer *w, const uint8_t *members, size_t n)
{
    if (!w) return CRUMBLE_BAM_EIO;
    const double tw = now_s();
    if (n && fwrite(members, 1, n, w->fp) != n) return fail(CRUMBLE_BAM_EIO, "write error: %s", strerror(errno));
    T_fwrite += now_s() - tw;
    return 0;
}

int crumble_bam_writer_close(crumble_bam_writer *w)
{
    if (!w) return 0;
    int rc = 0;
    if (fwrite(BGZF_EOF, 1, 28, w->fp) != 28) rc = fail(CRUMBLE_BAM_EIO, "write error: %s", strerror(errno));
    if (fclose(w->fp) != 0 && !rc) rc = fail(CRUMBLE_BAM_EIO, "close: %s", strerror(errno));
    team_destroy(w->tm);
    big_free(w->obuf, w->ocap); big_free(w->raw2, w->raw2_cap); free(w->noff); free(w->aux_set); free(w);
    return rc;
}

/* --------------------------------------------------- SoA -> BAM (test helper) */
static int reg2bin(int64_t beg, int64_t end)
{   /* SAMv1 5.3 */
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

/* records [lo, hi) of one batch -> BAM records (test helper below) */
typedef struct { const crumble_batch *b; uint8_t *buf; const size_t *off; int64_t serial; } wb_job;
static void wb_pack(void *ctx, int64_t lo, int64_t hi, int tid)
{
    wb_job *j = ctx;
    const crumble_batch *b = j->b; uint8_t *buf = j->buf; const size_t *off = j->off; const int64_t serial = j->serial;
    (void)tid;
    for (int64_t i = lo; i < hi; i++) {
        uint8_t *p = buf + off[i];
        char nm[32];
        const int ln = snprintf(nm, sizeof nm, "r%lld", (long long)(serial + i)) + 1;
        const int64_t lq = b->seq_off[i + 1] - b->seq_off[i], ncg = b->cig_off[i + 1] - b->cig_off[i];
        const uint32_t *cg = b->cigar + b->cig_off[i];
        int64_t rlen = 0;
        for (int64_t k = 0; k < ncg; k++) { const unsigned op = cg[k] & 15; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rlen += cg[k] >> 4; }
        put32(p, (uint32_t)(off[i + 1] - off[i] - 4));
        put32(p + F_REFID, (uint32_t)b->tid);
        put32(p + F_POS, (uint32_t)b->pos[i]);
        p[F_LNAME] = (uint8_t)ln; p[F_MAPQ] = b->mapq[i];
        put16(p + 14, (uint16_t)reg2bin(b->pos[i] < 0 ? 0 : b->pos[i], (b->pos[i] < 0 ? 0 : b->pos[i]) + (rlen ? rlen : 1)));
        put16(p + F_NCIG, (uint16_t)ncg); put16(p + F_FLAG, b->flag[i]);
        put32(p + F_LSEQ, (uint32_t)lq);
        put32(p + 24, (uint32_t)-1); put32(p + 28, (uint32_t)-1); put32(p + 32, 0);
        uint8_t *q = p + F_FIXED;
        memcpy(q, nm, (size_t)ln); q += ln;
        for (int64_t k = 0; k < ncg; k++) { put32(q, cg[k]); q += 4; }
        const uint8_t *bs = b->bases + b->seq_off[i];
        for (int64_t k = 0; k + 1 < lq; k += 2) *q++ = (uint8_t)((bs[k] & 15) << 4 | (bs[k + 1] & 15));
        if (lq & 1) *q++ = (uint8_t)((bs[lq - 1] & 15) << 4);
        memcpy(q, b->quals + b->seq_off[i], (size_t)lq);
    }
}

int crumble_bam_write_batches(const char *path, int32_t n_ref, const char *const *names, const int64_t *lens,
                              int32_t n_batches, const crumble_batch *batches, int level, int threads)
{
    int err = 0;
    crumble_bam_writer *w = writer_open(path, level, threads, &err);
    if (!w) return err;
    static const char text[] = "@HD\tVN:1.6\tSO:coordinate\n";
    int rc = write_header(w, text, (int64_t)strlen(text), NULL, n_ref, names, lens);
    int64_t serial = 0;
    for (int32_t bi = 0; bi < n_batches && !rc; bi++) {
        const crumble_batch *b = &batches[bi];
        const int64_t n = b->n_reads;
        size_t *off = malloc(((size_t)n + 1) * sizeof *off);
        if (!off) { rc = fail(CRUMBLE_BAM_ENOMEM, "out of memory"); break; }
        size_t tot = 0;
        for (int64_t i = 0; i < n; i++) {
            char nm[32];
            const int ln = snprintf(nm, sizeof nm, "r%lld", (long long)(serial + i)) + 1;
            const int64_t lq = b->seq_off[i + 1] - b->seq_off[i], ncg = b->cig_off[i + 1] - b->cig_off[i];
            off[i] = tot;
            tot += (size_t)F_FIXED + (size_t)ln + 4 * (size_t)ncg + ((size_t)lq + 1) / 2 + (size_t)lq;
        }
        off[n] = tot;
        uint8_t *buf = malloc(tot + 1);
        if (!buf) { free(off); rc = fail(CRUMBLE_BAM_ENOMEM, "out of memory"); break; }
        wb_job wj = { b, buf, off, serial };
        team_run(w->tm, n, 0, wb_pack, &wj);
        rc = bgzf_write(w, buf, tot);
        serial += n;
        free(buf); free(off);
    }
    const int rc2 = crumble_bam_writer_close(w);
    return rc ? rc : rc2;
}

/* ------------------------------------------------------------------- -R BED */
int crumble_bam_load_bed(const crumble_bam_reader *r, const char *path, int32_t **tid, int32_t **start, int32_t **end)
{
    FILE *fp = fopen(path, "r");
    if (!fp) return fail(CRUMBLE_BAM_EIO, "%s: %s", path, strerror(errno));
    char line[8192], chr[8192];
    int n = 0, cap = 0, s, e;
    int32_t *t = NULL, *st = NULL, *en = NULL;
    while (fgets(line, sizeof line, fp)) {
        /* same line filter and field scan as bed_load (bed.c:55-64) */
        if (line[0] == '#' || !strncmp(line, "track", 5) || !strncmp(line, "browser", 7) || line[0] == '\n') continue;
        if (sscanf(line, "%8191s %d %d", chr, &s, &e) != 3) { fclose(fp); free(t); free(st); free(en); return fail(CRUMBLE_BAM_EFORMAT, "malformed bed line: %s", line); }
        int id = -1;
        for (int32_t i = 0; i < r->n_ref; i++) if (!strcmp(r->ref_name[i], chr)) { id = i; break; }
        if (id < 0) { fclose(fp); free(t); free(st); free(en); return fail(CRUMBLE_BAM_EFORMAT, "unknown reference name: %s", chr); }
        if (n == cap) {
            cap = cap ? cap * 2 : 1024;
            t = realloc(t, cap * sizeof *t); st = realloc(st, cap * sizeof *st); en = realloc(en, cap * sizeof *en);
            if (!t || !st || !en) { fclose(fp); return fail(CRUMBLE_BAM_ENOMEM, "out of memory"); }
        }
        t[n] = id; st[n] = s; en[n] = e; n++;
    }
    fclose(fp);
    if (n) n = crumble_bed_collapse(t, st, en, n);
    *tid = t; *start = st; *end = en;
    return n;
}

/* ------------------------------------------------------------- htslib backend
 * SAM / BAM / CRAM in and out through a system libhts, resolved at run time: what reference
 * crumble does with sam_open_format / sam_hdr_read / sam_read1 / sam_write1 / sam_hdr_add_PG
 * (snp_score.c:2201-2207, 2561-2626, 1116-1117, 1101).  htslib is not in the build image and not under
 * /root/reference, so nothing here can be exercised offline beyond "library absent -> CRUMBLE_BAM_ENOHTS";
 * the code targets the libhts.so.3 ABI (htslib >= 1.10: 64-bit positions in bam1_core_t) and refuses anything
 * else.  Records of one contig are collected as bam1_t objects, unpacked into the same SoA crumble_batch the
 * native reader produces, and written back with the engine's qualities in place. */
#include <dlfcn.h>

typedef struct {                /* htslib/sam.h, libhts.so.3 */
    int64_t pos; int32_t tid; uint16_t bin; uint8_t qual, l_extranul; uint16_t flag, l_qname;
    uint32_t n_cigar; int32_t l_qseq, mtid; int64_t mpos, isize;
} hts_core_t;
typedef struct { hts_core_t core; uint64_t id; uint8_t *data; int l_data; uint32_t m_data; uint32_t mempolicy; } hts_bam1_t;
typedef char hts_core_size_check[sizeof(hts_core_t) == 48 ? 1 : -1];

static struct {
    void *lib; int tried, ok;
    const char *(*hts_version)(void);
    int   (*hts_parse_format)(void *fmt, const char *str);
    void *(*hts_open_format)(const char *fn, const char *mode, const void *fmt);
    int   (*sam_open_mode)(char *mode, const char *fn, const char *format);
    int   (*hts_close)(void *fp);
    int   (*hts_set_threads)(void *fp, int n);
    void *(*sam_hdr_read)(void *fp);
    int   (*sam_hdr_write)(void *fp, const void *h);
    void  (*sam_hdr_destroy)(void *h);
    int   (*sam_hdr_nref)(const void *h);
    const char *(*sam_hdr_tid2name)(const void *h, int tid);
    int64_t (*sam_hdr_tid2len)(const void *h, int tid);
    int   (*sam_hdr_add_pg)(void *h, const char *name, ...);
    int   (*sam_read1)(void *fp, void *h, hts_bam1_t *b);
    int   (*sam_write1)(void *fp, const void *h, const hts_bam1_t *b);
    hts_bam1_t *(*bam_init1)(void);
    void  (*bam_destroy1)(hts_bam1_t *b);
    void *(*sam_hdr_parse)(size_t l_text, const char *text);
    int   (*sam_parse1)(void *kstr, void *h, hts_bam1_t *b);
} H;

/* The structs above are a private copy of htslib's bam1_core_t / bam1_t.  Before anything is trusted, a record made BY
 * THE LIBRARY (one SAM line through sam_parse1) is read through the copy: every field this file uses and the layout of
 * the data block (qname, CIGAR words, packed bases, qualities, aux) must come out as written.  A libhts whose bam1_t
 * differs is refused (CRUMBLE_BAM_ENOHTS) instead of corrupting records silently. */
static int hts_layout_check(void)
{
    static const char hdr_text[] = "@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:chrT\tLN:100000\n";
    char line[] = "rd1\t99\tchrT\t1001\t37\t2S5M1I2M\t=\t1201\t210\tACGTNACGTA\t!+5?IIII#~\tNM:i:3\tXS:Z:ab";
    struct { size_t l, m; char *s; } ks = { sizeof line - 1, sizeof line, line };      /* kstring_t */
    void *hdr = H.sam_hdr_parse(sizeof hdr_text - 1, hdr_text);
    hts_bam1_t *b = H.bam_init1();
    int ok = hdr && b && H.sam_parse1(&ks, hdr, b) >= 0;
    if (ok) {
        const hts_core_t *c = &b->core;
        ok = c->pos == 1000 && c->tid == 0 && c->qual == 37 && c->flag == 99 && c->n_cigar == 4 && c->l_qseq == 10 &&
             c->mtid == 0 && c->mpos == 1200 && c->isize == 210 && c->l_qname >= 4 && c->l_qname % 4 == 0 &&
             c->l_extranul == c->l_qname - 4;
        const size_t fixed = (size_t)c->l_qname + 16 + 5 + 10;                  /* qname, 4 CIGAR words, 5 base bytes, 10 qualities; then the aux fields */
        ok = ok && b->data && b->l_data >= (int)fixed + 4 + 6 && b->l_data <= (int)fixed + 32 && b->m_data >= (uint32_t)b->l_data;
        if (ok) {
            const uint8_t *d = b->data;
            ok = !memcmp(d, "rd1", 4);
            const uint8_t *cg = d + c->l_qname;
            ok = ok && le32(cg) == (2u << 4 | 4u) && le32(cg + 4) == (5u << 4 | 0u) && le32(cg + 8) == (1u << 4 | 1u) && le32(cg + 12) == (2u << 4 | 0u);
            const uint8_t *sq = cg + 16;                                          /* A C G T N A C G T A -> 1 2 4 8 15 1 2 4 8 1 */
            ok = ok && sq[0] == 0x12 && sq[1] == 0x48 && sq[2] == 0xf1 && sq[3] == 0x24 && sq[4] == 0x81;
            const uint8_t *ql = sq + 5;
            ok = ok && ql[0] == 0 && ql[1] == 10 && ql[2] == 20 && ql[3] == 30 && ql[4] == 40 && ql[8] == 2 && ql[9] == 93;
            const uint8_t *ax = ql + 10;
            ok = ok && ax[0] == 'N' && ax[1] == 'M' && !memcmp(d + b->l_data - 6, "XSZab", 6);     /* (NM's integer width is the library's choice) */
        }
    }
    if (b) H.bam_destroy1(b);
    if (hdr) H.sam_hdr_destroy(hdr);
    return ok ? 0 : fail(CRUMBLE_BAM_ENOHTS, "this libhts lays bam1_t out differently from the libhts.so.3 ABI this build knows (a record "
                                             "parsed by the library does not read back through it): refusing to use it");
}

static int hts_load(void)
{
    if (H.tried) return H.ok ? 0 : CRUMBLE_BAM_ENOHTS;
    H.tried = 1;
    H.lib = dlopen("libhts.so.3", RTLD_NOW | RTLD_LOCAL);
    if (!H.lib) H.lib = dlopen("libhts.so", RTLD_NOW | RTLD_LOCAL);
    if (!H.lib) return fail(CRUMBLE_BAM_ENOHTS, "libhts.so.3 not found: SAM / CRAM and -I / -O need a system htslib (>= 1.10); BAM works without it");
#define HSYM(name) do { *(void **)&H.name = dlsym(H.lib, #name); if (!H.name) return fail(CRUMBLE_BAM_ENOHTS, "libhts lacks %s (htslib >= 1.10 is required)", #name); } while (0)
    HSYM(hts_version); HSYM(hts_parse_format); HSYM(hts_open_format); HSYM(sam_open_mode); HSYM(hts_close);
    HSYM(hts_set_threads); HSYM(sam_hdr_read); HSYM(sam_hdr_write); HSYM(sam_hdr_destroy); HSYM(sam_hdr_nref);
    HSYM(sam_hdr_tid2name); HSYM(sam_hdr_tid2len); HSYM(sam_hdr_add_pg); HSYM(sam_read1); HSYM(sam_write1);
    HSYM(bam_init1); HSYM(bam_destroy1); HSYM(sam_hdr_parse); HSYM(sam_parse1);
#undef HSYM
    const char *v = H.hts_version();
    int maj = 0, min = 0;
    if (!v || sscanf(v, "%d.%d", &maj, &min) < 2 || maj != 1 || min < 10)
        return fail(CRUMBLE_BAM_ENOHTS, "htslib %s: need 1.10 or later (64-bit positions in bam1_core_t)", v ? v : "?");
    if (hts_layout_check()) return CRUMBLE_BAM_ENOHTS;
    H.ok = 1;
    return 0;
}

int crumble_hts_available(char *version, size_t len)
{
    if (hts_load()) return 0;
    if (version && len) snprintf(version, len, "%s", H.hts_version());
    return 1;
}

struct crumble_hts {
    void *in, *out, *hdr;
    uint64_t in_fmt[8], out_fmt[8];          /* htsFormat, opaque (32 bytes in libhts.so.3) */
    hts_bam1_t **rec; int64_t n, cap;        /* the current group's records */
    hts_bam1_t *ahead;                       /* first record of the next group */
    hts_bam1_t **pool; int64_t npool, cappool;
    int eof;
    team *tm;                                /* host threads of the SoA unpack (team.h) */
    int threads;                             /* what crumble_hts_open was given (-@): the unpack team's size too */
    crumble_bam_writer filt;                 /* purge_tags settings only */
    /* SoA views */
    int32_t *pos; uint8_t *mapq; uint16_t *flag; int64_t *seq_off, *cig_off; uint32_t *cigar; uint8_t *bases, *quals;
    size_t cap_reads, cap_bases, cap_cig;
};

static hts_bam1_t *hts_get(crumble_hts *h)
{
    if (h->npool) return h->pool[--h->npool];
    return H.bam_init1();
}

crumble_hts *crumble_hts_open(const char *in_fn, const char *in_fmt, const char *out_fn, const char *out_fmt,
                              const char *pg_name, const char *pg_version, const char *pg_cmdline, int threads, int *err)
{
    int e = hts_load();
    if (e) { if (err) *err = e; return NULL; }
    crumble_hts *h = calloc(1, sizeof *h);
    if (!h) { if (err) *err = CRUMBLE_BAM_ENOMEM; return NULL; }
    if ((in_fmt && H.hts_parse_format(h->in_fmt, in_fmt) < 0) || (out_fmt && H.hts_parse_format(h->out_fmt, out_fmt) < 0)) {
        e = fail(CRUMBLE_EINVAL, "bad -I / -O format string");
        goto bad;
    }
    if (!(h->in = H.hts_open_format(in_fn, "r", in_fmt ? h->in_fmt : NULL))) { e = fail(CRUMBLE_BAM_EIO, "cannot open %s", in_fn); goto bad; }
    char mode[8] = "w";
    H.sam_open_mode(mode + 1, out_fn, NULL);                 /* format by file name, snp_score.c:2569 */
    if (!(h->out = H.hts_open_format(out_fn, mode, out_fmt ? h->out_fmt : NULL))) { e = fail(CRUMBLE_BAM_EIO, "cannot open %s", out_fn); goto bad; }
    h->threads = threads;
    if (threads > 1) { H.hts_set_threads(h->in, threads); H.hts_set_threads(h->out, threads); }
    if (!(h->hdr = H.sam_hdr_read(h->in))) { e = fail(CRUMBLE_BAM_EFORMAT, "failed to read the file header"); goto bad; }
    if (pg_name && H.sam_hdr_add_pg(h->hdr, pg_name, "VN", pg_version ? pg_version : "0", pg_cmdline ? "CL" : NULL, pg_cmdline, NULL) != 0) {
        e = fail(CRUMBLE_BAM_EFORMAT, "failed to add the @PG line"); goto bad;
    }
    if (H.sam_hdr_write(h->out, h->hdr) != 0) { e = fail(CRUMBLE_BAM_EIO, "failed to write the file header"); goto bad; }
    if (err) *err = 0;
    return h;
bad:
    if (err) *err = e;
    crumble_hts_close(h);
    return NULL;
}

int32_t crumble_hts_n_ref(const crumble_hts *h) { return H.sam_hdr_nref(h->hdr); }
const char *crumble_hts_ref_name(const crumble_hts *h, int32_t tid) { const char *s = tid >= 0 ? H.sam_hdr_tid2name(h->hdr, tid) : NULL; return s ? s : "*"; }
int crumble_hts_load_bed(const crumble_hts *h, const char *path, int32_t **tid, int32_t **start, int32_t **end)
{
    crumble_bam_reader fake;                                  /* bed_load only needs the reference names */
    memset(&fake, 0, sizeof fake);
    fake.n_ref = H.sam_hdr_nref(h->hdr);
    fake.ref_name = calloc((size_t)(fake.n_ref > 0 ? fake.n_ref : 1), sizeof *fake.ref_name);
    if (!fake.ref_name) return fail(CRUMBLE_BAM_ENOMEM, "out of memory");
    for (int32_t i = 0; i < fake.n_ref; i++) fake.ref_name[i] = (char *)H.sam_hdr_tid2name(h->hdr, i);
    const int n = crumble_bam_load_bed(&fake, path, tid, start, end);
    free(fake.ref_name);
    return n;
}
int crumble_hts_set_aux_filter(crumble_hts *h, const char *whitelist, const char *blacklist, const int bd[3], const int bi[3])
{ return crumble_bam_set_aux_filter(&h->filt, whitelist, blacklist, bd, bi); }

/* records [lo, hi) of the current group -> the SoA arrays */
static void hts_unpack(void *ctx, int64_t lo, int64_t hi, int tid)
{
    crumble_hts *h = ctx;
    (void)tid;
    for (int64_t i = lo; i < hi; i++) {
        const hts_bam1_t *r = h->rec[i];
        const int32_t lseq = r->core.l_qseq;
        h->pos[i] = (int32_t)r->core.pos; h->mapq[i] = r->core.qual; h->flag[i] = r->core.flag;
        const uint8_t *cg = r->data + r->core.l_qname;
        memcpy(h->cigar + h->cig_off[i], cg, 4 * (size_t)r->core.n_cigar);
        const uint8_t *sq = cg + 4 * (size_t)r->core.n_cigar;
        uint8_t *bo = h->bases + h->seq_off[i];
        for (int32_t k = 0; k + 1 < lseq; k += 2) { const uint8_t v = sq[k >> 1]; bo[k] = v >> 4; bo[k + 1] = v & 15; }
        if (lseq & 1) bo[lseq - 1] = sq[lseq >> 1] >> 4;
        const uint8_t *ql = sq + ((size_t)lseq + 1) / 2;
        uint8_t *qo = h->quals + h->seq_off[i];
        if (lseq > 0 && ql[0] == 0xff) memset(qo, 0, (size_t)lseq);     /* absent qualities */
        else memcpy(qo, ql, (size_t)lseq);
    }
}

/* the next run of records sharing one refID -> SoA views (same contract as crumble_bam_next_contig) */
int crumble_hts_next_contig(crumble_hts *h, crumble_batch *b)
{
    for (int64_t i = 0; i < h->n; i++) {                      /* recycle the previous group's records */
        if (h->npool == h->cappool) {
            const int64_t c = h->cappool ? h->cappool * 2 : 4096;
            hts_bam1_t **p = realloc(h->pool, (size_t)c * sizeof *p);
            if (!p) return fail(CRUMBLE_BAM_ENOMEM, "out of memory");
            h->pool = p; h->cappool = c;
        }
        h->pool[h->npool++] = h->rec[i];
    }
    h->n = 0;
    int32_t tid = 0;
    int64_t nb = 0, nc = 0;
    for (;;) {
        hts_bam1_t *r = h->ahead;
        h->ahead = NULL;
        if (!r) {
            if (h->eof) break;
            r = hts_get(h);
            if (!r) return fail(CRUMBLE_BAM_ENOMEM, "out of memory");
            const int rc = H.sam_read1(h->in, h->hdr, r);
            if (rc < -1) return fail(CRUMBLE_BAM_EFORMAT, "sam_read1 failed (%d)", rc);
            if (rc == -1) { h->eof = 1; H.bam_destroy1(r); break; }
        }
        if (h->n == 0) tid = r->core.tid;
        else if (r->core.tid != tid) { h->ahead = r; break; }
        if (r->core.pos > INT32_MAX - 64 || r->core.l_qseq < 0) return fail(CRUMBLE_BAM_EUNSUPP, "coordinate beyond 2^31");
        if (h->n == h->cap) {
            const int64_t c = h->cap ? h->cap * 2 : (1 << 16);
            hts_bam1_t **p = realloc(h->rec, (size_t)c * sizeof *p);
            if (!p) return fail(CRUMBLE_BAM_ENOMEM, "out of memory");
            h->rec = p; h->cap = c;
        }
        h->rec[h->n++] = r;
        nb += r->core.l_qseq; nc += r->core.n_cigar;
    }
    if (h->n == 0) return 0;
    const int64_t n = h->n;
    if (h->cap_reads < (size_t)n + 1) {
        const size_t c = (size_t)n + (size_t)n / 4 + 64;
        free(h->pos); free(h->mapq); free(h->flag); free(h->seq_off); free(h->cig_off);
        h->pos = malloc(c * 4); h->mapq = malloc(c); h->flag = malloc(c * 2); h->seq_off = malloc(c * 8); h->cig_off = malloc(c * 8);
        h->cap_reads = (h->pos && h->mapq && h->flag && h->seq_off && h->cig_off) ? c : 0;
        if (!h->cap_reads) return fail(CRUMBLE_BAM_ENOMEM, "out of memory");
    }
    if (h->cap_bases < (size_t)nb + 64) {
        big_free(h->bases, h->cap_bases); big_free(h->quals, h->cap_bases);
        const size_t c = (size_t)nb + (size_t)nb / 8 + 4096;
        h->bases = big_alloc(c); h->quals = big_alloc(c);
        h->cap_bases = (h->bases && h->quals) ? c : 0;
        if (!h->cap_bases) return fail(CRUMBLE_BAM_ENOMEM, "out of memory");
    }
    if (h->cap_cig < (size_t)nc + 1) {
        free(h->cigar);
        const size_t c = (size_t)nc + (size_t)nc / 8 + 64;
        h->cigar = malloc(c * 4);
        h->cap_cig = h->cigar ? c : 0;
        if (!h->cap_cig) return fail(CRUMBLE_BAM_ENOMEM, "out of memory");
    }
    int64_t ab = 0, ac = 0;
    for (int64_t i = 0; i < n; i++) { h->seq_off[i] = ab; h->cig_off[i] = ac; ab += h->rec[i]->core.l_qseq; ac += h->rec[i]->core.n_cigar; }
    h->seq_off[n] = ab; h->cig_off[n] = ac;
    if (!h->tm) h->tm = team_create(h->threads > 0 ? h->threads : default_threads());
    team_run(h->tm, n, 0, hts_unpack, h);               /* (no team: team_run does the loop itself) */
    memset(b, 0, sizeof *b);
    b->tid = tid; b->n_reads = n;
    b->pos = h->pos; b->mapq = h->mapq; b->flag = h->flag; b->seq_off = h->seq_off; b->cig_off = h->cig_off;
    b->cigar = h->cigar; b->bases = h->bases; b->quals = h->quals;
    return 1;
}

/* flush_bam_list() (snp_score.c:1081-1105): qualities replaced, purge_tags applied, sam_write1 per record */
int crumble_hts_write_group(crumble_hts *h, const uint8_t *qual_out)
{
    const crumble_bam_writer *w = &h->filt;
    const int binning = w->bd[0] | w->bd[1] | w->bd[2] | w->bi[0] | w->bi[1] | w->bi[2];
    for (int64_t i = 0; i < h->n; i++) {
        hts_bam1_t *r = h->rec[i];
        const int32_t lseq = r->core.l_qseq;
        uint8_t *ql = r->data + r->core.l_qname + 4 * (size_t)r->core.n_cigar + ((size_t)lseq + 1) / 2;
        if (qual_out && lseq > 0 && ql[0] != 0xff) memcpy(ql, qual_out + h->seq_off[i], (size_t)lseq);
        if (w->aux_mode || binning) {                        /* purge_tags, in place (records only shrink) */
            uint8_t *aux = ql + lseq, *lim = r->data + r->l_data, *q = aux;
            for (uint8_t *t = aux; t < lim; ) {
                const uint8_t *e2 = (t + 3 <= lim) ? aux_end(t + 2, lim) : NULL;
                if (!e2) return fail(CRUMBLE_BAM_EFORMAT, "malformed auxiliary field");
                const size_t len = (size_t)(e2 - t);
                if (aux_keep(w, t)) { memmove(q, t, len); if (binning) aux_bin(w, q); q += len; }
                t += len;
            }
            r->l_data = (int)(q - r->data);
        }
        if (H.sam_write1(h->out, h->hdr, r) < 0) return fail(CRUMBLE_BAM_EIO, "sam_write1 failed");
    }
    return 0;
}

int crumble_hts_close(crumble_hts *h)
{
    int rc = 0;
    if (!h) return 0;
    if (H.ok) {
        for (int64_t i = 0; i < h->n; i++) H.bam_destroy1(h->rec[i]);
        for (int64_t i = 0; i < h->npool; i++) H.bam_destroy1(h->pool[i]);
        if (h->ahead) H.bam_destroy1(h->ahead);
        if (h->hdr) H.sam_hdr_destroy(h->hdr);
        if (h->in && H.hts_close(h->in) != 0) rc = fail(CRUMBLE_BAM_EIO, "error while closing the input");
        if (h->out && H.hts_close(h->out) != 0) rc = fail(CRUMBLE_BAM_EIO, "error while closing the output");
    }
    team_destroy(h->tm);
    free(h->rec); free(h->pool); free(h->pos); free(h->mapq); free(h->flag); free(h->seq_off); free(h->cig_off); free(h->cigar);
    big_free(h->bases, h->cap_bases); big_free(h->quals, h->cap_bases); free(h->filt.aux_set);
    free(h);
    return rc;
}
