/*
 * cli.c -- `crumble-hip`: file -> file driver around the HIP column engine.
 *
 * The same shape as reference crumble's main() + transcode() (snp_score.c:2144-2680, 1336-2030)
 * with the loop body replaced by crumble_hip_contig(): options are parsed with the reference's own
 * getopt string (snp_score.c:2199) in argv order (presets overwrite, later options refine), records
 * are gathered per contig (the unit the reference resets its state on, snp_score.c:1478-1488), the
 * engine rewrites the qualities, records go out in input order.  BAM goes through the native reader / writer
 * (bamio.c) and the contig pipeline below; SAM, CRAM and the -I / -O format strings go through a system
 * htslib when one can be dlopen'ed (bamio.c, "htslib backend") and stop with a clear error otherwise; -r
 * (index iterator) is refused rather than silently ignored.  There is no CPU path: without a GPU the
 * program stops with CRUMBLE_ENODEVICE.
 */
#define _GNU_SOURCE
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <unistd.h>
#include <time.h>
#include <pthread.h>
#include "../../include/crumble_hip.h"
#include "../../include/crumble_bamio.h"

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

static void usage(FILE *fp)
{
    fprintf(fp,
        "crumble-hip: crumble's quality quantisation on an AMD MI355X (BAM in, BAM out)\n"
        "Usage: crumble-hip [options] in.bam out.bam\n"
        "  -1 -3 -5 -7 -8 -9       compression presets (as crumble)\n"
        "  -y illumina|pbccs       technology profile\n"
        "  -q -d -x  /  -Q -D -X   min SNP qual, indel qual, discrepancy (consensus A / B)\n"
        "  -m N  -l N  -u N  -c N  -U N   min mapq, qlow, qhigh, qcutoff, qcap\n"
        "  -i mul[,add]  -s mul[,add]     indel / SNP STR window scaling\n"
        "  -L 0|1  -B              reduce qualities; binary qualities\n"
        "  -C -M -Z -P -Y -V f     clip / low-mapq / ins-len / over-depth / indel fraction / indel overlap\n"
        "  -p N   -S   -N          P-block, soft-clip quantisation, perfect columns\n"
        "  -k list  -K list        qualities to preserve (-K: always)\n"
        "  -R keep.bed             regions whose qualities are kept\n"
        "  -b out.bed              write the suspicious regions (DEEP, CLIP, INDEL_LEN, ...) as BED\n"
        "  -t list  -T list        auxiliary tags to keep (white list) / to drop (black list)\n"
        "  -e -f -g / -E -F -G n   BD / BI tag low, cutoff, high values\n"
        "  -z                      no @PG header line\n"
        "  -I fmt  -O fmt          input / output format[,options] as htslib understands them; SAM and CRAM files and\n"
        "                          these two options need a system libhts (>= 1.10), looked up at run time\n"
        "  -v                      statistics and timings on stderr\n"
        "  -@ N                    host threads for BGZF/BAM work\n"
        "  -j N                    GPUs: contigs are LPT-packed onto devices 0..N-1 by reference length, one engine\n"
        "                          (and one host thread) per device, output in input order, one RCCL all-reduce of\n"
        "                          the statistics at the end\n"
        "  -w                      with -j N: every contig is cut into N windows of about equal bases, one per device\n"
        "                          (reads around a cut go to both sides; the running depth average and the STR window\n"
        "                          state are handed from window to window), for inputs with fewer large contigs than GPUs\n"
        "  (devices: environment CRUMBLE_HIP_DEVICES=\"0,1,..\" or CRUMBLE_HIP_DEVICE, default 0; output level: CRUMBLE_BAM_LEVEL)\n");
}

static const char *const ST_NAME[] = {
    "columns", "het_A", "het_qual_A", "hom_A", "hom_qual_A", "discrep_A", "het_B", "het_qual_B", "hom_B",
    "hom_qual_B", "discrep_B", "diff", "indel", "indel_qual", "low_mqual_perc", "clip_perc", "ins_len_perc",
    "indel_ov_perc", "over_depth", "reads", "bases", "bases_changed", "triggers", "too_deep" };

/* ---- the contig pipeline: reader (main thread) -> one worker per GPU -> writer, contigs in flight are detached
 * groups (crumble_bam_detach_group).  Every cross-column state of the reference resets on a contig change
 * (snp_score.c:1478-1488), so contigs are independent units; the writer restores input order. */
#define MAX_DEV 16
#define MAX_INFLIGHT 4              /* per device */
typedef struct job {
    struct job *next;
    int64_t seq; int dev, done, err;
    crumble_bam_group *g; crumble_batch B;
    uint8_t *qout; size_t qcap;
    crumble_coded coded; int is_coded;     /* the result as 2 bits per base + escapes (buffers inside qout) */
    int32_t *sus_pos; uint8_t *sus_why; int64_t n_sus;     /* -b */
    char errtxt[256];
} job;
typedef struct {
    pthread_mutex_t mu; pthread_cond_t cv;
    job *q_head[MAX_DEV], *q_tail[MAX_DEV]; int q_len[MAX_DEV];
    job *done_head;                 /* finished jobs, any order */
    int64_t next_write; int closing, failed;
    int ndev, devices[MAX_DEV]; crumble_hip *eng[MAX_DEV]; void *comm[MAX_DEV];
    int want_suspect, use_rccl;
    int plain_download;             /* CRUMBLE_HIP_PLAIN_DOWNLOAD=1: one byte per base back from the device (tests compare both ways) */
    int coded_quals;                /* CRUMBLE_HIP_CODED_QUALS=1: qualities to the device as 2- / 4-bit codes when a contig has at most sixteen values */
    crumble_bam_writer *wr; crumble_bam_reader *rd; FILE *bed_fp;
    double t_gpu[MAX_DEV], t_write; int verbose;
    int64_t stats[CRUMBLE_ST_N];
} pipe_t;
typedef struct { pipe_t *p; int d; } worker_arg;

/* the device BGZF codec (crumble_gz_* of crumble_hip.h) behind libcrumble_bamio's codec hooks */
static int gz_inflate_cb(void *ctx, const uint8_t *comp, int64_t comp_len, const crumble_bam_block *blk, int64_t n_blocks,
                         uint8_t *raw, int64_t raw_len)
{
    return crumble_gz_inflate((crumble_gz *)ctx, comp, comp_len, (const crumble_gz_block *)blk, n_blocks, raw, raw_len, 1);
}
static int gz_deflate_cb(void *ctx, const uint8_t *raw, int64_t raw_len, int level, uint8_t *out, int64_t out_cap, int64_t *out_len)
{
    return crumble_gz_deflate((crumble_gz *)ctx, raw, raw_len, level, out, out_cap, out_len);
}
typedef char gz_block_layout_check[sizeof(crumble_bam_block) == sizeof(crumble_gz_block) ? 1 : -1];

/* ---- the device pipeline: the file's records never leave the GPU (crumble_stream, include/crumble_hip.h).  The host reads
 * compressed bytes, walks the BGZF member headers and writes compressed bytes; inflate, record delimiting, the engine's input,
 * its output back into the records and deflate happen in HBM.  Three threads: a reader (fread + crumble_bgzf_scan into one
 * of two page-locked chunk buffers), this one (feed / process / drain), a writer (fwrite out of one of two page-locked
 * output buffers).  One device, no tag filters, no -b: everything else takes the host pipeline below.  Returns 0 done,
 * 1 / 2 failed (message printed), -1 "not for this file": the caller starts over the host way (a CG-tag CIGAR). */
#ifndef PIPELINE_AUTO_DEVICE
#define PIPELINE_AUTO_DEVICE 1
#endif
/* compressed bytes per feed (the inflate kernel's time hardly depends on the members' number) and record bytes per drain, at
 * least (whole contig runs); CRUMBLE_HIP_DP_CHUNK / CRUMBLE_HIP_DP_PIECE (bytes) let the tests cut a small file many times */
static size_t DP_CHUNK = (size_t)256 << 20;
static int64_t DP_PIECE = (int64_t)2048 * 65280;      /* (2048 members: eight per CU) */
typedef struct { uint8_t *buf; size_t cap; int pinned; crumble_gz_block *blk; size_t bcap; int64_t nb, used; int eof, full; } dp_in;
typedef struct { uint8_t *buf; size_t cap; int pinned; int64_t len; int full; } dp_out;
typedef struct {
    pthread_mutex_t mu; pthread_cond_t cv;
    FILE *fp; crumble_bam_writer *wr;
    dp_in in[2]; dp_out out[2];
    int fail, stop, in_done;             /* fail: 1 error (message printed); stop: everybody leaves */
    double t_read, t_write;
} dp_t;

static uint8_t *dp_alloc(size_t n, int *pinned)
{
    uint8_t *p = crumble_hip_host_alloc(n);
    *pinned = p != NULL;
    return p ? p : crumble_bam_big_alloc(n);
}
typedef struct { uint8_t **buf; size_t cap; int *pinned; } dp_pre;
static void *dp_prealloc(void *a) { dp_pre *p = a; *p->buf = dp_alloc(p->cap, p->pinned); return NULL; }
static void dp_free(uint8_t *p, size_t n, int pinned) { if (!p) return; if (pinned) crumble_hip_host_free(p); else crumble_bam_big_free(p, n); }
static void dp_fail(dp_t *d, int code) { pthread_mutex_lock(&d->mu); if (!d->fail) d->fail = code; d->stop = 1; pthread_cond_broadcast(&d->cv); pthread_mutex_unlock(&d->mu); }

/* n bytes at offset off, by four threads when there are many (a cached file comes at the speed of one core's memcpy otherwise);
 * returns the bytes read (short at the end of the file), -1 on an error */
typedef struct { int fd; uint8_t *buf; size_t n; off_t off; ssize_t got; } pr_job;
static void *pr_run(void *a)
{
    pr_job *j = a;
    size_t done = 0;
    while (done < j->n) {
        const ssize_t r = pread(j->fd, j->buf + done, j->n - done, j->off + (off_t)done);
        if (r < 0) { j->got = -1; return NULL; }
        if (r == 0) break;
        done += (size_t)r;
    }
    j->got = (ssize_t)done;
    return NULL;
}
static ssize_t par_pread(int fd, uint8_t *buf, size_t n, off_t off)
{
    const int nt = n >= ((size_t)16 << 20) ? 4 : 1;
    pr_job j[4]; pthread_t t[4];
    const size_t per = (n / (size_t)nt + 4095) & ~(size_t)4095;
    for (int i = 0; i < nt; i++) {
        const size_t lo = per * (size_t)i, hi = i == nt - 1 ? n : (per * (size_t)(i + 1) < n ? per * (size_t)(i + 1) : n);
        j[i].fd = fd; j[i].buf = buf + lo; j[i].n = hi > lo ? hi - lo : 0; j[i].off = off + (off_t)lo; j[i].got = 0;
        if (i) pthread_create(&t[i], NULL, pr_run, &j[i]);
    }
    pr_run(&j[0]);
    for (int i = 1; i < nt; i++) pthread_join(t[i], NULL);
    ssize_t total = 0;
    for (int i = 0; i < nt; i++) {
        if (j[i].got < 0) return -1;
        total += j[i].got;
        if ((size_t)j[i].got < j[i].n) break;                   /* the end of the file lies in this part */
    }
    return total;
}

static void *dp_reader(void *a)
{
    dp_t *d = a;
    size_t carry = 0; const uint8_t *carry_src = NULL;
    off_t foff = 0;
    const int fd = fileno(d->fp);
    for (int slot = 0;; slot ^= 1) {
        dp_in *c = &d->in[slot];
        pthread_mutex_lock(&d->mu);
        while (c->full && !d->stop) pthread_cond_wait(&d->cv, &d->mu);
        const int stop = d->stop;
        pthread_mutex_unlock(&d->mu);
        if (stop) break;
        const double t0 = now();
        if (!c->buf && !(c->buf = dp_alloc(c->cap, &c->pinned))) { fprintf(stderr, "crumble-hip: out of memory\n"); dp_fail(d, 1); break; }
        if (carry) memmove(c->buf, carry_src, carry);           /* the member cut by the end of the chunk before */
        size_t have = carry;
        const ssize_t got = par_pread(fd, c->buf + have, c->cap - have, foff);
        if (got < 0) { fprintf(stderr, "crumble-hip: read error: %s\n", strerror(errno)); dp_fail(d, 1); break; }
        c->eof = (size_t)got < c->cap - have;
        foff += got;
        have += (size_t)got;
        int64_t used = 0, raw_total = 0, nb = 0;
        for (;;) {
            nb = crumble_bgzf_scan(c->buf, (int64_t)have, c->blk, (int64_t)c->bcap, &used, &raw_total);
            if (nb < 0 || (size_t)nb < c->bcap) break;
            c->bcap *= 2; free(c->blk);
            if (!(c->blk = malloc(c->bcap * sizeof *c->blk))) { nb = -2; break; }
        }
        if (nb < 0) { fprintf(stderr, nb == -2 ? "crumble-hip: out of memory\n" : "crumble-hip: not a BGZF block (bad gzip header)\n"); dp_fail(d, 1); break; }
        if (c->eof && (size_t)used != have) { fprintf(stderr, "crumble-hip: truncated BGZF block at the end of the file\n"); dp_fail(d, 1); break; }
        if (!c->eof && !used) { fprintf(stderr, "crumble-hip: a BGZF block larger than the read buffer\n"); dp_fail(d, 1); break; }
        c->nb = nb; c->used = used;
        carry = have - (size_t)used; carry_src = c->buf + used;
        const int eof = c->eof;
        pthread_mutex_lock(&d->mu);
        d->t_read += now() - t0;
        c->full = 1;
        pthread_cond_broadcast(&d->cv);
        pthread_mutex_unlock(&d->mu);
        if (eof) break;
    }
    return NULL;
}

static void *dp_writer(void *a)
{
    dp_t *d = a;
    for (int slot = 0;; slot ^= 1) {
        dp_out *o = &d->out[slot];
        pthread_mutex_lock(&d->mu);
        while (!o->full && !d->stop && !d->in_done) pthread_cond_wait(&d->cv, &d->mu);
        const int go = o->full;
        pthread_mutex_unlock(&d->mu);
        if (!go) break;                                         /* (in_done with nothing queued, or a failure) */
        const double t0 = now();
        if (crumble_bam_write_members(d->wr, o->buf, (size_t)o->len)) { fprintf(stderr, "crumble-hip: %s\n", crumble_bam_last_error()); dp_fail(d, 1); break; }
        pthread_mutex_lock(&d->mu);
        d->t_write += now() - t0;
        o->full = 0;
        pthread_cond_broadcast(&d->cv);
        pthread_mutex_unlock(&d->mu);
    }
    return NULL;
}

/* the complete contig runs [0, k) of the stream through the engines, a thread each: whoever is free takes the next run */
typedef struct {
    crumble_stream *s; crumble_hip *eng; int k; int *next; pthread_mutex_t *mu;
    const int *bed_lo, *bed_n; const int32_t *bs, *be;
    int *err, *err_group; char *err_text;
} dp_proc;
static void *dp_process(void *a)
{
    dp_proc *w = a;
    for (;;) {
        pthread_mutex_lock(w->mu);
        const int i = *w->err ? w->k : (*w->next)++;
        pthread_mutex_unlock(w->mu);
        if (i >= w->k) break;
        const int e = crumble_stream_process(w->s, w->eng, i, w->bed_n[i], w->bs + w->bed_lo[i], w->be + w->bed_lo[i]);
        if (e) {
            pthread_mutex_lock(w->mu);
            if (!*w->err) { *w->err = e; *w->err_group = i; snprintf(w->err_text, 256, "%s", crumble_hip_last_error()); }
            pthread_mutex_unlock(w->mu);
            break;
        }
    }
    return NULL;
}

static int device_pipeline(const char *in_fn, crumble_bam_reader *rd, crumble_bam_writer *wr, crumble_hip **engs, int n_eng, int device, int level,
                           int nbed, const int32_t *bt, const int32_t *bs, const int32_t *be, int verbose, int64_t *reads_out, int64_t *bases_out,
                           double times[4])
{
    int err = 0;
    const int32_t nref = crumble_bam_n_ref(rd);
    crumble_stream *s = crumble_stream_create(device, nref, crumble_bam_header_bytes(rd), &err);
    if (!s) { fprintf(stderr, "crumble-hip: device pipeline unavailable (%s)\n", crumble_hip_last_error()); return -1; }
    static dp_t d;
    memset(&d, 0, sizeof d);
    pthread_mutex_init(&d.mu, NULL); pthread_cond_init(&d.cv, NULL);
    d.wr = wr;
    if (!(d.fp = fopen(in_fn, "rb"))) { crumble_stream_destroy(s); return -1; }
    if (getenv("CRUMBLE_HIP_DP_CHUNK") && atoll(getenv("CRUMBLE_HIP_DP_CHUNK")) >= 4096) DP_CHUNK = (size_t)atoll(getenv("CRUMBLE_HIP_DP_CHUNK"));
    if (getenv("CRUMBLE_HIP_DP_PIECE") && atoll(getenv("CRUMBLE_HIP_DP_PIECE")) > 0) DP_PIECE = atoll(getenv("CRUMBLE_HIP_DP_PIECE"));
    size_t fsize = DP_CHUNK;
    if (!fseek(d.fp, 0, SEEK_END)) { const long fl = ftell(d.fp); if (fl >= 0) fsize = (size_t)fl; }
    rewind(d.fp);
    for (int i = 0; i < 2; i++) {
        d.in[i].cap = (fsize < DP_CHUNK ? fsize : DP_CHUNK) + ((size_t)1 << 16);
        d.in[i].bcap = d.in[i].cap / 4096 + 64;
        d.in[i].blk = malloc(d.in[i].bcap * sizeof(crumble_gz_block));
        /* (a buffer's worth of members, less for a small file -- page-locked memory costs ~0.3 ms per MB to get --, two members at least) */
        int64_t piece = DP_PIECE;
        if ((int64_t)fsize * 6 < piece) piece = ((int64_t)fsize * 6 / 65280 + 1) * 65280;
        d.out[i].cap = (size_t)crumble_gz_deflate_bound(piece > 2 * 65280 ? piece : 2 * 65280) + 64;
    }
    /* the first buffer of each kind before the clock starts (like the engines: set-up); the second ones when they are needed */
    /* (page-locked memory comes at ~0.3 ms per MB: the three buffers by a thread each, beside the device's allocations) */
    dp_pre pre[3] = { { &d.in[0].buf, d.in[0].cap, &d.in[0].pinned }, { &d.out[0].buf, d.out[0].cap, &d.out[0].pinned }, { &d.out[1].buf, d.out[1].cap, &d.out[1].pinned } };
    pthread_t pth[3];
    for (int i = 0; i < 3; i++) pthread_create(&pth[i], NULL, dp_prealloc, &pre[i]);
    (void)crumble_stream_reserve(s, (int64_t)d.in[0].cap, (int64_t)d.in[0].cap * 4);   /* (a hint: what it cannot get now it asks for again later) */
    for (int i = 0; i < 3; i++) pthread_join(pth[i], NULL);
    int rc = (!d.in[0].buf || !d.out[0].buf || !d.out[1].buf || !d.in[0].blk || !d.in[1].blk) ? 1 : 0;
    if (rc) fprintf(stderr, "crumble-hip: out of memory\n");
    int64_t ac0 = 0; double am0 = 0;
    crumble_hip_alloc_stats(&ac0, &am0);
    const double t_all = now();
    double t_gpu = 0;
    int64_t n_reads = 0, n_bases = 0;
    pthread_t rt, wt;
    int threads_up = 0;
    if (!rc) { pthread_create(&rt, NULL, dp_reader, &d); pthread_create(&wt, NULL, dp_writer, &d); threads_up = 1; }
    int bed_i = 0, oslot = 0, eof = 0;
    dp_out *inflight = NULL;                                  /* the output buffer of a drain that was begun and not yet collected */
    /* (CRUMBLE_HIP_ASYNC_DRAIN=0: the synchronous order -- 24 contigs 99-107 ms against 94-97) */
    const int async_drain = !(getenv("CRUMBLE_HIP_ASYNC_DRAIN") && !atoi(getenv("CRUMBLE_HIP_ASYNC_DRAIN")));
    for (int slot = 0; !rc && !eof; slot ^= 1) {
        dp_in *c = &d.in[slot];
        pthread_mutex_lock(&d.mu);
        while (!c->full && !d.stop) pthread_cond_wait(&d.cv, &d.mu);
        const int stop = d.stop;
        pthread_mutex_unlock(&d.mu);
        if (stop) break;
        double t0 = now();
        eof = c->eof;
        int e = 0;
        if (inflight) {                                         /* (a feed may move the buffer: the drain in flight first) */
            e = crumble_stream_drain_end(s, inflight->buf, (int64_t)inflight->cap, &inflight->len);
            if (e) { fprintf(stderr, "crumble-hip: %s (%s)\n", crumble_hip_strerror(e), crumble_hip_last_error()); rc = 1; break; }
            pthread_mutex_lock(&d.mu); inflight->full = 1; pthread_cond_broadcast(&d.cv); pthread_mutex_unlock(&d.mu);
            inflight = NULL;
        }
        e = crumble_stream_feed(s, c->buf, c->used, c->blk, c->nb, eof);
        pthread_mutex_lock(&d.mu); c->full = 0; pthread_cond_broadcast(&d.cv); pthread_mutex_unlock(&d.mu);
        if (e) {                                                /* (a file the device cannot take apart: the host reader has the better diagnostics) */
            if (e == CRUMBLE_ENOMEM || e == CRUMBLE_EHIP) { fprintf(stderr, "crumble-hip: %s (%s)\n", crumble_hip_strerror(e), crumble_hip_last_error()); rc = 1; }
            else { if (verbose) fprintf(stderr, "crumble-hip: device pipeline: %s\n", crumble_hip_last_error()); rc = -1; }
            break;
        }
        for (;;) {                                              /* whole contig runs until DP_PIECE bytes of records wait, then a drain */
            crumble_stream_group g[256];
            const int ng = crumble_stream_groups(s, g, 256);
            int k = 0;
            int64_t bytes = crumble_stream_pending(s);
            int bed_lo[256], bed_n[256];
            for (int i = 0; i < ng && i < 256 && g[i].complete && crumble_gz_deflate_bound(bytes) <= (int64_t)d.out[oslot].cap; i++) {
                while (bed_i < nbed && g[i].tid >= 0 && bt[bed_i] < g[i].tid) bed_i++;
                int j2 = bed_i;
                while (j2 < nbed && bt[j2] == g[i].tid) j2++;
                bed_lo[i] = bed_i; bed_n[i] = j2 - bed_i;
                n_reads += g[i].n_reads; bytes = g[i].byte_end; k++;
            }
            if (k) {
                int next = 0, perr = 0, pgrp = 0;
                char ptext[256] = "";
                pthread_mutex_t pmu = PTHREAD_MUTEX_INITIALIZER;
                dp_proc w[4]; pthread_t pt[4];
                const int nw = k < n_eng ? k : n_eng;
                for (int q = 0; q < nw; q++) {
                    w[q] = (dp_proc){ s, engs[q], k, &next, &pmu, bed_lo, bed_n, bs, be, &perr, &pgrp, ptext };
                    if (q) pthread_create(&pt[q], NULL, dp_process, &w[q]);
                }
                dp_process(&w[0]);
                for (int q = 1; q < nw; q++) pthread_join(pt[q], NULL);
                if (perr) {
                    if (strstr(ptext, "CG tag")) rc = -1;
                    else { fprintf(stderr, "crumble-hip: contig %s: %s (%s)\n", g[pgrp].tid >= 0 ? crumble_bam_ref_name(rd, g[pgrp].tid) : "*", crumble_hip_strerror(perr), ptext); rc = 2; }
                }
            }
            if (rc) break;
            if (k) {
                crumble_stream_groups(s, g, 256);               /* (n_bases of the runs just processed) */
                for (int i = 0; i < k; i++) n_bases += g[i].n_bases;
            }
            /* the last bytes of the file: the short last member too; otherwise whole members, a buffer's worth at a time, and
             * what is less than that waits for the next runs.  A drain is begun and left in flight: its members are made while
             * the next runs are processed (other streams), and collected before the next one starts */
            const int last = eof && k == ng && ng <= 256;
            if (!k && !(last && bytes)) break;
            if (inflight) {                                     /* the drain begun before these runs were processed */
                e = crumble_stream_drain_end(s, inflight->buf, (int64_t)inflight->cap, &inflight->len);
                if (e) { fprintf(stderr, "crumble-hip: %s (%s)\n", crumble_hip_strerror(e), crumble_hip_last_error()); rc = 1; break; }
                pthread_mutex_lock(&d.mu); inflight->full = 1; pthread_cond_broadcast(&d.cv); pthread_mutex_unlock(&d.mu);
                inflight = NULL;
            }
            int first = 1;
            while (first || (last ? crumble_stream_pending(s) > 0 : crumble_gz_deflate_bound(crumble_stream_pending(s)) > (int64_t)d.out[oslot].cap)) {
                dp_out *o = &d.out[oslot];
                pthread_mutex_lock(&d.mu);
                t_gpu += now() - t0;
                while (o->full && !d.stop) pthread_cond_wait(&d.cv, &d.mu);
                const int stop2 = d.stop;
                pthread_mutex_unlock(&d.mu);
                t0 = now();
                if (stop2) { rc = rc ? rc : 1; break; }
                const int whole = !last && crumble_gz_deflate_bound(bytes) <= (int64_t)o->cap && first;   /* less than a buffer waits: only mark the runs */
                int64_t taken = 0;
                e = crumble_stream_drain_begin(s, k, level, last, whole ? 0 : (int64_t)o->cap, &taken);
                k = 0; first = 0;
                if (e) { fprintf(stderr, "crumble-hip: %s (%s)\n", crumble_hip_strerror(e), crumble_hip_last_error()); rc = 1; break; }
                if (!taken) continue;
                oslot ^= 1;
                const int more = last ? crumble_stream_pending(s) > 0 : crumble_gz_deflate_bound(crumble_stream_pending(s)) > (int64_t)d.out[oslot].cap;
                if (!more && async_drain && !last) { inflight = o; break; }      /* (collected after the next runs, or in front of the next feed) */
                e = crumble_stream_drain_end(s, o->buf, (int64_t)o->cap, &o->len);
                if (e) { fprintf(stderr, "crumble-hip: %s (%s)\n", crumble_hip_strerror(e), crumble_hip_last_error()); rc = 1; break; }
                pthread_mutex_lock(&d.mu); o->full = 1; pthread_cond_broadcast(&d.cv); pthread_mutex_unlock(&d.mu);
            }
            if (rc) break;
        }
        t_gpu += now() - t0;
    }
    if (inflight && !rc) {                                    /* (cannot be: the file's last drain is not left in flight) */
        const int e = crumble_stream_drain_end(s, inflight->buf, (int64_t)inflight->cap, &inflight->len);
        if (e) { fprintf(stderr, "crumble-hip: %s (%s)\n", crumble_hip_strerror(e), crumble_hip_last_error()); rc = 1; }
        else { pthread_mutex_lock(&d.mu); inflight->full = 1; pthread_cond_broadcast(&d.cv); pthread_mutex_unlock(&d.mu); }
    }
    pthread_mutex_lock(&d.mu);
    if (rc) d.stop = 1;
    d.in_done = 1;
    pthread_cond_broadcast(&d.cv);
    pthread_mutex_unlock(&d.mu);
    if (threads_up) {
        pthread_join(wt, NULL);                                 /* (writes what is queued, unless something failed) */
        pthread_mutex_lock(&d.mu); d.stop = 1; pthread_cond_broadcast(&d.cv); pthread_mutex_unlock(&d.mu);
        pthread_join(rt, NULL);
    }
    if (!rc && d.fail) rc = d.fail;
    times[0] = d.t_read; times[1] = t_gpu; times[2] = d.t_write; times[3] = now() - t_all;
    if (!rc && verbose) {
        float ms[5] = {0}; int64_t bi = 0, bo = 0;
        crumble_stream_times(s, ms, &bi, &bo);
        int64_t ac1 = 0; double am1 = 0;
        crumble_hip_alloc_stats(&ac1, &am1);
        fprintf(stderr, "BGZF codec: inflate device, deflate device\n");
        fprintf(stderr, "pipeline: device (records inflated, delimited, run and deflated in HBM; %.1f MB in, %.1f MB out over PCIe; GPU stage: inflate %.3f s, "
                "delimit %.3f, engine %.3f, deflate %.3f, move %.3f; device allocations inside: %lld, %.3f s)\n", bi / 1e6, bo / 1e6, ms[0] / 1e3, ms[1] / 1e3, ms[2] / 1e3, ms[3] / 1e3, ms[4] / 1e3,
                (long long)(ac1 - ac0), (am1 - am0) / 1e3);
    }
    *reads_out = n_reads; *bases_out = n_bases;
    fclose(d.fp);
    for (int i = 0; i < 2; i++) { dp_free(d.in[i].buf, d.in[i].cap, d.in[i].pinned); dp_free(d.out[i].buf, d.out[i].cap, d.out[i].pinned); free(d.in[i].blk); }
    crumble_stream_destroy(s);
    return rc;
}

static void *gpu_worker(void *a_)
{
    worker_arg *a = a_; pipe_t *p = a->p; const int d = a->d;
    for (;;) {
        pthread_mutex_lock(&p->mu);
        while (!p->q_head[d] && !p->closing) pthread_cond_wait(&p->cv, &p->mu);
        job *j = p->q_head[d];
        if (!j) { pthread_mutex_unlock(&p->mu); break; }
        p->q_head[d] = j->next; if (!p->q_head[d]) p->q_tail[d] = NULL;
        pthread_mutex_unlock(&p->mu);
        const double t0 = now();
        const size_t nb = (size_t)j->B.seq_off[j->B.n_reads];
        /* upload + run, then the result in coded form (2 bits per base + the escape bytes: a quarter of the bytes over
         * PCIe; the writer expands it into the records, which still hold the input qualities).  A batch with many escapes
         * (long reads with P-block) comes back as plain bytes. */
        crumble_hip *eng = p->eng[d];
        uint8_t *qp = NULL;
        const size_t qpcap = (size_t)CRUMBLE_QUALSP_SIZE(nb, 4) + 64;
        /* binned qualities (four or eight values) cross PCIe as codes: crumble_batch.qualsp (include/crumble_hip.h).  The
         * cheap look (which values occur) comes first: a contig with more than sixteen values allocates nothing. */
        if (p->coded_quals && nb > 0 && j->B.quals && crumble_pack_quals((int64_t)nb, j->B.quals, NULL, j->B.qual_lut) &&
            (qp = crumble_bam_big_alloc(qpcap))) {
            const int bits = crumble_pack_quals((int64_t)nb, j->B.quals, qp, j->B.qual_lut);
            if (bits) { j->B.qualsp = qp; j->B.qual_bits = bits; }
        }
        j->err = crumble_hip_contig_run(eng, &j->B);
        j->B.qualsp = NULL; j->B.qual_bits = 0;
        if (qp) crumble_bam_big_free(qp, qpcap);
        if (j->err || (j->err = crumble_hip_encode(eng, &j->coded)))
            snprintf(j->errtxt, sizeof j->errtxt, "%s (%s)", crumble_hip_strerror(j->err), crumble_hip_last_error());
        else {
            crumble_coded *c = &j->coded;
            j->is_coded = !p->plain_download && (size_t)c->n_escapes <= nb / 8;
            const size_t o1 = ((size_t)c->n_words * 4 + 63) & ~(size_t)63, o2 = o1 + (((size_t)c->n_blocks * 8 + 63) & ~(size_t)63);
            j->qcap = (j->is_coded ? o2 + (size_t)c->n_escapes : nb) + 64;
            j->qout = crumble_bam_big_alloc(j->qcap);
            if (!j->qout) { j->err = CRUMBLE_ENOMEM; snprintf(j->errtxt, sizeof j->errtxt, "out of memory"); }
            else {
                if (j->is_coded) {
                    c->codes = (uint32_t *)j->qout; c->esc_off = (int64_t *)(j->qout + o1); c->esc = j->qout + o2;
                    j->err = crumble_hip_download_coded(eng, c, NULL, NULL);            /* stats stay on the device */
                } else j->err = crumble_hip_download(eng, j->qout, NULL, NULL);
                if (j->err) snprintf(j->errtxt, sizeof j->errtxt, "%s (%s)", crumble_hip_strerror(j->err), crumble_hip_last_error());
            }
        }
        if (!j->err && p->want_suspect && j->B.tid >= 0) {        /* -b: snp_score.c:1496,1676,1768,1802,1810 */
            const int64_t ns = crumble_hip_suspect(p->eng[d], NULL, NULL, 0);
            j->sus_pos = malloc((size_t)(ns > 0 ? ns : 1) * 4); j->sus_why = malloc((size_t)(ns > 0 ? ns : 1));
            if (ns < 0 || !j->sus_pos || !j->sus_why || crumble_hip_suspect(p->eng[d], j->sus_pos, j->sus_why, ns) != ns) {
                j->err = ns < 0 ? (int)ns : CRUMBLE_ENOMEM; snprintf(j->errtxt, sizeof j->errtxt, "-b failed");
            }
            j->n_sus = ns > 0 ? ns : 0;
        }
        pthread_mutex_lock(&p->mu);
        p->t_gpu[d] += now() - t0;
        p->q_len[d]--;
        j->done = 1; j->next = p->done_head; p->done_head = j;
        pthread_cond_broadcast(&p->cv);
        pthread_mutex_unlock(&p->mu);
    }
    /* the one collective of the path: every engine's accumulated statistics vector, summed over the devices */
    if (p->use_rccl) {
        const int e = crumble_hip_stats_allreduce(p->eng[d], p->comm[d]);
        if (e) { pthread_mutex_lock(&p->mu); p->failed = 1; fprintf(stderr, "crumble-hip: stats all-reduce on device %d: %s (%s)\n", p->devices[d], crumble_hip_strerror(e), crumble_hip_last_error()); pthread_mutex_unlock(&p->mu); }
    }
    return NULL;
}

/* -b lines of one contig (snp_score.c:1496,1676,1768,1802,1810): name, pos-50 (>= 0), pos+50, reason */
static void print_suspects(FILE *fp, const char *ref, const int32_t *pos, const uint8_t *bits, int64_t n)
{
    static const struct { int bit; const char *name; } why[] = {
        { CRUMBLE_WHY_VDEEP, "VDEEP" }, { CRUMBLE_WHY_DEEP, "DEEP" }, { CRUMBLE_WHY_CLIP, "CLIP" },
        { CRUMBLE_WHY_INDEL_LEN, "INDEL_LEN" }, { CRUMBLE_WHY_INDEL_COVERAGE, "INDEL_COVERAGE" } };
    for (int64_t k = 0; k < n; k++)
        for (int y = 0; y < 5; y++)
            if (bits[k] & why[y].bit)
                fprintf(fp, "%s\t%d\t%d\t%s\n", ref, pos[k] > 50 ? pos[k] - 50 : 0, pos[k] + 50, why[y].name);
}

static void *writer_thread(void *p_)
{
    pipe_t *p = p_;
    for (;;) {
        pthread_mutex_lock(&p->mu);
        job *j = NULL;
        for (;;) {
            job **pp = &p->done_head;
            while (*pp && (*pp)->seq != p->next_write) pp = &(*pp)->next;
            if (*pp) { j = *pp; *pp = j->next; break; }
            int pending = 0;
            for (int d = 0; d < p->ndev; d++) pending += p->q_len[d];
            if (p->closing && !pending && !p->done_head) break;
            pthread_cond_wait(&p->cv, &p->mu);
        }
        pthread_mutex_unlock(&p->mu);
        if (!j) break;
        const double t0 = now();
        if (j->err && !p->failed) {
            fprintf(stderr, "crumble-hip: contig %s: %s\n", crumble_bam_ref_name(p->rd, j->B.tid), j->errtxt);
            p->failed = 2;
        }
        if (!p->failed) {
            if (j->n_sus) print_suspects(p->bed_fp, crumble_bam_ref_name(p->rd, j->B.tid), j->sus_pos, j->sus_why, j->n_sus);
            if (j->is_coded ? crumble_bam_write_detached_coded(p->wr, j->g, &j->coded) : crumble_bam_write_detached(p->wr, j->g, j->qout)) {
                fprintf(stderr, "crumble-hip: %s\n", crumble_bam_last_error()); p->failed = 1;
            }
        }
        if (p->verbose > 1) fprintf(stderr, "  %s: %lld reads on device %d\n", crumble_bam_ref_name(p->rd, j->B.tid), (long long)j->B.n_reads, p->devices[j->dev]);
        crumble_bam_group_free(j->g); crumble_bam_big_free(j->qout, j->qcap); free(j->sus_pos); free(j->sus_why);
        pthread_mutex_lock(&p->mu);
        p->t_write += now() - t0;
        p->next_write++;
        free(j);
        pthread_cond_broadcast(&p->cv);
        pthread_mutex_unlock(&p->mu);
    }
    return NULL;
}

/* ---- one contig over all devices (-w): windows of about equal bases (include/crumble_hip.h "windows",
 * crumble_amd/windows.py is the same plan and schedule in Python).  Window k owns the reads that start in
 * [w_k, w_k+1); its batch is the slice of the contig's reads that overlap the columns it decides. */
typedef struct {
    int64_t lo, hi;                 /* slice of the contig's reads */
    crumble_window w;
    crumble_batch vb;               /* view into the contig's arrays, offsets rebased */
    int64_t *seq_off, *cig_off;
    uint8_t *q; size_t qcap;        /* the slice's output qualities */
    int32_t *sus_pos; uint8_t *sus_why; int64_t n_sus;
    int err; char errtxt[200];
} win_t;
typedef struct { pipe_t *p; int d, phase; win_t *w; } win_task;

static void win_fail(win_t *w, int e)
{
    w->err = e;
    snprintf(w->errtxt, sizeof w->errtxt, "%s (%s)", crumble_hip_strerror(e), crumble_hip_last_error());
}

static void *win_phase(void *a_)
{
    win_task *t = a_; pipe_t *p = t->p; win_t *w = t->w; crumble_hip *eng = p->eng[t->d];
    int e = 0;
    if (w->err) return NULL;
    if (t->phase == 0) {
        if ((e = crumble_hip_upload(eng, &w->vb)) || (e = crumble_hip_set_window(eng, &w->w)) || (e = crumble_hip_run_columns(eng))) win_fail(w, e);
    } else if (t->phase == 1) {
        if ((e = crumble_hip_run_triggers(eng))) win_fail(w, e);
    } else {
        if ((e = crumble_hip_run_finish(eng)) || (e = crumble_hip_download(eng, w->q, NULL, NULL))) { win_fail(w, e); return NULL; }
        if (p->want_suspect) {                                   /* -b: the columns this window counts */
            const int64_t ns = crumble_hip_suspect(eng, NULL, NULL, 0);
            int32_t *sp = malloc((size_t)(ns > 0 ? ns : 1) * 4); uint8_t *sw = malloc((size_t)(ns > 0 ? ns : 1));
            if (ns < 0 || !sp || !sw || crumble_hip_suspect(eng, sp, sw, ns) != ns) { w->err = CRUMBLE_ENOMEM; snprintf(w->errtxt, sizeof w->errtxt, "-b failed"); free(sp); free(sw); return NULL; }
            int64_t k2 = 0;
            for (int64_t k = 0; k < ns; k++) if (sp[k] >= w->w.col_start && sp[k] < w->w.stat_end) { sp[k2] = sp[k]; sw[k2] = sw[k]; k2++; }
            w->sus_pos = sp; w->sus_why = sw; w->n_sus = k2;
        }
    }
    return NULL;
}

static void win_run_phase(pipe_t *p, win_t *wins, int nw, int phase)
{
    pthread_t th[MAX_DEV]; win_task tk[MAX_DEV];
    for (int k = 0; k < nw; k++) { tk[k].p = p; tk[k].d = k; tk[k].phase = phase; tk[k].w = &wins[k]; pthread_create(&th[k], NULL, win_phase, &tk[k]); }
    for (int k = 0; k < nw; k++) pthread_join(th[k], NULL);
}

/* the whole contig of job j through up to p->ndev windows; fills j->qout */
static void run_windowed(pipe_t *p, job *j)
{
    const crumble_batch *B = &j->B;
    const int64_t n = B->n_reads, nb = n ? B->seq_off[n] : 0;
    win_t wins[MAX_DEV];
    int nw = 0;
    memset(wins, 0, sizeof wins);
    j->qcap = (size_t)nb + 64; j->qout = crumble_bam_big_alloc(j->qcap);
    int64_t *end = malloc((size_t)(n > 0 ? n : 1) * 8), *pmax = malloc((size_t)(n > 0 ? n : 1) * 8);
    if (!j->qout || !end || !pmax) { j->err = CRUMBLE_ENOMEM; snprintf(j->errtxt, sizeof j->errtxt, "out of memory"); goto done; }
    if (B->tid < 0 || n < 4 * p->ndev) {                        /* the unmapped tail / a tiny contig: one device */
        j->err = crumble_hip_contig(p->eng[0], &j->B, j->qout, NULL, NULL);
        if (j->err) snprintf(j->errtxt, sizeof j->errtxt, "%s (%s)", crumble_hip_strerror(j->err), crumble_hip_last_error());
        goto done;
    }
    for (int64_t i = 0; i < n; i++) {                           /* pos + reference length (M D N = X) */
        int64_t rl = 0;
        for (int64_t c = B->cig_off[i]; c < B->cig_off[i + 1]; c++) {
            const unsigned op = B->cigar[c] & 15;
            if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rl += B->cigar[c] >> 4;
        }
        end[i] = (int64_t)B->pos[i] + rl;
        pmax[i] = i && pmax[i - 1] > end[i] ? pmax[i - 1] : end[i];
    }
    int64_t firsts[MAX_DEV + 1];
    int nf = 0;
    firsts[nf++] = 0;
    for (int k = 1; k < p->ndev; k++) {
        const int64_t target = nb / p->ndev * k;
        int64_t lo = 0, hi = n;                                 /* first read whose bases start at or after the target */
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (B->seq_off[mid] < target) lo = mid + 1; else hi = mid; }
        int64_t r = lo < n ? lo : n - 1;
        while (r > 0 && B->pos[r - 1] == B->pos[r]) r--;        /* all reads of one start position go together */
        if (r > firsts[nf - 1]) firsts[nf++] = r;
    }
    firsts[nf] = n;
    nw = nf;
    for (int k = 0; k < nw; k++) {
        win_t *w = &wins[k];
        const int64_t r0 = firsts[k], r1 = firsts[k + 1];
        const int64_t w0 = B->pos[r0];
        int64_t ek = w0;
        for (int64_t i = r0; i < r1; i++) if (end[i] > ek) ek = end[i];
        int64_t stat_end = r1 < n ? (int64_t)B->pos[r1] : (ek > pmax[n - 1] ? ek : pmax[n - 1]);
        if (stat_end > ek) ek = stat_end;                       /* decide and count through to the next start: a read of an earlier
                                                                 * window may cover columns behind the owned reads' end */
        int64_t lo = 0;
        if (k) { int64_t a = 0, b = r0; while (a < b) { const int64_t mid = (a + b) >> 1; if (pmax[mid] <= w0) a = mid + 1; else b = mid; } lo = a; }
        int64_t hi; { int64_t a = r1, b = n; while (a < b) { const int64_t mid = (a + b) >> 1; if (B->pos[mid] < ek) a = mid + 1; else b = mid; } hi = a; }
        if (ek > INT32_MAX || stat_end > INT32_MAX) { j->err = CRUMBLE_EINVAL; snprintf(j->errtxt, sizeof j->errtxt, "coordinate out of range"); goto done; }
        w->lo = lo; w->hi = hi;
        w->w.col_start = (int32_t)w0; w->w.col_end = (int32_t)ek; w->w.stat_end = (int32_t)stat_end;
        w->w.read_first = r0 - lo; w->w.read_end = r1 - lo;
        const int64_t m = hi - lo, b0 = B->seq_off[lo], c0 = B->cig_off[lo];
        w->seq_off = malloc((size_t)(m + 1) * 8); w->cig_off = malloc((size_t)(m + 1) * 8);
        w->qcap = (size_t)(B->seq_off[hi] - b0) + 64; w->q = crumble_bam_big_alloc(w->qcap);
        if (!w->seq_off || !w->cig_off || !w->q) { j->err = CRUMBLE_ENOMEM; snprintf(j->errtxt, sizeof j->errtxt, "out of memory"); goto done; }
        for (int64_t i = 0; i <= m; i++) { w->seq_off[i] = B->seq_off[lo + i] - b0; w->cig_off[i] = B->cig_off[lo + i] - c0; }
        w->vb = *B;
        w->vb.n_reads = m; w->vb.pos = B->pos + lo; w->vb.mapq = B->mapq + lo; w->vb.flag = B->flag + lo;
        w->vb.seq_off = w->seq_off; w->vb.cig_off = w->cig_off; w->vb.cigar = B->cigar + c0;
        w->vb.bases = B->bases + b0; w->vb.quals = B->quals + b0;
        if (B->l_qseq && B->n_cigar) { w->vb.l_qseq = B->l_qseq + lo; w->vb.n_cigar = B->n_cigar + lo; w->vb.n_bases = B->seq_off[hi] - b0; w->vb.n_cigar_ops = B->cig_off[hi] - c0; }
    }
    {
        crumble_carry carry, nxt;
        crumble_carry_init(&carry);
        win_run_phase(p, wins, nw, 0);                          /* upload + columns: all windows at once */
        for (int k = 0; k < nw && !wins[k].err; k++) {          /* the running depth average, window to window */
            crumble_carry_init(&nxt);
            const int e = crumble_hip_run_depth(p->eng[k], &carry, &nxt);
            if (e) { win_fail(&wins[k], e); break; }
            carry.total_col = nxt.total_col; carry.total_depth = nxt.total_depth;
        }
        win_run_phase(p, wins, nw, 1);                          /* over-depth test, trigger lists, STR search */
        for (int k = 0; k < nw && !wins[k].err; k++) {          /* the STR window state, window to window */
            crumble_carry_init(&nxt);
            const int e = crumble_hip_run_fold(p->eng[k], &carry, &nxt);
            if (e) { win_fail(&wins[k], e); break; }
            carry.min_pos = nxt.min_pos; carry.max_pos = nxt.max_pos; carry.min_pos2 = nxt.min_pos2; carry.max_pos2 = nxt.max_pos2;
        }
        win_run_phase(p, wins, nw, 2);                          /* replay, P-block, statistics, download */
    }
    {
        int64_t ns = 0;
        for (int k = 0; k < nw; k++) {
            win_t *w = &wins[k];
            if (w->err) { if (!j->err) { j->err = w->err; snprintf(j->errtxt, sizeof j->errtxt, "window %d: %s", k, w->errtxt); } continue; }
            const int64_t r0 = w->w.read_first, r1 = w->w.read_end;
            memcpy(j->qout + B->seq_off[w->lo + r0], w->q + w->seq_off[r0], (size_t)(w->seq_off[r1] - w->seq_off[r0]));
            ns += w->n_sus;
        }
        if (!j->err && p->want_suspect) {
            j->sus_pos = malloc((size_t)(ns > 0 ? ns : 1) * 4); j->sus_why = malloc((size_t)(ns > 0 ? ns : 1));
            if (!j->sus_pos || !j->sus_why) { j->err = CRUMBLE_ENOMEM; snprintf(j->errtxt, sizeof j->errtxt, "out of memory"); }
            else for (int k = 0; k < nw; k++) {
                memcpy(j->sus_pos + j->n_sus, wins[k].sus_pos, (size_t)wins[k].n_sus * 4);
                memcpy(j->sus_why + j->n_sus, wins[k].sus_why, (size_t)wins[k].n_sus);
                j->n_sus += wins[k].n_sus;
            }
        }
    }
done:
    for (int k = 0; k < MAX_DEV; k++) { free(wins[k].seq_off); free(wins[k].cig_off); crumble_bam_big_free(wins[k].q, wins[k].qcap); free(wins[k].sus_pos); free(wins[k].sus_why); }
    free(end); free(pmax);
}

static void *allreduce_thread(void *a_)
{
    win_task *t = a_; pipe_t *p = t->p;
    const int e = crumble_hip_stats_allreduce(p->eng[t->d], p->comm[t->d]);
    if (e) { pthread_mutex_lock(&p->mu); p->failed = 1; fprintf(stderr, "crumble-hip: stats all-reduce on device %d: %s (%s)\n", p->devices[t->d], crumble_hip_strerror(e), crumble_hip_last_error()); pthread_mutex_unlock(&p->mu); }
    return NULL;
}

/* lpt_pack (crumble_amd/shard.py): longest first onto the least loaded device */
static void lpt_assign(const crumble_bam_reader *rd, int ndev, int *dev_of)
{
    const int32_t n = crumble_bam_n_ref(rd);
    int32_t *order = malloc((size_t)(n > 0 ? n : 1) * sizeof *order);
    int64_t load[MAX_DEV] = {0};
    for (int32_t i = 0; i < n; i++) order[i] = i;
    for (int32_t i = 1; i < n; i++) {                      /* insertion sort by (-length, index): n is a few thousand at most */
        const int32_t v = order[i]; int32_t k = i;
        while (k > 0 && crumble_bam_ref_len(rd, order[k - 1]) < crumble_bam_ref_len(rd, v)) { order[k] = order[k - 1]; k--; }
        order[k] = v;
    }
    for (int32_t i = 0; i < n; i++) {
        int best = 0;
        for (int d = 1; d < ndev; d++) if (load[d] < load[best]) best = d;
        dev_of[order[i]] = best; load[best] += crumble_bam_ref_len(rd, order[i]);
    }
    free(order);
}

int main(int argc, char **argv)
{
    crumble_params P;
    crumble_params_default(&P);
    int opt, verbose = 0, threads = 0, no_pg = 0, ngpu = 1, windowed = 0;
    const char *bed_fn = NULL, *wl = NULL, *bl = NULL, *in_fmt = NULL, *out_fmt = NULL;
    int bd[3] = {0, 0, 0}, bi[3] = {0, 0, 0};
    FILE *bed_fp = NULL;
    while ((opt = getopt(argc, argv, "I:O:q:d:x:Q:D:X:m:l:u:U:c:i:L:Bs:t:T:hr:b:vC:M:Z:P:V:p:e:f:g:E:F:G:S135789zR:Y:y:k:K:N@:j:w")) != -1) {
        switch (opt) {
        case 'q': P.min_qual_A = atoi(optarg); break;
        case 'd': P.min_indel_A = atoi(optarg); break;
        case 'x': P.min_discrep_A = atof(optarg); break;
        case 'Q': P.min_qual_B = atoi(optarg); break;
        case 'D': P.min_indel_B = atoi(optarg); break;
        case 'X': P.min_discrep_B = atof(optarg); break;
        case 'm': P.min_mqual = atoi(optarg); break;
        case 'l': P.qlow = atoi(optarg); break;
        case 'u': P.qhigh = atoi(optarg); break;
        case 'c': P.qcutoff = atoi(optarg); break;
        case 'U': P.qcap = atoi(optarg); break;
        case 'i': P.iSTR_mul = atof(optarg); if (strchr(optarg, ',')) P.iSTR_add = atoi(strchr(optarg, ',') + 1); break;
        case 's': P.sSTR_mul = atof(optarg); if (strchr(optarg, ',')) P.sSTR_add = atoi(strchr(optarg, ',') + 1); break;
        case 'L': P.reduce_qual = atoi(optarg); break;
        case 'B': P.binary_qual = 1; break;
        case 'R': bed_fn = optarg; break;
        case 'C': P.clip_perc = atof(optarg); break;
        case 'M': P.low_mqual_perc = atof(optarg); break;
        case 'Z': P.ins_len_perc = atof(optarg); break;
        case 'P': P.over_depth = atof(optarg); break;
        case 'Y': P.indel_fract = atof(optarg); break;
        case 'V': P.indel_ov_perc = atof(optarg); break;
        case 'p': P.pblock = atoi(optarg); break;
        case 'y': crumble_params_tech(&P, optarg); break;
        case 'k': crumble_params_keep_quals(&P, optarg, 0); break;
        case 'K': crumble_params_keep_quals(&P, optarg, 1); break;
        case 'N': P.perfect_col = 1; break;
        case 'S': P.softclip = 1; break;
        case '1': case '3': case '5': case '7': case '8': case '9': crumble_params_preset(&P, opt - '0'); break;
        case 'z': no_pg = 1; break;
        case 'v': verbose++; break;
        case '@': threads = atoi(optarg); break;
        case 'j': ngpu = atoi(optarg); break;
        case 'w': windowed = 1; break;
        case 'h': usage(stdout); return 0;
        case 't': wl = optarg; break;
        case 'T': bl = optarg; break;
        case 'e': bd[0] = atoi(optarg) + 33; break;
        case 'f': bd[1] = atoi(optarg) + 33; break;
        case 'g': bd[2] = atoi(optarg) + 33; break;
        case 'E': bi[0] = atoi(optarg) + 33; break;
        case 'F': bi[1] = atoi(optarg) + 33; break;
        case 'G': bi[2] = atoi(optarg) + 33; break;
        case 'b': if (!(bed_fp = fopen(optarg, "w"))) { perror(optarg); return 1; } break;
        case 'I': in_fmt = optarg; break;                       /* hts_parse_format strings, snp_score.c:2201-2207 */
        case 'O': out_fmt = optarg; break;
        case 'r':
            fprintf(stderr, "crumble-hip: option -r (index region iterator) is host "
                            "plumbing that this build does not provide\n");
            return 1;
        default: usage(stderr); return 1;
        }
    }
    if (argc - optind != 2) { usage(stderr); return 1; }
    if (crumble_hip_abi_version() != CRUMBLE_HIP_ABI || crumble_hip_struct_size(0) != sizeof(crumble_batch) ||
        crumble_hip_struct_size(1) != sizeof(crumble_params) || crumble_hip_struct_size(2) != sizeof(crumble_coded)) {
        fprintf(stderr, "crumble-hip: libcrumble_hip.so was built from another version of crumble_hip.h (ABI %d, this driver %d)\n",
                crumble_hip_abi_version(), CRUMBLE_HIP_ABI);
        return 1;
    }
    const char *in_fn = argv[optind], *out_fn = argv[optind + 1];
    const int level = getenv("CRUMBLE_BAM_LEVEL") ? atoi(getenv("CRUMBLE_BAM_LEVEL")) : -1;
    /* -@ also bounds the engine library's host helpers (crumble_pack_quals / crumble_decode_quals run per GPU worker:
     * each gets its share of the budget, not the whole of it beside the reader's and the writer's teams) */
    if (threads > 0) crumble_set_host_threads(threads / (ngpu > 0 ? ngpu : 1) > 0 ? threads / (ngpu > 0 ? ngpu : 1) : 1);

    {   /* SAM / CRAM / explicit formats: the htslib backend, one contig at a time on one device */
        unsigned char magic[4] = {0, 0, 0, 0};
        FILE *mf = fopen(in_fn, "rb");
        if (mf) { if (fread(magic, 1, 4, mf) != 4) magic[0] = 0; fclose(mf); }
        const size_t ol = strlen(out_fn);
        const int bgzf = magic[0] == 0x1f && magic[1] == 0x8b && magic[2] == 8 && (magic[3] & 4);
        const int out_other = (ol > 5 && !strcasecmp(out_fn + ol - 5, ".cram")) || (ol > 4 && !strcasecmp(out_fn + ol - 4, ".sam"));
        if (in_fmt || out_fmt || !bgzf || out_other) {
            size_t len = 1;
            for (int i = 0; i < argc; i++) len += strlen(argv[i]) + 1;
            char *cl = malloc(len), *q = cl;
            for (int i = 0; i < argc; i++) q += sprintf(q, "%s%s", i ? " " : "", argv[i]);
            int herr = 0;
            if (ngpu > 1 || windowed)
                fprintf(stderr, "crumble-hip: -j / -w apply to BAM in and out only; SAM / CRAM / -I / -O run one contig at a time on one device\n");
            crumble_hts *hs = crumble_hts_open(in_fn, in_fmt, out_fn, out_fmt, no_pg ? NULL : "crumble-hip", "0.2", cl, threads, &herr);
            if (!hs) { fprintf(stderr, "crumble-hip: %s\n", crumble_bam_last_error()); return 1; }
            if (crumble_hts_set_aux_filter(hs, wl, bl, bd, bi)) { fprintf(stderr, "crumble-hip: %s\n", crumble_bam_last_error()); return 1; }
            int32_t *bt = NULL, *bs = NULL, *be = NULL; int nbed = 0, bed_i = 0, e2 = 0;
            if (bed_fn && (nbed = crumble_hts_load_bed(hs, bed_fn, &bt, &bs, &be)) < 0) { fprintf(stderr, "crumble-hip: %s\n", crumble_bam_last_error()); return 1; }
            const int device = getenv("CRUMBLE_HIP_DEVICE") ? atoi(getenv("CRUMBLE_HIP_DEVICE")) : 0;
            crumble_hip *eng = crumble_hip_create(&P, device, &e2);
            if (!eng) { fprintf(stderr, "crumble-hip: %s (%s)\n", crumble_hip_strerror(e2), crumble_hip_last_error()); return 2; }
            int64_t stats[CRUMBLE_ST_N] = {0};
            uint8_t *qout = NULL; size_t qcap = 0;
            for (;;) {
                crumble_batch B;
                const int rc = crumble_hts_next_contig(hs, &B);
                if (rc < 0) { fprintf(stderr, "crumble-hip: %s\n", crumble_bam_last_error()); return 1; }
                if (rc == 0) break;
                while (bed_i < nbed && bt[bed_i] < B.tid) bed_i++;
                int j2 = bed_i;
                while (j2 < nbed && bt[j2] == B.tid) j2++;
                if (B.tid >= 0 && j2 > bed_i) { B.n_bed = j2 - bed_i; B.bed_start = bs + bed_i; B.bed_end = be + bed_i; }
                const size_t nb = (size_t)B.seq_off[B.n_reads];
                if (qcap < nb + 16) { crumble_bam_big_free(qout, qcap); qcap = nb + nb / 8 + 64; qout = crumble_bam_big_alloc(qcap); if (!qout) { fprintf(stderr, "crumble-hip: out of memory\n"); return 1; } }
                const int e = crumble_hip_contig(eng, &B, qout, NULL, stats);
                if (e) { fprintf(stderr, "crumble-hip: contig %s: %s (%s)\n", crumble_hts_ref_name(hs, B.tid), crumble_hip_strerror(e), crumble_hip_last_error()); return 2; }
                if (bed_fp && B.tid >= 0) {                              /* -b: the same lines as the BAM path */
                    const int64_t ns = crumble_hip_suspect(eng, NULL, NULL, 0);
                    int32_t *sp = malloc((size_t)(ns > 0 ? ns : 1) * 4); uint8_t *sw = malloc((size_t)(ns > 0 ? ns : 1));
                    if (ns < 0 || !sp || !sw || crumble_hip_suspect(eng, sp, sw, ns) != ns) { fprintf(stderr, "crumble-hip: -b failed\n"); return 2; }
                    print_suspects(bed_fp, crumble_hts_ref_name(hs, B.tid), sp, sw, ns);
                    free(sp); free(sw);
                }
                if (crumble_hts_write_group(hs, qout)) { fprintf(stderr, "crumble-hip: %s\n", crumble_bam_last_error()); return 1; }
            }
            if (crumble_hts_close(hs)) { fprintf(stderr, "crumble-hip: %s\n", crumble_bam_last_error()); return 1; }
            if (verbose) for (int i = 0; i < (int)(sizeof ST_NAME / sizeof *ST_NAME); i++) fprintf(stderr, "%-16s %lld\n", ST_NAME[i], (long long)stats[i]);
            crumble_hip_destroy(eng); crumble_bam_big_free(qout, qcap); free(bt); free(bs); free(be); free(cl);
            if (bed_fp && fclose(bed_fp)) { perror("-b"); return 1; }
            return 0;
        }
    }

    static pipe_t p;
    memset(&p, 0, sizeof p);
    pthread_mutex_init(&p.mu, NULL); pthread_cond_init(&p.cv, NULL);
    p.verbose = verbose; p.bed_fp = bed_fp; p.want_suspect = bed_fp != NULL;
    p.plain_download = getenv("CRUMBLE_HIP_PLAIN_DOWNLOAD") && atoi(getenv("CRUMBLE_HIP_PLAIN_DOWNLOAD"));
    p.coded_quals = getenv("CRUMBLE_HIP_CODED_QUALS") && atoi(getenv("CRUMBLE_HIP_CODED_QUALS"));
    if (ngpu < 1 || ngpu > MAX_DEV) { fprintf(stderr, "crumble-hip: -j must be 1..%d\n", MAX_DEV); return 1; }
    /* One device: several engines on it (CRUMBLE_HIP_ENGINES=n, default 2), a worker thread each -- while one contig's bytes
     * cross PCIe another one's kernels run, and a file of thousands of small contigs (hs38DH's decoys) pays their fixed cost
     * n at a time.  The engines' statistics are added up on the host (as for a device named twice in CRUMBLE_HIP_DEVICES);
     * with -j N > 1 every device has one engine and RCCL sums. */
    /* CRUMBLE_HIP_PIPELINE = device | host | auto: the device pipeline (device_pipeline() above) for BAM -> BAM on one
     * device without tag filters and without -b */
    int dev_mode = 0;
    {
        const char *pm = getenv("CRUMBLE_HIP_PIPELINE");
        if (!pm) pm = "auto";
        if (strcmp(pm, "auto") && strcmp(pm, "device") && strcmp(pm, "host")) { fprintf(stderr, "crumble-hip: CRUMBLE_HIP_PIPELINE must be auto, device or host\n"); return 1; }
        const int can = ngpu == 1 && !windowed && !getenv("CRUMBLE_HIP_DEVICES") && !wl && !bl && !bed_fp &&
                        !bd[0] && !bd[1] && !bd[2] && !bi[0] && !bi[1] && !bi[2] && !p.plain_download && !p.coded_quals;
        if (!strcmp(pm, "device") && !can) fprintf(stderr, "crumble-hip: CRUMBLE_HIP_PIPELINE=device needs one device and none of -w -t -T -b -e..-G: the host pipeline runs\n");
        /* auto: the device pipeline whenever it applies (measured on MI355X boxes, profiles/NOTES_r06.md: 4.7-5.0 Gbases/s file to
         * file against the host pipeline's 1.8-2.3), unless the caller asks for the slow levels 7-9, which search harder than
         * the device's one level does */
        dev_mode = can && (!strcmp(pm, "device") || (!strcmp(pm, "auto") && PIPELINE_AUTO_DEVICE && level < 7));
    }
    /* The device pipeline runs a dozen streams (four engines, their field streams, the codec's); the runtime spreads streams
     * over GPU_MAX_HW_QUEUES hardware queues, four by default, and with four the engines' kernels and the deflate's persistent
     * workgroups get in each other's way on some boxes and not on others (measured, 24 contigs, a box each: 0.088 s and
     * 0.118-0.182 s of loop with four queues, 0.092-0.096 s on all of them with two; six and more: slower everywhere).  Two
     * unless the caller says otherwise; it must be in the environment before the runtime starts. */
    if (dev_mode && !getenv("GPU_MAX_HW_QUEUES")) setenv("GPU_MAX_HW_QUEUES", "2", 0);
    int engines_one = 0;
    if (ngpu == 1 && !windowed && !getenv("CRUMBLE_HIP_DEVICES")) {      /* (the device pipeline too: its contig runs go to whichever engine is free) */
        const char *ev = getenv("CRUMBLE_HIP_ENGINES");
        int n = ev ? atoi(ev) : dev_mode ? 4 : 2;               /* (measured through the device pipeline with 1 / 2 / 3 / 4: 24 contigs 107 / 103 / 99 / 100 ms, 1500 contigs of 5 kb 775 / 494 / 397 / 344 ms) */
        if (n < 1) n = 1;
        if (n > (dev_mode ? 4 : 8)) n = dev_mode ? 4 : 8;
        if (n > 1) { engines_one = 1; ngpu = n; }
    }
    p.ndev = ngpu;
    {   /* device list: CRUMBLE_HIP_DEVICES="a,b,.." (tests may name one device twice), else CRUMBLE_HIP_DEVICE + k */
        const char *dl = getenv("CRUMBLE_HIP_DEVICES");
        const int base = getenv("CRUMBLE_HIP_DEVICE") ? atoi(getenv("CRUMBLE_HIP_DEVICE")) : 0;
        for (int d = 0; d < ngpu; d++) {
            p.devices[d] = engines_one ? base : base + d;
            if (dl && *dl) { p.devices[d] = atoi(dl); dl = strchr(dl, ','); dl = dl ? dl + 1 : ""; }
        }
    }
    int err = 0, distinct = 1;
    for (int d = 0; d < ngpu; d++) {
        for (int k = 0; k < d; k++) if (p.devices[k] == p.devices[d]) distinct = 0;
        p.eng[d] = crumble_hip_create(&P, p.devices[d], &err);
        if (!p.eng[d]) { fprintf(stderr, "crumble-hip: device %d: %s (%s)\n", p.devices[d], crumble_hip_strerror(err), crumble_hip_last_error()); return 2; }
    }
    if (ngpu > 1 && distinct) {
        const int e = crumble_hip_comm_init_all(ngpu, p.devices, p.comm);
        if (e) { fprintf(stderr, "crumble-hip: RCCL communicator: %s (%s)\n", crumble_hip_strerror(e), crumble_hip_last_error()); return 2; }
        p.use_rccl = 1;
    }
    crumble_bam_reader *rd = crumble_bam_open(in_fn, threads, &err);
    if (!rd) { fprintf(stderr, "crumble-hip: %s\n", crumble_bam_last_error()); return 1; }
    p.rd = rd;
    /* the bases go to the device as BAM packs them (the window planner of -w slices one-code-per-byte arrays) */
    if (!windowed) crumble_bam_set_packed_bases(rd, 1);

    int32_t *bt = NULL, *bs = NULL, *be = NULL; int nbed = 0;
    if (bed_fn && (nbed = crumble_bam_load_bed(rd, bed_fn, &bt, &bs, &be)) < 0) { fprintf(stderr, "crumble-hip: %s\n", crumble_bam_last_error()); return 1; }

    char *pg = NULL;
    if (!no_pg) {
        size_t len = 64;
        for (int i = 0; i < argc; i++) len += strlen(argv[i]) + 1;
        pg = malloc(len);
        char *q = pg + sprintf(pg, "@PG\tID:crumble-hip\tPN:crumble-hip\tVN:0.2\tCL:");
        for (int i = 0; i < argc; i++) q += sprintf(q, "%s%s", i ? " " : "", argv[i]);
    }
    crumble_bam_writer *wr = crumble_bam_create(out_fn, rd, pg, level, threads, &err);
    if (!wr) { fprintf(stderr, "crumble-hip: %s\n", crumble_bam_last_error()); return 1; }
    if (crumble_bam_set_aux_filter(wr, wl, bl, bd, bi)) { fprintf(stderr, "crumble-hip: %s\n", crumble_bam_last_error()); return 1; }
    p.wr = wr;
    /* BGZF members inflated / deflated on the device: CRUMBLE_HIP_CODEC = auto | gpu | gpu-inflate | gpu-deflate | host.
     * auto (the default) goes by what was measured on MI355X boxes: the device deflates a gigabyte of
     * records in ~0.1 s whatever the level (copies included), sixteen host threads need 0.15-0.17 s at level 1 and twice that
     * at the default level, so the device deflates unless the caller asked for level 0 (stored) or for level 7 and up (the
     * host's slow levels search harder than the device's one); the host's threads inflate (a gigabyte in 0.08-0.1 s on
     * sixteen of them, less than the device needs with the bytes' way back over PCIe) unless there are fewer than eight. */
    crumble_gz *gz_in = NULL, *gz_out = NULL;
    double dev_times[4] = {0, 0, 0, 0};
    int64_t dev_reads = 0, dev_bases = 0;
    int dev_done = 0;
    const double t_all = now();
    if (dev_mode) {
        const int rcd = device_pipeline(in_fn, rd, wr, p.eng, ngpu, p.devices[0], level, nbed, bt, bs, be, verbose, &dev_reads, &dev_bases, dev_times);
        if (rcd > 0) return rcd;
        if (rcd == 0) dev_done = 1;
        else {                                                  /* not for this file: from the top, the host way */
            if (verbose) fprintf(stderr, "pipeline: the device pipeline does not take this file; the host pipeline runs\n");
            crumble_bam_writer_close(wr);
            if (!(wr = crumble_bam_create(out_fn, rd, pg, level, threads, &err))) { fprintf(stderr, "crumble-hip: %s\n", crumble_bam_last_error()); return 1; }
            p.wr = wr;
            for (int d2 = 0; d2 < ngpu; d2++) (void)crumble_hip_stats_reset(p.eng[d2]);
        }
    }
    if (!dev_done) {
        const char *cm = getenv("CRUMBLE_HIP_CODEC");
        if (!cm) cm = "auto";
        if (strcmp(cm, "auto") && strcmp(cm, "gpu") && strcmp(cm, "gpu-inflate") && strcmp(cm, "gpu-deflate") && strcmp(cm, "host")) {
            fprintf(stderr, "crumble-hip: CRUMBLE_HIP_CODEC must be auto, gpu, gpu-inflate, gpu-deflate or host\n"); return 1;
        }
        const int ht = threads > 0 ? threads : crumble_bam_default_threads();
        int want_in = !strcmp(cm, "gpu") || !strcmp(cm, "gpu-inflate"), want_out = !strcmp(cm, "gpu") || !strcmp(cm, "gpu-deflate");
        if (!strcmp(cm, "auto")) {
            want_in = ht < 8;
            want_out = level != 0 && level < 7;
        }
        int ge = 0;
        if (want_in && !(gz_in = crumble_gz_create(p.devices[0], &ge)))
            fprintf(stderr, "crumble-hip: device inflate unavailable (%s): host threads inflate\n", crumble_hip_last_error());
        if (want_out && !(gz_out = crumble_gz_create(p.devices[ngpu - 1], &ge)))
            fprintf(stderr, "crumble-hip: device deflate unavailable (%s): host threads deflate\n", crumble_hip_last_error());
        crumble_bam_coder cin = { gz_in, gz_inflate_cb, NULL, NULL }, cout = { gz_out, NULL, gz_deflate_cb, crumble_gz_deflate_bound };
        if (gz_in) crumble_bam_reader_set_codec(rd, &cin);
        if (gz_out) crumble_bam_writer_set_codec(wr, &cout);
        if (verbose) fprintf(stderr, "BGZF codec: inflate %s, deflate %s\n", gz_in ? "device" : crumble_bam_codec(), gz_out ? "device" : crumble_bam_codec());
    }

    double t_read = 0;
    int bed_i = 0, rc_main = 0;
    int64_t seq = 0, total_reads = 0, total_bases = 0;
    const int32_t nref = crumble_bam_n_ref(rd);
    int *dev_of = calloc((size_t)(nref > 0 ? nref : 1), sizeof *dev_of);
    lpt_assign(rd, ngpu, dev_of);

    pthread_t wt, gt[MAX_DEV]; worker_arg wa[MAX_DEV];
    if (dev_done) goto finish;
    pthread_create(&wt, NULL, writer_thread, &p);
    if (!windowed)
        for (int d = 0; d < ngpu; d++) { wa[d].p = &p; wa[d].d = d; pthread_create(&gt[d], NULL, gpu_worker, &wa[d]); }

    for (;;) {
        crumble_batch B;
        double t0 = now();
        const int rc = crumble_bam_next_contig(rd, &B);
        t_read += now() - t0;
        if (rc < 0) { fprintf(stderr, "crumble-hip: %s\n", crumble_bam_last_error()); rc_main = 1; break; }
        if (rc == 0) break;
        while (bed_i < nbed && bt[bed_i] < B.tid) bed_i++;      /* groups arrive in header order, tid -1 last */
        int j2 = bed_i;
        while (j2 < nbed && bt[j2] == B.tid) j2++;
        if (B.tid >= 0 && j2 > bed_i) { B.n_bed = j2 - bed_i; B.bed_start = bs + bed_i; B.bed_end = be + bed_i; }
        job *j = calloc(1, sizeof *j);
        t0 = now();
        if (j) j->g = crumble_bam_detach_group(rd);
        t_read += now() - t0;
        if (!j || !j->g) { fprintf(stderr, "crumble-hip: %s\n", crumble_bam_last_error()); rc_main = 1; free(j); break; }
        j->B = B; j->seq = seq++;
        j->dev = (B.tid >= 0 && B.tid < nref) ? dev_of[B.tid] : 0;
        total_reads += B.n_reads; total_bases += B.seq_off[B.n_reads];
        if (windowed) {                                         /* -w: this contig over all devices, then to the writer */
            t0 = now();
            run_windowed(&p, j);
            pthread_mutex_lock(&p.mu);
            p.t_gpu[0] += now() - t0;
            j->done = 1; j->next = p.done_head; p.done_head = j;
            const int failed = p.failed;
            pthread_cond_broadcast(&p.cv);
            while (seq - p.next_write > MAX_INFLIGHT && !p.failed) pthread_cond_wait(&p.cv, &p.mu);   /* bounds the host memory */
            pthread_mutex_unlock(&p.mu);
            if (failed) break;
            continue;
        }
        pthread_mutex_lock(&p.mu);
        /* bounds the host memory: a detached contig holds its records and output bytes until the WRITER has
         * written it, so the bound counts contigs read but not yet written, not only the device's queue */
        while ((p.q_len[j->dev] >= MAX_INFLIGHT || j->seq - p.next_write >= (int64_t)p.ndev * MAX_INFLIGHT) && !p.failed)
            pthread_cond_wait(&p.cv, &p.mu);
        if (p.q_tail[j->dev]) p.q_tail[j->dev]->next = j; else p.q_head[j->dev] = j;
        p.q_tail[j->dev] = j; p.q_len[j->dev]++;
        const int failed = p.failed;
        pthread_cond_broadcast(&p.cv);
        pthread_mutex_unlock(&p.mu);
        if (failed) break;
    }
    pthread_mutex_lock(&p.mu); p.closing = 1; pthread_cond_broadcast(&p.cv); pthread_mutex_unlock(&p.mu);
    if (!windowed) for (int d = 0; d < ngpu; d++) pthread_join(gt[d], NULL);
    else if (p.use_rccl) {
        win_task tk[MAX_DEV];
        for (int d = 0; d < ngpu; d++) { tk[d].p = &p; tk[d].d = d; pthread_create(&gt[d], NULL, allreduce_thread, &tk[d]); }
        for (int d = 0; d < ngpu; d++) pthread_join(gt[d], NULL);
    }
    pthread_join(wt, NULL);
    if (p.failed) rc_main = p.failed == 2 ? 2 : 1;
finish:
    if (dev_done) { t_read = dev_times[0]; p.t_gpu[0] = dev_times[1]; p.t_write = dev_times[2]; total_reads = dev_reads; total_bases = dev_bases; }
    if (bed_fp && fclose(bed_fp)) { perror("-b"); rc_main = 1; }
    if (crumble_bam_writer_close(wr)) { fprintf(stderr, "crumble-hip: %s\n", crumble_bam_last_error()); rc_main = 1; }
    /* statistics: after the all-reduce every engine holds the sum; without RCCL (one device, or one device named
     * twice) the host adds the engines' vectors */
    for (int d = 0; d < (p.use_rccl ? 1 : ngpu); d++) {
        int64_t s1[CRUMBLE_ST_N];
        if (crumble_hip_stats_read(p.eng[d], s1) == 0) for (int i = 0; i < CRUMBLE_ST_N; i++) p.stats[i] += s1[i];
    }
    const double wall = dev_done ? dev_times[3] : now() - t_all;
    if (verbose && !rc_main) {
        for (int i = 0; i < (int)(sizeof ST_NAME / sizeof *ST_NAME); i++) fprintf(stderr, "%-16s %lld\n", ST_NAME[i], (long long)p.stats[i]);
        const double gb = p.stats[CRUMBLE_ST_BASES] / 1e9;
        double tg = 0;
        for (int d = 0; d < ngpu; d++) tg += p.t_gpu[d];
        fprintf(stderr, "end-to-end %.3f s (read+parse %.3f, upload+kernels+download %.3f, write %.3f): %.4f Gbases/s file to file\n",
                wall, t_read, tg, p.t_write, wall > 0 ? gb / wall : 0.0);
        if (engines_one) fprintf(stderr, "1 device, %d engines on it, ", ngpu);
        if (dev_done) fprintf(stderr, "1 device, records resident; %lld reads, %lld bases\n", (long long)total_reads, (long long)total_bases);
        else fprintf(stderr, "%d device(s)%s, reader / GPU workers / writer overlapped; %lld reads, %lld bases\n", engines_one ? 1 : ngpu,
                p.use_rccl ? ", statistics summed by RCCL all-reduce" : "", (long long)total_reads, (long long)total_bases);
        char tb[512];
        crumble_bam_timing(tb, sizeof tb);
        fprintf(stderr, "%s\n", tb);
    }
    crumble_bam_close(rd);
    crumble_gz_destroy(gz_in); crumble_gz_destroy(gz_out);
    for (int d = 0; d < ngpu; d++) { if (p.use_rccl) crumble_hip_comm_destroy(p.comm[d]); crumble_hip_destroy(p.eng[d]); }
    free(bt); free(bs); free(be); free(pg); free(dev_of);
    return rc_main;
}
