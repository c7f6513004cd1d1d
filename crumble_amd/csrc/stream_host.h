/*
 * stream_host.h -- crumble_stream: a BAM file's records on the device from inflate to deflate (include/crumble_hip.h).
 * Included by engine.hip behind gz_host.h (one translation unit: it reaches into crumble_hip and crumble_gz).
 *
 *   feed     whole BGZF members (host bytes) -> inflated behind what the device buffer holds (crumble_gz, CRC-32 checked),
 *            record starts found (stream_kernels.h: guess / walk / verify), contig runs found
 *   process  one contig's records -> an engine (k_st_fields, the engine's own expansion and run, k_st_scatter): the
 *            engine's input never exists on the host, its output qualities go straight back into the records
 *   drain    the records of the first k contigs deflated into BGZF members (crumble_gz), host bytes out; the rest moves
 *            to the front of the buffer
 * What crosses PCIe is the compressed file, once in each direction, plus a few words per contig.
 */
#ifndef CRUMBLE_STREAM_HOST_H
#define CRUMBLE_STREAM_HOST_H

#include "stream_kernels.h"
#include <mutex>
#define ST_SLOTS 4

struct crumble_stream {
    int device = 0;
    crumble_gz *gz = nullptr;                    /* codec context (its stream carries this object's kernels too) */
    int n_ref = 0;
    DevBuf d_rec[2];                             /* the live bytes: [base, base + len) of d_rec[cur]; offsets below count from base */
    int cur = 0;
    int64_t base = 0;                            /* (advanced by drain; back to 0 when a feed moves the live bytes to the other buffer) */
    int64_t sync_base = 0;                       /* base when the records were last delimited: d_recoff and the groups' bytes count from there */
    int64_t len = 0;                             /* live bytes */
    int64_t hdr = 0;                             /* bytes of the BAM header still to be dropped at the front (first feeds) */
    int64_t start = 0;                           /* first record not yet processed: what lies in front of it was processed and waits for a drain */
    int64_t tail = 0;                            /* where the incomplete last record starts (== len when there is none) */
    int64_t n_rec = 0;
    DevBuf d_guess, d_exit, d_count, d_slots, d_coff, d_recoff, d_list, d_misc, d_err, d_pred;
    /* per engine that processes runs of this stream (up to ST_SLOTS at once, a host thread each): the per-read fields, the
     * totals, a stream for them */
    struct Slot { crumble_hip *owner = nullptr; hipStream_t st = nullptr; DevBuf d_pos, d_mapq, d_flag, d_lq, d_ncg, d_noq, d_tot; float ms = 0; };
    Slot slot[ST_SLOTS];
    std::mutex slot_mu;
    std::vector<crumble_stream_group> groups;    /* in record order; the last one may go on in the next feed */
    bool eof = false;
    double t_begin = 0;
    bool inflight = false;                       /* a drain was begun and not ended: its members are being made */
    void *pinned = nullptr;
    int64_t bytes_in = 0, bytes_out = 0;         /* compressed bytes fed / drained so far */
    float ms_inflate = 0, ms_sync = 0, ms_engine = 0, ms_deflate = 0, ms_move = 0;   /* host clock around each part */
};

extern "C" crumble_stream *crumble_stream_create(int device, int32_t n_ref, int64_t header_bytes, int *err)
{
    int dummy; if (!err) err = &dummy;
    *err = CRUMBLE_OK;
    crumble_gz *gz = crumble_gz_create(device, err);
    if (!gz) return nullptr;
    crumble_stream *s = new (std::nothrow) crumble_stream();
    if (!s) { crumble_gz_destroy(gz); *err = CRUMBLE_ENOMEM; return nullptr; }
    s->device = device; s->gz = gz; s->n_ref = n_ref; s->hdr = header_bytes > 0 ? header_bytes : 0;
    bool ok = hipHostMalloc(&s->pinned, 4096, 0) == hipSuccess;
    uint32_t ppm[128];
    for (int q = 0; q < 128; q++) ppm[q] = (uint32_t)(1e6 * pow(10.0, -q / 10.0) + 0.5);
    ok = ok && s->d_err.reserve(sizeof ppm) == CRUMBLE_OK && copy_sync(gz->stream, s->d_err.p, ppm, sizeof ppm, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && s->d_misc.reserve(256) == CRUMBLE_OK;
    if (!ok) { g_last_error = "crumble_stream: device set-up failed"; *err = CRUMBLE_EHIP; crumble_stream_destroy(s); return nullptr; }
    return s;
}

extern "C" void crumble_stream_destroy(crumble_stream *s)
{
    if (!s) return;
    dev_free_deferred(0);
    (void)hipSetDevice(s->device);
    if (s->gz && s->gz->stream) (void)hipStreamSynchronize(s->gz->stream);
    DevBuf *all[] = { &s->d_rec[0], &s->d_rec[1], &s->d_guess, &s->d_exit, &s->d_count, &s->d_slots, &s->d_coff, &s->d_recoff, &s->d_list,
                      &s->d_misc, &s->d_err, &s->d_pred };
    for (DevBuf *b : all) b->release();
    for (auto &sl : s->slot) {
        if (sl.st) { (void)hipStreamSynchronize(sl.st); (void)hipStreamDestroy(sl.st); }
        DevBuf *sb[] = { &sl.d_pos, &sl.d_mapq, &sl.d_flag, &sl.d_lq, &sl.d_ncg, &sl.d_noq, &sl.d_tot };
        for (DevBuf *b : sb) b->release();
    }
    if (s->pinned) (void)hipHostFree(s->pinned);
    crumble_gz_destroy(s->gz);
    delete s;
}

/* Set-up before the first feed (optional): device memory for feeds of up to comp_bytes of members that inflate to about
 * raw_bytes -- the record buffers, the codec's token and slot arrays -- so that no feed / drain stops to allocate. */
extern "C" int crumble_stream_reserve(crumble_stream *s, int64_t comp_bytes, int64_t raw_bytes)
{
    if (!s || comp_bytes < 0 || raw_bytes < 0) return CRUMBLE_EINVAL;
    HIPCHK(hipSetDevice(s->device));
    crumble_gz *g = s->gz;
    int e;
    const size_t raw = (size_t)raw_bytes + 4096, nmem = (size_t)(raw_bytes / 32768 + comp_bytes / 16384 + 64);
    const size_t in_round = raw < (size_t)8 * GZ_BATCH_RAW ? raw : (size_t)8 * GZ_BATCH_RAW;
    const size_t out_round = raw < (size_t)GZ_BATCH_RAW ? raw : (size_t)GZ_BATCH_RAW, nb = out_round / GZ_MAXBLK + 1;
    const size_t tok = (in_round + 64) * 4 > nb * 65536 * 4 ? (in_round + 64) * 4 : nb * 65536 * 4;
    if ((e = s->d_rec[0].reserve(raw)) || (e = s->d_rec[1].reserve(raw)) || (e = g->d_comp.reserve((size_t)comp_bytes + GZ_IN_PAD)) ||
        (e = g->d_blk.reserve(nmem * sizeof(GzBlock))) || (e = g->d_tokoff.reserve((nmem + 1) * 8)) || (e = g->d_tok.reserve(tok)) ||
        (e = g->d_ntok.reserve(nmem * 4)) || (e = g->d_status.reserve(nmem * 4)) ||
        (e = g->d_cand.reserve(nb * 65536 * 4)) || (e = g->d_slots.reserve(nb * GZ_SLOT)) || (e = g->d_sizes.reserve(nb * 4)) ||
        (e = g->d_off.reserve((nb + 1) * 8)) || (e = g->d_out.reserve(nb * GZ_SLOT)))
        return e;
    const size_t nch = raw / ST_CHUNK + 2;
    if ((e = s->d_guess.reserve(nch * 8)) || (e = s->d_exit.reserve(nch * 8)) || (e = s->d_count.reserve(nch * 4)) ||
        (e = s->d_slots.reserve(nch * ST_MAXREC * 8)) || (e = s->d_coff.reserve((nch + 1) * 8)) || (e = s->d_recoff.reserve(raw / 36 * 8 / 4 + 64)) ||
        (e = s->d_list.reserve((size_t)(1 << 20) * 16)))
        return e;
    return CRUMBLE_OK;
}

/* record starts and contig runs of d_rec[cur][start, len) */
static double st_now() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + 1e-6 * t.tv_nsec; }
struct StClock { float *acc; double t0; StClock(float *a) : acc(a), t0(st_now()) {} ~StClock() { *acc += (float)(st_now() - t0); } };

static int stream_sync(crumble_stream *s)
{
    StClock clk(&s->ms_sync);
    hipStream_t st = s->gz->stream;
    const uint8_t *R = s->d_rec[s->cur].as<uint8_t>() + s->base;
    s->sync_base = s->base;
    s->groups.clear(); s->n_rec = 0; s->tail = s->start;
    if (s->len <= s->start) return CRUMBLE_OK;
    const int nch = (int)((s->len - s->start + ST_CHUNK - 1) / ST_CHUNK);
    int e;
    if ((e = s->d_guess.reserve((size_t)nch * 8)) || (e = s->d_exit.reserve((size_t)nch * 8)) || (e = s->d_count.reserve((size_t)nch * 4)) ||
        (e = s->d_slots.reserve((size_t)nch * ST_MAXREC * 8)) || (e = s->d_coff.reserve((size_t)(nch + 1) * 8)) || (e = s->d_pred.reserve((size_t)nch * 4)))
        return e;
    int *bad = s->d_misc.as<int>();
    int64_t *tailp = (int64_t *)(s->d_misc.as<char>() + 16);
    int *nlist = (int *)(s->d_misc.as<char>() + 32);
    HIPCHK(hipMemsetAsync(s->d_misc.p, 0, 64, st));
    CRUMBLE_LAUNCH(k_st_guess, nch, 64, st, R, s->start, s->len, s->n_ref, nch, s->d_guess.as<int64_t>());
    int *changed = (int *)(s->d_misc.as<char>() + 48);
    for (int round = 0;; round++) {                         /* (one round when every guess was right: the usual case) */
        CRUMBLE_LAUNCH(k_st_walk, (nch + 63) / 64, 64, st, R, s->start, s->len, nch, s->d_guess.as<int64_t>(), s->d_slots.as<int64_t>(),
                       s->d_count.as<int32_t>(), s->d_exit.as<int64_t>());
        HIPCHK(hipMemsetAsync(changed, 0, 4, st));
        CRUMBLE_LAUNCH(k_st_check, (nch + 255) / 256, 256, st, nch, s->start, s->d_guess.as<int64_t>(), s->d_exit.as<int64_t>(), s->d_count.as<int32_t>(),
                       s->d_coff.as<int64_t>(), s->d_pred.as<int32_t>(), bad);             /* (d_coff: free until k_st_offsets below) */
        CRUMBLE_LAUNCH(k_st_fix, (nch + 255) / 256, 256, st, nch, s->d_guess.as<int64_t>(), s->d_coff.as<int64_t>(), s->d_pred.as<int32_t>(), changed);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync((char *)s->pinned + 128, changed, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        int hchanged;
        memcpy(&hchanged, (char *)s->pinned + 128, 4);
        if (!hchanged) break;
        if (round > nch || round > 4096) { g_last_error = "crumble_stream: the record chain did not settle"; return CRUMBLE_EINVAL; }
    }
    CRUMBLE_LAUNCH(k_st_tail, 1, 1, st, nch, s->d_guess.as<int64_t>(), s->d_exit.as<int64_t>(), s->len, tailp, bad);
    CRUMBLE_LAUNCH(k_st_offsets, 1, 1024, st, nch, s->d_count.as<int32_t>(), s->d_coff.as<int64_t>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(s->pinned, s->d_misc.p, 32, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync((char *)s->pinned + 64, s->d_coff.as<int64_t>() + nch, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    int hbad; int64_t htail, nrec;
    memcpy(&hbad, s->pinned, 4); memcpy(&htail, (char *)s->pinned + 16, 8); memcpy(&nrec, (char *)s->pinned + 64, 8);
    if (hbad) {
        char buf[160];
        snprintf(buf, sizeof buf, "crumble_stream: the BAM records could not be delimited (check %d): not a BAM record stream, or corrupt", hbad);
        g_last_error = buf;
        return CRUMBLE_EINVAL;
    }
    s->tail = htail; s->n_rec = nrec;
    if (!nrec) return CRUMBLE_OK;
    const int max_list = 1 << 20;
    if ((e = s->d_recoff.reserve((size_t)(nrec + 1) * 8)) || (e = s->d_list.reserve((size_t)max_list * 16))) return e;
    CRUMBLE_LAUNCH(k_st_compact, nch, 256, st, nch, s->d_slots.as<int64_t>(), s->d_count.as<int32_t>(), s->d_coff.as<int64_t>(), s->d_recoff.as<int64_t>());
    CRUMBLE_LAUNCH(k_st_groups, (unsigned)((nrec + 255) / 256), 256, st, R, s->len, s->n_ref, nrec, s->d_recoff.as<int64_t>(), s->d_list.as<int64_t>(),
                   nlist, max_list, bad);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(s->pinned, s->d_misc.p, 48, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    int nl;
    memcpy(&hbad, s->pinned, 4); memcpy(&nl, (char *)s->pinned + 32, 4);
    if (hbad || nl > max_list) { g_last_error = "crumble_stream: implausible BAM record (or more than a million contig runs in one feed)"; return CRUMBLE_EINVAL; }
    std::vector<int64_t> list((size_t)nl * 2);
    HIPCHK(copy_sync(st, list.data(), s->d_list.p, (size_t)nl * 16, hipMemcpyDeviceToHost));
    std::vector<std::pair<int64_t, int64_t>> runs((size_t)nl);
    for (int i = 0; i < nl; i++) runs[(size_t)i] = { list[2 * (size_t)i], list[2 * (size_t)i + 1] };
    std::sort(runs.begin(), runs.end());
    /* the byte range of a run: from its first record to the next run's first record (the last one: to the tail) */
    std::vector<int64_t> first((size_t)nl + 1);
    {
        std::vector<int64_t> idx((size_t)nl);
        for (int i = 0; i < nl; i++) idx[(size_t)i] = runs[(size_t)i].first;
        /* rec_off of the runs' first records: gathered one by one (a handful per feed) */
        for (int i = 0; i < nl; i++) HIPCHK(hipMemcpyAsync(&first[(size_t)i], s->d_recoff.as<int64_t>() + idx[(size_t)i], 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        first[(size_t)nl] = s->tail;
    }
    for (int i = 0; i < nl; i++) {
        crumble_stream_group g;
        memset(&g, 0, sizeof g);
        g.tid = (int32_t)runs[(size_t)i].second;
        g.first_read = runs[(size_t)i].first;
        g.n_reads = (i + 1 < nl ? runs[(size_t)i + 1].first : nrec) - g.first_read;
        g.byte_start = first[(size_t)i]; g.byte_end = first[(size_t)i + 1];
        g.complete = (i + 1 < nl || s->eof) ? 1 : 0;
        if (i && g.tid >= 0 && runs[(size_t)i - 1].second >= 0 && g.tid < runs[(size_t)i - 1].second) { g_last_error = "crumble_stream: contigs out of header order"; return CRUMBLE_EUNSORTED; }
        s->groups.push_back(g);
    }
    return CRUMBLE_OK;
}

/* More of the file: whole BGZF members (crumble_bgzf_scan's table; uoff is ignored -- the members' bytes go behind what the
 * buffer holds, in order).  eof: nothing follows (the last contig run is then complete, and an incomplete record an error). */
extern "C" int crumble_stream_feed(crumble_stream *s, const uint8_t *comp, int64_t comp_len, const crumble_gz_block *blk, int64_t n_blocks, int eof)
{
    if (!s || comp_len < 0 || n_blocks < 0 || (n_blocks && (!comp || !blk))) return CRUMBLE_EINVAL;
    if (s->inflight) { g_last_error = "crumble_stream_feed: a drain is in flight (crumble_stream_drain_end first)"; return CRUMBLE_EINVAL; }
    HIPCHK(hipSetDevice(s->device));
    hipStream_t st = s->gz->stream;
    int64_t add = 0;
    for (int64_t i = 0; i < n_blocks; i++) add += blk[i].ulen;
    DevBuf &B = s->d_rec[s->cur];
    const size_t need = (size_t)(s->len + add) + 4096;
    int e;
    if ((size_t)s->base + need > B.cap) {                   /* no room behind the live bytes: they move to the front of the other buffer */
        DevBuf &N = s->d_rec[s->cur ^ 1];
        size_t want = B.cap > N.cap ? B.cap : N.cap;
        if (!want) want = (size_t)1 << 30;
        while (want < need) want += want / 2;
        if (N.cap < want) {
            if (N.cap >= need) want = N.cap;
            else if ((e = N.reserve(want))) return e;
        }
        if (s->len) CRUMBLE_LAUNCH(k_st_copy, 1024, 256, st, B.as<const uint8_t>() + s->base, N.as<uint8_t>(), s->len);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(st));
        s->cur ^= 1; s->base = 0;
    }
    uint8_t *R = s->d_rec[s->cur].as<uint8_t>() + s->base;
    std::vector<crumble_gz_block> b2((size_t)n_blocks);
    int64_t u = s->len;
    for (int64_t i = 0; i < n_blocks; i++) { b2[(size_t)i] = blk[i]; b2[(size_t)i].uoff = u; u += blk[i].ulen; }
    { StClock clk(&s->ms_inflate); if (n_blocks && (e = gz_inflate_impl(s->gz, comp, comp_len, b2.data(), n_blocks, nullptr, R, u, 1))) return e; }
    s->len = u; s->eof = eof != 0; s->bytes_in += comp_len;
    if (s->hdr && s->len >= s->hdr) { s->base += s->hdr; s->len -= s->hdr; s->hdr = 0; R = s->d_rec[s->cur].as<uint8_t>() + s->base; }   /* the header goes */
    HIPCHK(hipMemsetAsync(R + s->len, 0, 4096, st));
    if (s->hdr) { if (s->eof) { g_last_error = "crumble_stream: the file ends inside the BAM header"; return CRUMBLE_EINVAL; } return CRUMBLE_OK; }
    if ((e = stream_sync(s))) return e;
    if (s->eof && s->tail != s->len) { g_last_error = "crumble_stream: truncated BAM record at the end of the file"; return CRUMBLE_EINVAL; }
    return CRUMBLE_OK;
}

/* the contig runs the buffer holds, in order (valid until the next feed / drain); returns their number */
extern "C" int crumble_stream_groups(crumble_stream *s, crumble_stream_group *out, int max)
{
    if (!s) return CRUMBLE_EINVAL;
    const int n = (int)s->groups.size();
    const int64_t shift = s->base - s->sync_base;           /* (a drain since the records were delimited moved the origin) */
    for (int i = 0; i < n && i < max; i++) { out[i] = s->groups[(size_t)i]; out[i].byte_start -= shift; out[i].byte_end -= shift; }
    return n;
}

/* One contig run through an engine of the same device: the engine's batch is made from the records on the device
 * (crumble_hip_upload's work without a host array), run, and the output qualities written back into the records.
 * bed_*: this contig's keep regions (host, after crumble_bed_collapse), n_bed 0 for none.  The engine's statistics
 * accumulate as with crumble_hip_contig().  CRUMBLE_EUNSUPP-like cases (a CG-tag CIGAR) return CRUMBLE_EINVAL with a
 * message: the caller falls back to the host path for the file. */
extern "C" int crumble_stream_process(crumble_stream *s, crumble_hip *h, int group, int32_t n_bed, const int32_t *bed_start, const int32_t *bed_end)
{
    if (!s || !h || group < 0 || group >= (int)s->groups.size() || h->device != s->device) return CRUMBLE_EINVAL;
    const crumble_stream_group &g = s->groups[(size_t)group];
    if (!g.complete) { g_last_error = "crumble_stream_process: the contig run is not complete yet"; return CRUMBLE_EINVAL; }
    HIPCHK(hipSetDevice(s->device));
    crumble_stream::Slot *sl = nullptr;
    {   /* this engine's slot (several engines, a host thread each, may process different runs of one stream at once) */
        std::lock_guard<std::mutex> lk(s->slot_mu);
        for (auto &c : s->slot) if (c.owner == h) { sl = &c; break; }
        if (!sl) for (auto &c : s->slot) if (!c.owner) { c.owner = h; sl = &c; break; }
        if (!sl) { g_last_error = "crumble_stream_process: more than ST_SLOTS engines on one stream"; return CRUMBLE_EINVAL; }
        if (!sl->st && hipStreamCreateWithFlags(&sl->st, hipStreamNonBlocking) != hipSuccess) { sl->st = nullptr; g_last_error = "hipStreamCreate failed"; return CRUMBLE_EHIP; }
    }
    hipStream_t st = sl->st;
    const int64_t n = g.n_reads;
    int e;
    StClock clk(&sl->ms);
    if ((e = sl->d_pos.reserve((size_t)n * 4 + 64)) || (e = sl->d_mapq.reserve((size_t)n + 64)) || (e = sl->d_flag.reserve((size_t)n * 2 + 64)) ||
        (e = sl->d_lq.reserve((size_t)n * 4 + 64)) || (e = sl->d_ncg.reserve((size_t)n * 2 + 64)) || (e = sl->d_noq.reserve((size_t)n + 64)) ||
        (e = sl->d_tot.reserve(sizeof(StTotals))))
        return e;
    uint8_t *R = s->d_rec[s->cur].as<uint8_t>() + s->sync_base;
    const int64_t *ro = s->d_recoff.as<int64_t>() + g.first_read;
    HIPCHK(hipMemsetAsync(sl->d_tot.p, 0, sizeof(StTotals), st));
    const int64_t step = n > 65536 ? n / 65536 : 1;
    if (n) CRUMBLE_LAUNCH(k_st_fields, (unsigned)((n + 255) / 256 < 512 ? (n + 255) / 256 : 512), 256, st, R, ro, n, sl->d_pos.as<int32_t>(), sl->d_mapq.as<uint8_t>(), sl->d_flag.as<uint16_t>(),
                          sl->d_lq.as<int32_t>(), sl->d_ncg.as<uint16_t>(), sl->d_noq.as<uint8_t>(), sl->d_tot.as<StTotals>(), s->d_err.as<uint32_t>(), step);
    HIPCHK(hipGetLastError());
    StTotals tot;
    HIPCHK(copy_sync(st, &tot, sl->d_tot.p, sizeof tot, hipMemcpyDeviceToHost));
    if (tot.unsupported) { g_last_error = "crumble_stream: a CIGAR of more than 65535 operations (CG tag): not taken on the device path"; return CRUMBLE_EINVAL; }
    crumble_batch b;
    memset(&b, 0, sizeof b);
    b.tid = g.tid; b.n_reads = n; b.n_bases = (int64_t)tot.n_bases; b.n_cigar_ops = (int64_t)tot.n_cigar;
    b.n_bed = g.tid >= 0 ? n_bed : 0; b.bed_start = bed_start; b.bed_end = bed_end;
    DevSrc ds;
    memset(&ds, 0, sizeof ds);
    ds.rec = R; ds.rec_off = ro;
    ds.pos = sl->d_pos.as<int32_t>(); ds.mapq = sl->d_mapq.as<uint8_t>(); ds.flag = sl->d_flag.as<uint16_t>();
    ds.lq = sl->d_lq.as<int32_t>(); ds.ncg = sl->d_ncg.as<uint16_t>(); ds.noq = sl->d_noq.as<uint8_t>();
    for (int i = 0; i < 256; i++) ds.mq_hist[i] = tot.mq_hist[i];
    ds.lowerr = tot.err_cnt > 0 && (double)tot.err_ppm < 20000.0 * (double)tot.err_cnt;
    /* the engine's own stream from here on: what the slot's stream wrote is complete (the read-back above) */
    HIPCHK(hipEventRecord(h->ev_up[0], h->stream));
    if ((e = copy_raw(h, &b, &ds))) return e;
    if ((e = expand(h, true))) return e;
    if ((e = crumble_hip_run(h))) return e;
    if (n && b.n_bases)
        CRUMBLE_LAUNCH(k_st_scatter, (unsigned)((n + 63) / 64), 256, h->stream, R, ro, n, h->d_seq_off.as<int64_t>(), sl->d_noq.as<uint8_t>(), h->d_out.as<uint8_t>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    s->groups[(size_t)group].n_bases = b.n_bases;
    return CRUMBLE_OK;
}

/* The first k contig runs are done (processed, or meant to pass unchanged): they join what waits at the front of the buffer
 * (crumble_stream_pending), and as much of that as out_cap holds leaves as BGZF members of 0xff00 bytes -- whole members
 * only, unless flush is set and everything fits: then the last, shorter member too (the end of the file, or of a piece the
 * caller wants closed).  out_cap 0: nothing is written, the runs are only marked.  The remaining runs keep their order and
 * are numbered from 0 again. */
/* marks the first k runs done and says how many bytes of the front of the buffer a drain into out_cap bytes takes (0: none) */
static int stream_drain_plan(crumble_stream *s, int k, int flush, int64_t out_cap, int64_t *take_out)
{
    *take_out = 0;
    if (s->inflight) { g_last_error = "crumble_stream_drain: a drain is in flight (crumble_stream_drain_end first)"; return CRUMBLE_EINVAL; }
    if (s->hdr) return k ? CRUMBLE_EINVAL : CRUMBLE_OK;
    if (k) {
        s->start = s->groups[(size_t)k - 1].byte_end - (s->base - s->sync_base);
        s->groups.erase(s->groups.begin(), s->groups.begin() + k);
    }
    const int64_t avail = s->start;
    if (!avail || !out_cap) return CRUMBLE_OK;
    int64_t take = avail;
    if (crumble_gz_deflate_bound(avail) > out_cap) take = out_cap / (GZ_MAXBLK + 31) * GZ_MAXBLK;
    else if (!flush) take = avail / GZ_MAXBLK * GZ_MAXBLK;
    if (take <= 0) {
        if (crumble_gz_deflate_bound(avail) > out_cap) { g_last_error = "crumble_stream_drain: the output buffer does not hold one BGZF member"; return CRUMBLE_EINVAL; }
        return CRUMBLE_OK;                                  /* (less than a member waits, and no flush) */
    }
    *take_out = take;
    return CRUMBLE_OK;
}

extern "C" int crumble_stream_drain(crumble_stream *s, int k, int level, int flush, uint8_t *out, int64_t out_cap, int64_t *out_len)
{
    if (!s || k < 0 || k > (int)s->groups.size() || !out_len || out_cap < 0 || (out_cap && !out)) return CRUMBLE_EINVAL;
    *out_len = 0;
    int64_t take = 0;
    int e;
    if ((e = stream_drain_plan(s, k, flush, out_cap, &take)) || !take) return e;
    HIPCHK(hipSetDevice(s->device));
    uint8_t *R = s->d_rec[s->cur].as<uint8_t>() + s->base;
    { StClock clk(&s->ms_deflate); if ((e = gz_deflate_impl(s->gz, nullptr, R, take, level, out, out_cap, out_len))) return e; }
    s->bytes_out += *out_len;
    s->base += take; s->len -= take; s->start -= take; s->tail -= take;
    return CRUMBLE_OK;
}

/* The same drain in two halves: begin marks the runs, enqueues the deflate of what a buffer of out_cap bytes will take
 * (*taken: that many record bytes; 0: nothing to do, no end needed) and returns; crumble_stream_drain_end waits and copies
 * the members out.  In between the caller may process further runs (crumble_stream_process works on other streams and
 * other bytes) -- not feed, not drain.  out_cap: at most a codec round (256 MB of records). */
extern "C" int crumble_stream_drain_begin(crumble_stream *s, int k, int level, int flush, int64_t out_cap, int64_t *taken)
{
    if (!s || k < 0 || k > (int)s->groups.size() || !taken || out_cap < 0) return CRUMBLE_EINVAL;
    *taken = 0;
    int64_t take = 0;
    int e;
    if ((e = stream_drain_plan(s, k, flush, out_cap, &take)) || !take) return e;
    HIPCHK(hipSetDevice(s->device));
    uint8_t *R = s->d_rec[s->cur].as<uint8_t>() + s->base;
    s->t_begin = st_now();
    if ((e = gz_deflate_begin(s->gz, R, take, level))) return e;
    s->base += take; s->len -= take; s->start -= take; s->tail -= take;
    s->inflight = true;
    *taken = take;
    return CRUMBLE_OK;
}
extern "C" int crumble_stream_drain_end(crumble_stream *s, uint8_t *out, int64_t out_cap, int64_t *out_len)
{
    if (!s || !out || !out_len) return CRUMBLE_EINVAL;
    *out_len = 0;
    if (!s->inflight) return CRUMBLE_OK;
    s->inflight = false;
    const int e = gz_deflate_end(s->gz, out, out_cap, out_len);
    s->ms_deflate += (float)(st_now() - s->t_begin);         /* (wall time from begin: includes what the caller did in between) */
    if (e) return e;
    s->bytes_out += *out_len;
    return CRUMBLE_OK;
}

/* bytes that were processed (or skipped) and not drained yet: the front of the buffer */
extern "C" int64_t crumble_stream_pending(crumble_stream *s) { return s && !s->hdr ? s->start : 0; }

/* the bytes of the buffer (tests): d_rec[cur][from, from + n) -> host */
extern "C" int crumble_stream_peek(crumble_stream *s, int64_t from, int64_t n, uint8_t *out, int64_t *len, int64_t *n_rec, int64_t *rec_off, int64_t max_rec)
{
    if (!s) return CRUMBLE_EINVAL;
    HIPCHK(hipSetDevice(s->device));
    if (len) *len = s->len;
    if (n_rec) *n_rec = s->n_rec;
    if (out && n > 0) { if (from < 0 || from + n > s->len) return CRUMBLE_EINVAL; HIPCHK(copy_sync(s->gz->stream, out, s->d_rec[s->cur].as<uint8_t>() + s->base + from, (size_t)n, hipMemcpyDeviceToHost)); }
    if (rec_off && max_rec > 0 && s->n_rec) HIPCHK(copy_sync(s->gz->stream, rec_off, s->d_recoff.p, (size_t)(s->n_rec < max_rec ? s->n_rec : max_rec) * 8, hipMemcpyDeviceToHost));
    return CRUMBLE_OK;
}

/* host-clock milliseconds so far: ms[0..4] = inflate (H2D + kernels), record delimiting, engine (fields + expansion + run +
 * scatter), deflate (kernels + D2H), moving the rest of the buffer */
extern "C" int crumble_stream_times(crumble_stream *s, float ms[5], int64_t *bytes_in, int64_t *bytes_out)
{
    if (!s) return CRUMBLE_EINVAL;
    if (ms) { ms[0] = s->ms_inflate; ms[1] = s->ms_sync; ms[2] = 0; for (auto &c : s->slot) ms[2] += c.ms; ms[3] = s->ms_deflate; ms[4] = s->ms_move; }
    if (bytes_in) *bytes_in = s->bytes_in;
    if (bytes_out) *bytes_out = s->bytes_out;
    return CRUMBLE_OK;
}

#endif /* CRUMBLE_STREAM_HOST_H */
