/*
 * engine.hip -- host layer of the engine + the C ABI of include/crumble_hip.h.
 *
 * Replaces the body of transcode()'s pileup loop and flush_bam_list()'s
 * per-read epilogue (/root/reference/snp_score.c:1437-1983, 1090-1100) for one
 * contig: the batch is expanded on the host (P0: CIGAR -> reference-op segment
 * tables, read ends, per-tile candidate ranges), made resident in HBM, and run
 * through the kernels of kernels.h on one HIP stream.  No CPU fallback exists:
 * without a HIP device crumble_hip_create() fails with CRUMBLE_ENODEVICE.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <float.h>
#include <vector>
#include <string>
#include <algorithm>
#include <thread>
#include <atomic>
#include <mutex>
#ifndef CRUMBLE_EMU
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#endif

#include "../../include/crumble_hip.h"
#include "dev_common.h"
#include "kernels.h"
#include "p0_kernels.h"
#include "gz_kernels.h"
#include "stream_kernels.h"

static thread_local std::string g_last_error;
extern "C" const char *crumble_hip_last_error(void) { return g_last_error.c_str(); }

#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    char buf_[512]; snprintf(buf_, sizeof buf_, "%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
    g_last_error = buf_; return CRUMBLE_EHIP; } } while (0)

/* blocking copies / fills ON THE ENGINE'S STREAM (a non-blocking stream: the null stream's implicit ordering does not
 * reach it, so nothing in this file uses the null stream) */
static inline hipError_t copy_sync(hipStream_t st, void *dst, const void *src, size_t n, hipMemcpyKind kind)
{
    hipError_t e = hipMemcpyAsync(dst, src, n, kind, st);
    return e != hipSuccess ? e : hipStreamSynchronize(st);
}
static inline hipError_t fill_sync(hipStream_t st, void *dst, int v, size_t n)
{
    hipError_t e = hipMemsetAsync(dst, v, n, st);
    return e != hipSuccess ? e : hipStreamSynchronize(st);
}

/* grow-only device buffer */
/* device allocations of this process: how many, and the host time they took (crumble_hip_alloc_stats: a diagnostic -- a first
 * contig's seventy-odd buffers are allocated inside a driver's timed loop) */
static std::atomic<long long> g_alloc_calls{0}, g_alloc_us{0};
/* A buffer that grows does not free its old memory on the spot: hipFree waits for the whole device, i.e. for whatever the other
 * engines of the process (or a deflate in flight) have running -- milliseconds per growing buffer while a driver's first contigs
 * size seventy of them.  The old blocks wait here and go when an object is destroyed, or at the start of a contig once more
 * than a gigabyte of them has piled up. */
static std::mutex g_defer_mu;
static std::vector<void *> g_defer_ptr;
static size_t g_defer_bytes = 0;
static void dev_free_later(void *p, size_t n)
{
    std::lock_guard<std::mutex> lk(g_defer_mu);
    g_defer_ptr.push_back(p); g_defer_bytes += n;
}
static void dev_free_deferred(size_t keep_below)
{
    std::vector<void *> v;
    {
        std::lock_guard<std::mutex> lk(g_defer_mu);
        if (g_defer_bytes <= keep_below) return;
        v.swap(g_defer_ptr); g_defer_bytes = 0;
    }
    for (void *p : v) (void)hipFree(p);
}
struct DevBuf {
    void *p = nullptr; size_t cap = 0;
    bool owned = true;              /* false: a view into another buffer (the raw arena of an upload) */
    void view(void *ptr, size_t n) { if (owned && p) dev_free_later(p, cap); p = ptr; cap = n; owned = false; }
    int reserve(size_t n) {
        if (!owned) { p = nullptr; cap = 0; owned = true; }
        if (n <= cap && p) return CRUMBLE_OK;
        struct timespec t0_, t1_;
        clock_gettime(CLOCK_MONOTONIC, &t0_);
        if (p) dev_free_later(p, cap);
        p = nullptr; cap = 0;
        size_t want = n + n / 8 + 256;
        bool ok = hipMalloc(&p, want) == hipSuccess;
        if (!ok) { dev_free_deferred(0); (void)hipGetLastError(); ok = hipMalloc(&p, want) == hipSuccess; }      /* (out of memory with old blocks waiting: they go first) */
        clock_gettime(CLOCK_MONOTONIC, &t1_);
        g_alloc_calls.fetch_add(1, std::memory_order_relaxed);
        g_alloc_us.fetch_add((long long)((t1_.tv_sec - t0_.tv_sec) * 1000000ll + (t1_.tv_nsec - t0_.tv_nsec) / 1000), std::memory_order_relaxed);
        if (!ok) { p = nullptr; g_last_error = "hipMalloc failed"; return CRUMBLE_ENOMEM; }
        cap = want;
        return CRUMBLE_OK;
    }
    void release() { if (p && owned) (void)hipFree(p); p = nullptr; cap = 0; owned = true; }
    template <class T> T *as() const { return (T *)p; }
};

#define MAX_STAGES 16
#define FOLD_SMALL_MAX 8192         /* trigger columns (or capacity) up to which one workgroup does the whole fold */
#define DL_LANES 8
#define DL_CHUNK ((size_t)8 << 20)
struct crumble_hip {
    int device = 0;
    hipStream_t stream = nullptr;
    crumble_params params;
    DParams hparams;
    DevBuf d_params, d_tables, d_stats, d_batch_stats, d_misc, d_mqslot;
    DevBuf d_qualsp;                /* the coded qualities of an upload (crumble_batch.qualsp; a view into d_raw): k_p0_quals makes d_quals from them */
    int raw_qbits = 0; uint8_t raw_qlut[16] = {0};
    DevBuf d_lq, d_ncg;             /* BAM's own per-read lengths, when the batch carries them instead of the offset tables (views into d_raw) */
    size_t up_bytes = 0, down_bytes = 0;    /* bytes the last upload copied from the host / the last download copied to it */
    DevBuf d_raw;                   /* arena of the raw arrays of an upload: d_pos, d_mapq, d_flag, d_seq_off, d_cig_off, d_cigar, d_quals,
                                     * d_bases4 (or d_bases), d_bed_*, d_mqslot are views into it */
    void *stage = nullptr;          /* page-locked staging buffer of small uploads (RAW_STAGE_MAX bytes) */
    bool late_pending = false;      /* an expansion is in flight whose late errors (pinned + 256) have not been looked at */
    /* crumble_hip_run's fast path: the work lists are reserved with capacities guessed from the batch's shape and every
     * kernel takes the lists' true lengths from device memory, so the launches need no size read-back in the middle;
     * the scan sets an abort flag when a list would not fit (or a column is too deep) and the run is repeated the
     * careful way with exact sizes (what a window of a contig always does). */
    bool spec = false, spec_aborted = false;
    bool fold_small = false;        /* the last fold took k_fold_small (it marks the windows too) */
    int cap_trig = 0, cap_items = 0, cap_ins = 0;       /* capacities the lists hold (grow-only, remembered across batches) */
    int opt_spec = 1;               /* "run" = fast | careful: crumble_hip_run's first attempt (tests force the second way) */
    int opt_caps = 0;               /* "caps" = n: the fast path's capacities forced to n (tests: a forced abort) */
    int opt_stages = 0;             /* "stages" = on: HIP events around every stage of a run (crumble_hip_kernel_times; bench.py) */
    /* batch (inputs) */
    DevBuf d_genlist, d_vrec, d_pbdirty, d_pbcarry, d_pbchg, d_bases4, d_ufirst, d_tcum, d_tpart, d_pbwmm, d_act, d_pbstop, d_pbrs, d_pbruns, d_pbval, d_flag, d_p0cnt, d_p0part, d_sr_ppos, d_sr_pend, d_kpos, d_kend, d_slow_g, d_slow_r, d_cix, d_sr, d_stile_lo, d_stile_hi, d_rr, d_pos, d_end, d_mapq, d_inplp, d_seq_off, d_cig_off, d_seg_off, d_cigar, d_segs,
           d_bases, d_quals, d_bases_x, d_quals_x, d_tile_lo, d_tile_hi, d_bed_start, d_bed_end, d_bed_pmax, d_tile_bed, d_hitoff, d_hits, d_rtoff, d_first;
    /* per position */
    DevBuf d_colrec, d_nplp, d_score, d_trigcnt, d_dpre, d_col_ord, d_trig_ord, d_item_off,
           d_ins_ord, d_pos_mn2, d_partial, d_epochs, d_ht;
    /* work lists */
    DevBuf d_trigcol_pos, d_trigcol_off, d_inscol_pos, d_items, d_colhi, d_ub, d_pmax_partial, d_trig_mx2;
    /* outputs */
    DevBuf d_out, d_rflags, d_keepf, d_b15f, d_blk_read, d_chunk_read, d_kept, d_marks;
    DevBuf d_codes, d_escoff, d_esc;        /* the result in coded form (crumble_hip_encode) */
    bool have_coded = false;
    int64_t coded_escapes = 0;              /* what the last crumble_hip_encode reported (crumble_hip_download_coded copies exactly that) */
    DBatch B;
    bool have_batch = false, have_run = false;
    /* what crumble_hip_expand() needs beyond the raw arrays on the device (set by the copy half of an upload) */
    bool have_raw = false, raw_packed4 = false, raw_lowerr = false;
    int32_t raw_tid = 0, raw_nbed = 0;
    int raw_hot_mq = 0, raw_nslots = 0;
    uint8_t raw_mqtab[256 + 64];
    std::vector<int32_t> raw_pmax;
    std::vector<int64_t> raw_so, raw_co;    /* a small batch's offset tables made from its per-read lengths on the host */
    unsigned persistent_blocks = 1024;
    bool col_perm = false;          /* k_columns<.., PERM>: picked per batch in crumble_hip_upload */
    /* crumble_hip_set_option(): kernel choices that never change a result (tests force every variant) */
    int opt_kcol = 0;               /* 0 auto (k_columns2 where it applies), 1 k_columns, 2 k_columns PERM, 3 k_columns2 */
    int opt_pblock = 0;             /* 0 auto, 1 one lane per read, 2 segment-parallel (round 2), 3 word-parallel (k_pb_words / k_pb_chains) */
    int opt_replay = 0;             /* 0 auto, 1 k_replay_flat */
    int b_kcol = 0, b_pblock = 0;   /* the batch's snapshot of opt_kcol / opt_pblock (options take effect at an upload) */
    int64_t n_reads = 0, n_bases = 0, n_cigar = 0, n_slow = 0, n_segrec = 0, n_general_max = 0;
    void *pinned = nullptr;         /* 256 bytes of page-locked host memory: the small device->host read-backs land here */
    /* download staging (staged_d2h): DL_LANES host threads, each with its own stream and two page-locked buffers */
    struct DlLane { void *buf[2] = {nullptr, nullptr}; hipStream_t st = nullptr; hipEvent_t ev[2] = {nullptr, nullptr}; } dl[8];
    bool dl_ok = false;
    size_t dl_bufsz = 0;            /* size of each staging buffer */
    size_t dl_chunk = (size_t)8 << 20;
    int opt_dl_kernel = 1;          /* "dl_kernel": a page-locked destination is written by k_d2h_store (the GPU's own stores over the link)
                                     * instead of a copy-engine transfer that queues behind the uploads of the other handles */
    bool packed_ok = false;         /* the packed replay's value formulas hold for these parameters (set at create) */
    bool have_vrec = false;         /* long-read batch: territories + chunk table built (k_p0_vrecs) */
    bool batch_flat = false;        /* "replay" = flat as it stood when the batch was uploaded (decides which tables exist) */
    ScanVal totals;
    /* window of a contig (crumble_hip_set_window) and the phase the run is in */
    bool windowed = false;
    crumble_window win;
    int64_t own_b_lo = 0, own_b_hi = 0;     /* byte range of the owned reads */
    int phase = 0;                          /* 0 nothing run; 1 columns; 2 depth; 3 triggers; 4 fold; 5 finished */
    bool have_hits = false;                 /* long-read batch: hit_off / hits are built (k_p0_hitfill) */
    bool scan_fused = false;                /* no per-column prefix arrays: k_epochs_find + k_scan_overdepth (whole contig, no too-deep column) */
    crumble_carry carry_in;                 /* what phases 2 and 4 started from */
    /* timing */
    hipEvent_t ev[MAX_STAGES + 1];
    hipEvent_t ev_up[3];            /* crumble_hip_upload: start / raw arrays copied / expanded */
    float up_h2d_ms = 0, up_p0_ms = 0;
    const char *stage_name[MAX_STAGES];
    double stage_bytes[MAX_STAGES];
    float stage_ms[MAX_STAGES];
    int n_stages = 0;
    bool events_ok = false;
};

/* ---------------------------------------------------------------- tables */
static void build_tables(DTables *t, const DParams *dp)
{
    /* consensus_init(P_HET = 1e-6) for STECH_SOLEXA (snp:378-489) and the
     * function statics of calculate_consensus_pileup (snp:560-575), evaluated
     * with the host libm exactly as the reference does. */
    double pMM[101], p__[101], p_M[101], prior[25], q2p[101], mqual_pow[256];
    const double p_het = 1e-6;
    int i;
    for (i = -500; i <= 500; i++) t->e_tab[i + 500] = exp(i);
    for (i = -500; i <= 500; i++) t->e_tab2[i + 500] = exp(i / 10.);
    for (i = 0; i < 25; i++) prior[i] = p_het / 20;
    prior[0] = prior[6] = prior[12] = prior[18] = prior[24] = (1 - p_het) / 5;
    {
        static const int idx[15] = {0,1,2,3,4, 6,7,8,9, 12,13,14, 18,19, 24};
        static const int dbl[15] = {0,1,1,1,1, 0,1,1,1, 0,1,1, 0,1, 0};
        for (i = 0; i < 15; i++) t->lprior15[i] = dbl[i] ? log(prior[idx[i]] * 2) : log(prior[idx[i]]);
    }
    for (i = 1; i < 101; i++) {
        double prob = 1 - pow(10, -i / 10.0);
        pMM[i] = log(prob / 5);
        p__[i] = log((1 - prob) / 20);
        p_M[i] = log((exp(pMM[i]) + exp(p__[i])) / 2);
    }
    pMM[0] = pMM[1]; p__[0] = p__[1]; p_M[0] = p_M[1];
    for (i = 0; i <= 100; i++) q2p[i] = pow(10, -i / 10.0);
    for (i = 0; i < 255; i++) mqual_pow[i] = 1 - pow(10, -(i / 2 + .05) / 10.0);
    mqual_pow[255] = mqual_pow[10];
    for (i = 0; i <= 100; i++) {
        t->MMd[i] = pMM[i] - p__[i];          /* snp:645 */
        t->_Md[i] = p_M[i] - p__[i];          /* snp:646 */
        t->omq[i] = 1 - q2p[i];               /* snp:651 */
    }
    t->min_e_exp = DBL_MIN_EXP * log(2) + 1;  /* snp:540 */
    for (int m = 0; m < 256; m++)
        for (int q = 0; q < 256; q++) {       /* snp:632-638, ph_log on the host */
            double _p = mqual_pow[q], _m = mqual_pow[m];
            double x = 1 - (_m * _p + (1 - _m) / 4);
            int64_t bits; memcpy(&bits, &x, 8);
            int log_2 = (int)((bits >> 52) & 2047) - 1024;
            bits &= ~(2047LL << 52); bits += 1023LL << 52;
            double val; memcpy(&val, &bits, 8);
            val = ((-1.0f / 3) * val + 2) * val - 2.0f / 3;
            double ph = -3.0103 * (val + log_2);
            t->adj[m * 256 + q] = (uint8_t)ph;
        }
    /* parameter-dependent compositions for k_columns2 (row indices of its LDS table) */
    for (int q = 0; q < 256; q++) {
        const int cq = (q > dp->qcap && !dp->preserve_qual[q]) ? dp->qcap : q;      /* cap_quality snp:1325-1332 */
        t->capc[q] = (uint8_t)(cq < 1 ? 1 : (cq > 100 ? 100 : cq));
        t->pvc[q] = dp->preserve_qual[cq];
        for (int m = 0; m < 256; m++) {
            const int a = t->adj[m * 256 + cq];
            t->adjc[m * 256 + q] = (uint8_t)(a < 1 ? 1 : (a > 100 ? 100 : a));
        }
    }
}

static int fill_dparams(const crumble_params *p, DParams *d)
{
    memset(d, 0, sizeof(*d));
    d->reduce_qual = p->reduce_qual; d->binary_qual = p->binary_qual;
    d->iSTR_add = p->iSTR_add; d->sSTR_add = p->sSTR_add;
    d->iSTR_mul = p->iSTR_mul; d->sSTR_mul = p->sSTR_mul;
    d->qlow = p->qlow; d->qcutoff = p->qcutoff; d->qhigh = p->qhigh; d->qcap = p->qcap;
    d->min_mqual = p->min_mqual;
    d->min_qual_A = p->min_qual_A; d->min_indel_A = p->min_indel_A;
    d->min_qual_B = p->min_qual_B; d->min_indel_B = p->min_indel_B;
    d->pblock = p->pblock; d->softclip = p->softclip; d->perfect_col = p->perfect_col;
    d->indel_fract = p->indel_fract; d->min_discrep_A = p->min_discrep_A;
    d->min_discrep_B = p->min_discrep_B; d->low_mqual_perc = p->low_mqual_perc;
    d->clip_perc = p->clip_perc; d->ins_len_perc = p->ins_len_perc;
    d->over_depth = p->over_depth; d->indel_ov_perc = p->indel_ov_perc;
    d->str_snp = (p->sSTR_add || p->sSTR_mul) ? 1 : 0;
    for (int i = 0; i < 256; i++) {
        d->preserve_qual[i] = p->preserve_qual[i];
        if (p->preserve_qual[i]) d->any_preserve = 1;
    }
    crumble_init_bins(p, d->bin2);
    d->n_pv = 0;                                    /* the preserved qualities as a short list (packed replay) */
    for (int i = 0; i < 256 && d->n_pv >= 0; i++)
        if (p->preserve_qual[i]) {
            if (i >= 128 || d->n_pv == 4) { d->n_pv = -1; break; }
            d->pv_val[d->n_pv] = (uint8_t)i; d->pv_lvl[d->n_pv] = p->preserve_qual[i]; d->n_pv++;
        }
    /* fences */
    if (p->qcap < 0 || p->qcap > 100 || p->qlow < 0 || p->qlow > 127 || p->qhigh < 0 || p->qhigh > 127) {
        g_last_error = "qcap must be in 0..100 and qlow/qhigh in 0..127 (SURVEY.md Q5/Q13)";
        return CRUMBLE_EINVAL;
    }
    for (int i = 101; i < 256; i++)
        if (p->preserve_qual[i]) { g_last_error = "preserved qualities must be <= 100 (SURVEY.md Q5)"; return CRUMBLE_EINVAL; }
    return CRUMBLE_OK;
}

/* ---------------------------------------------------------------- create */
extern "C" crumble_hip *crumble_hip_create(const crumble_params *p, int device, int *err)
{
    int e = CRUMBLE_OK, ndev = 0;
    crumble_hip *h = nullptr;
    DTables *tab = nullptr;
    auto fail = [&](int code) -> crumble_hip * {
        if (err) *err = code;
        if (h) crumble_hip_destroy(h);
        free(tab);
        return nullptr;
    };
    if (!p) return fail(CRUMBLE_EINVAL);
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        g_last_error = "no HIP device visible";
        return fail(CRUMBLE_ENODEVICE);
    }
    if (device < 0 || device >= ndev) { g_last_error = "device index out of range"; return fail(CRUMBLE_EINVAL); }
    h = new crumble_hip();
    h->device = device;
    h->params = *p;
    if ((e = fill_dparams(p, &h->hparams)) != CRUMBLE_OK) return fail(e);
    {   /* k_replay_fast's packed formulas: no -S, byte values that leave bit 7 free, at most four preserved qualities
         * (all < 128), and in binary mode bin2[] == (q >= qcutoff ? qhigh : qlow) on the values a base can have */
        const DParams &d = h->hparams;
        bool ok = !d.softclip && d.qcutoff >= 0 && d.qcutoff < 128 && d.qcap < 128 && d.n_pv >= 0;
        if (ok && d.reduce_qual && d.binary_qual)
            for (int q = 0; q < 128; q++) if (d.bin2[q] != (q >= d.qcutoff ? d.qhigh : d.qlow)) ok = false;
        h->packed_ok = ok;
    }
    if (hipSetDevice(device) != hipSuccess) { g_last_error = "hipSetDevice failed"; return fail(CRUMBLE_EHIP); }
    /* a non-blocking stream: nothing this engine does waits for (or holds up) the null stream -- several engines on one
     * device (crumble-hip's pipeline, windows) overlap their copies and kernels */
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { g_last_error = "hipStreamCreate failed"; return fail(CRUMBLE_EHIP); }
    if (hipHostMalloc(&h->pinned, 512, 0) != hipSuccess) { h->pinned = nullptr; g_last_error = "hipHostMalloc failed"; return fail(CRUMBLE_ENOMEM); }
    for (int i = 0; i <= MAX_STAGES; i++)
        if (hipEventCreate(&h->ev[i]) != hipSuccess) { g_last_error = "hipEventCreate failed"; return fail(CRUMBLE_EHIP); }
    for (int i = 0; i < 3; i++)
        if (hipEventCreate(&h->ev_up[i]) != hipSuccess) { g_last_error = "hipEventCreate failed"; return fail(CRUMBLE_EHIP); }
    h->events_ok = true;
#ifndef CRUMBLE_EMU
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0)
            h->persistent_blocks = (unsigned)cus * 4;          /* 4 x 256 threads per CU: 4 waves per SIMD */
    }
#else
    h->persistent_blocks = 3;
    h->opt_kcol = 1;        /* under the CPU emulator k_columns2's per-row wave rendezvous cost ~10x k_columns: the
                             * emulated tier takes k_columns unless a test asks for "kcol" = v2 */
#endif
    tab = (DTables *)calloc(1, sizeof(DTables));
    if (!tab) return fail(CRUMBLE_ENOMEM);
    build_tables(tab, &h->hparams);
    if ((e = h->d_params.reserve(sizeof(DParams))) || (e = h->d_tables.reserve(sizeof(DTables))) ||
        (e = h->d_stats.reserve(CRUMBLE_ST_N * 8)) || (e = h->d_batch_stats.reserve(CRUMBLE_ST_N * 8 + 256)) ||
        (e = h->d_misc.reserve(256)))
        return fail(e);
    if (copy_sync(h->stream, h->d_params.p, &h->hparams, sizeof(DParams), hipMemcpyHostToDevice) != hipSuccess ||
        copy_sync(h->stream, h->d_tables.p, tab, sizeof(DTables), hipMemcpyHostToDevice) != hipSuccess ||
        fill_sync(h->stream, h->d_stats.p, 0, CRUMBLE_ST_N * 8) != hipSuccess ||
        fill_sync(h->stream, h->d_batch_stats.p, 0, CRUMBLE_ST_N * 8) != hipSuccess) {
        g_last_error = "uploading tables failed";
        return fail(CRUMBLE_EHIP);
    }
    free(tab);
    if (err) *err = CRUMBLE_OK;
    return h;
}

extern "C" void crumble_hip_destroy(crumble_hip *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    dev_free_deferred(0);
    DevBuf *all[] = { &h->d_genlist, &h->d_vrec, &h->d_params, &h->d_tables, &h->d_stats, &h->d_batch_stats, &h->d_misc, &h->d_mqslot,
        &h->d_ufirst, &h->d_tcum, &h->d_tpart, &h->d_pbdirty, &h->d_pbcarry, &h->d_pbchg, &h->d_bases4, &h->d_pbwmm, &h->d_act, &h->d_pbstop, &h->d_pbrs, &h->d_pbruns, &h->d_pbval, &h->d_flag, &h->d_p0cnt, &h->d_p0part, &h->d_sr_ppos, &h->d_sr_pend, &h->d_kpos, &h->d_kend,
        &h->d_slow_g, &h->d_slow_r, &h->d_cix, &h->d_sr, &h->d_stile_lo, &h->d_stile_hi, &h->d_rr, &h->d_pos, &h->d_end, &h->d_mapq, &h->d_inplp, &h->d_seq_off, &h->d_cig_off, &h->d_seg_off,
        &h->d_cigar, &h->d_segs, &h->d_bases, &h->d_quals, &h->d_bases_x, &h->d_quals_x, &h->d_tile_lo, &h->d_tile_hi,
        &h->d_bed_start, &h->d_bed_end, &h->d_bed_pmax, &h->d_tile_bed, &h->d_hitoff, &h->d_hits, &h->d_rtoff, &h->d_first, &h->d_colrec, &h->d_nplp, &h->d_score,
        &h->d_trigcnt, &h->d_dpre, &h->d_col_ord, &h->d_trig_ord, &h->d_item_off, &h->d_ins_ord,
        &h->d_pos_mn2, &h->d_partial, &h->d_epochs, &h->d_ht, &h->d_trigcol_pos, &h->d_trigcol_off,
        &h->d_inscol_pos, &h->d_items, &h->d_colhi, &h->d_ub, &h->d_pmax_partial, &h->d_trig_mx2,
        &h->d_out, &h->d_rflags, &h->d_keepf, &h->d_b15f, &h->d_blk_read, &h->d_chunk_read, &h->d_kept, &h->d_marks,
        &h->d_codes, &h->d_escoff, &h->d_esc, &h->d_lq, &h->d_ncg };
    for (DevBuf *b : all) b->release();
    if (h->events_ok) { for (int i = 0; i <= MAX_STAGES; i++) (void)hipEventDestroy(h->ev[i]); for (int i = 0; i < 3; i++) (void)hipEventDestroy(h->ev_up[i]); }
    if (h->pinned) (void)hipHostFree(h->pinned);
    if (h->stage) (void)hipHostFree(h->stage);
    h->d_raw.release();
    for (auto &l : h->dl) {
        for (int i = 0; i < 2; i++) { if (l.buf[i]) (void)hipHostFree(l.buf[i]); if (l.ev[i]) (void)hipEventDestroy(l.ev[i]); }
        if (l.st) (void)hipStreamDestroy(l.st);
    }
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

extern "C" int crumble_hip_set_option(crumble_hip *h, const char *name, const char *value)
{
    if (!h || !name || !value) return CRUMBLE_EINVAL;
    auto pick = [&](const char *const *names, int n) -> int {
        for (int i = 0; i < n; i++) if (!strcmp(value, names[i])) return i;
        return -1;
    };
    int v;
    if (!strcmp(name, "kcol")) {
        static const char *const nm[] = {"auto", "v1", "v1perm", "v2"};
        if ((v = pick(nm, 4)) < 0) return CRUMBLE_EINVAL;
        h->opt_kcol = v;
    } else if (!strcmp(name, "pblock")) {
        static const char *const nm[] = {"auto", "reads", "segments", "words"};
        if ((v = pick(nm, 4)) < 0) return CRUMBLE_EINVAL;
        h->opt_pblock = v;
    } else if (!strcmp(name, "replay")) {
        static const char *const nm[] = {"auto", "flat"};
        if ((v = pick(nm, 2)) < 0) return CRUMBLE_EINVAL;
        h->opt_replay = v;
    } else if (!strcmp(name, "run")) {
        static const char *const nm[] = {"careful", "fast"};
        if ((v = pick(nm, 2)) < 0) return CRUMBLE_EINVAL;
        h->opt_spec = v;
    } else if (!strcmp(name, "caps")) {                     /* tests: the fast path's list capacities (0: guessed from the batch) */
        const long long c = atoll(value);
        if (c < 0 || c > INT_MAX / 2) return CRUMBLE_EINVAL;
        h->opt_caps = (int)c;
    } else if (!strcmp(name, "stages")) {                   /* HIP events around the stages of a run (crumble_hip_kernel_times) */
        static const char *const nm[] = {"off", "on"};
        if ((v = pick(nm, 2)) < 0) return CRUMBLE_EINVAL;
        h->opt_stages = v;
    } else if (!strcmp(name, "dl_kernel")) {                /* page-locked destinations: written by a kernel instead of the copy engine */
        static const char *const nm[] = {"off", "on"};
        if ((v = pick(nm, 2)) < 0) return CRUMBLE_EINVAL;
        h->opt_dl_kernel = v;
    } else if (!strcmp(name, "dl_chunk")) {                 /* bytes per staged download copy (tests shrink it) */
        const long long c = atoll(value);
        if (c < 256 || c > (long long)DL_CHUNK) return CRUMBLE_EINVAL;
        h->dl_chunk = (size_t)c;
    } else return CRUMBLE_EINVAL;
    return CRUMBLE_OK;
}

/* ------------------------------------------------------------- P0 expand */
static inline unsigned nblk(int64_t n, int b) { return (unsigned)((n + b - 1) / b > 0 ? (n + b - 1) / b : 1); }

/* in-place inclusive prefix max of an int32 device array (chunked; the fold's bounds) */
/* (n_dev: the true length is on the device, n its capacity -- workgroups beyond the length leave at once) */
static int prefix_max(crumble_hip *h, int32_t *v, int64_t n, const int *n_dev = nullptr)
{
    if (n <= 0) return CRUMBLE_OK;
    const int nch = (int)((n + PMAX_CHUNK - 1) / PMAX_CHUNK);
    int e = h->d_pmax_partial.reserve((size_t)(nch + 2) * 4);
    if (e) return e;
    int32_t *pp = h->d_pmax_partial.as<int32_t>();
    CRUMBLE_LAUNCH(k_pmax_reduce, nch, 256, h->stream, (int)n, n_dev, v, pp);
    CRUMBLE_LAUNCH(k_pmax_partials, 1, 64, h->stream, pp, nch, (int)n, n_dev);
    CRUMBLE_LAUNCH(k_pmax_apply, nch, 256, h->stream, (int)n, n_dev, v, pp);
    return CRUMBLE_OK;
}

/* The copy half of an upload: the raw BAM-shaped arrays go to the device as they are, into ONE device arena
 * (d_raw; d_pos, d_mapq ... are views into it), plus two sampled hints taken from the host arrays (the batch's mapq
 * histogram, its mean base error): no per-read host work.  A small batch is packed into a page-locked staging buffer
 * and crosses in one copy (ten copies of a few kilobytes each were a third of a small contig's fixed cost); a large
 * one is copied array by array straight from the caller's memory. */
#define RAW_STAGE_MAX ((size_t)4 << 20)
#define LENS_ON_DEVICE_MIN 65536    /* reads from which on the offset tables are made on the device from the uploaded lengths */
/* the batch's per-read and per-base arrays are ON THE DEVICE already, inside BAM records (crumble_stream, stream_host.h):
 * where the records are, what k_st_fields found in them, and the hints copy_raw otherwise samples from host arrays */
struct DevSrc {
    const uint8_t *rec; const int64_t *rec_off;          /* device: the record bytes, record i at rec[rec_off[i]] */
    const int32_t *pos; const uint8_t *mapq; const uint16_t *flag; const int32_t *lq; const uint16_t *ncg; const uint8_t *noq;   /* device, [n] */
    int64_t mq_hist[256];                                /* host: sampled mapq histogram */
    bool lowerr;                                         /* host: mean base error below 2 % */
};
static int copy_raw(crumble_hip *h, const crumble_batch *b, const DevSrc *ds = nullptr)
{
    dev_free_deferred((size_t)1 << 30);                      /* (old blocks of grown buffers, once a gigabyte of them waits) */
    hipStream_t st = h->stream;
    const int64_t n = b->n_reads;
    /* the offset tables, or BAM's own per-read lengths (6 bytes per read over PCIe instead of 16; offsets made on the device) */
    const bool lens = ds || (b->l_qseq != nullptr && b->n_cigar != nullptr);
    if (!lens && n && (!b->seq_off || !b->cig_off)) { g_last_error = "crumble_batch: neither seq_off / cig_off nor l_qseq / n_cigar"; return CRUMBLE_EINVAL; }
    const int64_t nb = !n ? 0 : lens ? b->n_bases : b->seq_off[n], nc = !n ? 0 : lens ? b->n_cigar_ops : b->cig_off[n];
    if (n >= (int64_t)INT_MAX - 64) { g_last_error = "more than 2^31 reads in one batch"; return CRUMBLE_EINVAL; }
    if (nb < 0 || nc < 0) { g_last_error = "negative total length"; return CRUMBLE_EINVAL; }
    h->have_batch = false; h->have_run = false; h->have_raw = false; h->have_hits = false;
    const bool packed4 = ds || b->bases4 != nullptr;      /* BAM-packed bases: half the bytes over PCIe, unpacked by k_p0_bases */
    if (!packed4 && !b->bases && nb > 0) { g_last_error = "crumble_batch: neither bases nor bases4"; return CRUMBLE_EINVAL; }
    const int qbits = (!ds && b->qualsp) ? b->qual_bits : 0;   /* coded qualities: a quarter / half of the bytes over PCIe, expanded by k_p0_quals */
    if (qbits != 0 && qbits != 2 && qbits != 4) { g_last_error = "crumble_batch: qual_bits must be 0, 2 or 4"; return CRUMBLE_EINVAL; }
    if (!ds && !qbits && !b->quals && nb > 0) { g_last_error = "crumble_batch: neither quals nor qualsp"; return CRUMBLE_EINVAL; }
    const int n_bed = b->n_bed > 0 ? b->n_bed : 0;
    int e;
    {   /* the mapq most reads carry: the column kernels keep a composed quality table for it in LDS; the next
         * KC2_SLOTS most frequent ones get a row of k_columns2's compact adjustment table (their slot travels in
         * SegRec.info); anything else is looked up in global memory.  Performance hints only. */
        int64_t hist[256] = {0};
        const int64_t hstep = n > 65536 ? n / 65536 : 1;     /* a sample: mapq values it misses are looked up in global memory */
        if (ds) memcpy(hist, ds->mq_hist, sizeof hist);
        else for (int64_t i = 0; i < n; i += hstep) hist[b->mapq[i]]++;
        int order[256], best = 0;
        for (int i = 0; i < 256; i++) { order[i] = i; if (hist[i] > hist[best]) best = i; }
        h->raw_hot_mq = best;
        std::sort(order, order + 256, [&](int x, int y) { return hist[x] != hist[y] ? hist[x] > hist[y] : x < y; });
        uint8_t *tab = h->raw_mqtab;
        memset(tab, 255, 256); memset(tab + 256, 0, KC2_SLOTS);
        int ns = 0;
        for (int k = 0; k < 256 && ns < KC2_SLOTS; k++) {
            const int m = order[k];
            if (m == best || hist[m] == 0) continue;
            tab[m] = (uint8_t)ns; tab[256 + ns] = (uint8_t)m; ns++;
        }
        h->raw_nslots = ns;
    }
    {   /* k_columns variant (a performance choice only; both give the same bits): per-column base classes pay
         * when few base calls disagree with their column.  Estimate the batch's error rate from a sample of its
         * qualities; measured break-even is near 2 % (see kernels.h).  Option "kcol" overrides. */
        double esum = 0; int64_t cnt = 0;
        double q2e[128];                                     /* (per call: handles may be used from several threads) */
        for (int q = 0; q < 128; q++) q2e[q] = pow(10.0, -q / 10.0);
        const int64_t step = nb > 65536 ? nb / 65536 : 1;
        for (int64_t i = 0; !ds && i < nb; i += step, cnt++) {
            const int q = !qbits ? b->quals[i] : b->qual_lut[(b->qualsp[(i * qbits) >> 3] >> ((i * qbits) & 7)) & ((1 << qbits) - 1)];
            esum += q2e[q & 127];
        }
        h->raw_lowerr = ds ? ds->lowerr : (cnt > 0 && esum < 0.02 * (double)cnt);
    }
    /* keep-regions: the prefix max of `end` drives the cursor emulation (bed_preserve) */
    std::vector<int32_t> &pmax = h->raw_pmax;           /* (lives in the handle: the copy below is asynchronous) */
    pmax.resize(n_bed ? n_bed : 1);
    for (int i = 0; i < n_bed; i++) pmax[i] = (i && pmax[i - 1] > b->bed_end[i]) ? pmax[i - 1] : b->bed_end[i];

    /* the arena: every array 256-byte aligned with 64 bytes of slack behind it (kernels read whole 16-byte vectors);
     * qualities (and one-byte-per-base bases) carry BASE_FRONT_PAD zeroed bytes in front: k_columns2 loads the whole
     * 16-column segment that holds a read's first aligned base */
    static const int64_t zero64 = 0;
    struct Item { DevBuf *d; const void *src; size_t bytes, front; size_t off; size_t room; };
    const size_t b4bytes = packed4 ? (size_t)CRUMBLE_BASES4_SIZE(nb, n) : 0;
    /* a small batch's offsets are made right here (three more launches would cost a 5 kb contig more than its 6 bytes per
     * read save); a large one's by a scan on the device from the lengths it uploads instead */
    const bool dev_offsets = lens && (ds || n > LENS_ON_DEVICE_MIN);
    const int64_t *h_so = b->seq_off, *h_co = b->cig_off;
    if (lens && !dev_offsets && n > 0) {
        h->raw_so.resize((size_t)n + 1); h->raw_co.resize((size_t)n + 1);      /* (live in the handle: the copy below is asynchronous) */
        int64_t a = 0, c = 0;
        for (int64_t i = 0; i < n; i++) { h->raw_so[i] = a; h->raw_co[i] = c; a += b->l_qseq[i]; c += b->n_cigar[i]; }
        h->raw_so[n] = a; h->raw_co[n] = c;
        if (a != nb || c != nc) { g_last_error = "crumble_batch: n_bases / n_cigar_ops do not match the per-read lengths"; return CRUMBLE_EINVAL; }
        h_so = h->raw_so.data(); h_co = h->raw_co.data();
    }                                                     /* (room: bytes the view holds; bytes: what is copied from the host) */
    /* (a device source: nothing is copied from the host for the per-read / per-base arrays -- bytes 0, room = their size --,
     * k_st_gather fills them below) */
#define HB(x_) (ds ? (size_t)0 : (size_t)(x_))
    Item items[] = {
        {&h->d_pos, b->pos, HB(n * 4), 0, 0, (size_t)n * 4}, {&h->d_mapq, b->mapq, HB(n), 0, 0, (size_t)n}, {&h->d_flag, b->flag, HB(n * 2), 0, 0, (size_t)n * 2},
        {&h->d_seq_off, dev_offsets ? nullptr : n ? (const void *)h_so : (const void *)&zero64, dev_offsets ? 0 : (size_t)(n + 1) * 8, 0, 0, (size_t)(n + 1) * 8},
        {&h->d_cig_off, dev_offsets ? nullptr : n ? (const void *)h_co : (const void *)&zero64, dev_offsets ? 0 : (size_t)(n + 1) * 8, 0, 0, (size_t)(n + 1) * 8},
        {&h->d_lq, b->l_qseq, HB(dev_offsets ? n * 4 : 0), 0, 0, dev_offsets ? (size_t)n * 4 : 0}, {&h->d_ncg, b->n_cigar, HB(dev_offsets ? n * 2 : 0), 0, 0, dev_offsets ? (size_t)n * 2 : 0},
        {&h->d_cigar, b->cigar, HB(nc * 4), 0, 0, (size_t)nc * 4},
        {qbits ? &h->d_qualsp : &h->d_quals, qbits ? (const void *)b->qualsp : (const void *)b->quals,
         HB(qbits ? (size_t)CRUMBLE_QUALSP_SIZE(nb, qbits) - 16 : (size_t)nb), qbits ? 0 : (size_t)BASE_FRONT_PAD, 0, qbits ? (size_t)CRUMBLE_QUALSP_SIZE(nb, qbits) - 16 : (size_t)nb},
        {packed4 ? &h->d_bases4 : &h->d_bases, packed4 ? (const void *)b->bases4 : (const void *)b->bases, HB(packed4 ? b4bytes : (size_t)nb), packed4 ? 0 : (size_t)BASE_FRONT_PAD, 0, packed4 ? b4bytes : (size_t)nb},
        {&h->d_bed_start, b->bed_start, (size_t)n_bed * 4, 0, 0}, {&h->d_bed_end, b->bed_end, (size_t)n_bed * 4, 0, 0},
        {&h->d_bed_pmax, pmax.data(), (size_t)n_bed * 4, 0, 0}, {&h->d_mqslot, h->raw_mqtab, sizeof h->raw_mqtab, 0, 0},
    };
    size_t total = 0;
    for (Item &it : items) { if (it.room < it.bytes) it.room = it.bytes; it.off = total; total += (it.front + it.room + 2 * BASE_FRONT_PAD + 64 + 255) & ~(size_t)255; }
    if ((e = h->d_raw.reserve(total))) return e;
    uint8_t *base = h->d_raw.as<uint8_t>();
    for (Item &it : items) it.d->view(base + it.off, it.front + it.room + 2 * BASE_FRONT_PAD + 64);
    /* d_bases / d_quals are always VIEWS: into the arena when the batch carries the one-byte form, into a grow-only buffer
     * of their own (d_bases_x / d_quals_x) when the device makes it.  Batches that alternate between the two forms (coded
     * and plain qualities from one contig to the next) neither free nor allocate: hipFree synchronises the whole device. */
    if (packed4) {                                       /* the one-byte-per-base form is made on the device */
        if ((e = h->d_bases_x.reserve((size_t)nb + 2 * BASE_FRONT_PAD + 64))) return e;
        h->d_bases.view(h->d_bases_x.p, h->d_bases_x.cap);
        HIPCHK(hipMemsetAsync(h->d_bases.p, 0, BASE_FRONT_PAD, st));
    }
    if (qbits) {                                         /* likewise the one-byte-per-quality form (k_p0_quals, with the expansion) */
        if ((e = h->d_quals_x.reserve((size_t)nb + 2 * BASE_FRONT_PAD + 64))) return e;
        h->d_quals.view(h->d_quals_x.p, h->d_quals_x.cap);
        HIPCHK(hipMemsetAsync(h->d_quals.p, 0, BASE_FRONT_PAD, st));
    }
    bool staged = false;
    if (total <= RAW_STAGE_MAX) {
        if (!h->stage && hipHostMalloc(&h->stage, RAW_STAGE_MAX, 0) != hipSuccess) h->stage = nullptr;
        if (h->stage) {
            uint8_t *sg = (uint8_t *)h->stage;
            for (Item &it : items) {
                if (it.front) memset(sg + it.off, 0, it.front);
                if (it.bytes) memcpy(sg + it.off + it.front, it.src, it.bytes);
            }
            HIPCHK(hipMemcpyAsync(base, sg, total, hipMemcpyHostToDevice, st));
            staged = true;
        }
    }
    if (!staged)
        for (Item &it : items) {
            if (it.front) HIPCHK(hipMemsetAsync(base + it.off, 0, it.front, st));
            if (it.bytes) HIPCHK(hipMemcpyAsync(base + it.off + it.front, it.src, it.bytes, hipMemcpyHostToDevice, st));
        }
    if (ds && n > 0) {                                  /* the per-read fields: device to device (k_st_fields left them in the stream's arrays) */
        HIPCHK(hipMemcpyAsync(h->d_pos.p, ds->pos, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemcpyAsync(h->d_mapq.p, ds->mapq, (size_t)n, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemcpyAsync(h->d_flag.p, ds->flag, (size_t)n * 2, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemcpyAsync(h->d_lq.p, ds->lq, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemcpyAsync(h->d_ncg.p, ds->ncg, (size_t)n * 2, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemsetAsync(h->d_quals.p, 0, BASE_FRONT_PAD, st));
    }
    if (dev_offsets && n > 0) {                         /* seq_off[n + 1], cig_off[n + 1] from the lengths, on the device */
        const int nch = (int)((n + P0_CHUNK - 1) / P0_CHUNK);
        if ((e = h->d_tpart.reserve((size_t)(2 * (nch + 1) + 2) * 8))) return e;
        int64_t *tp = h->d_tpart.as<int64_t>();
        CRUMBLE_LAUNCH(k_p0_len_reduce, nch, 256, st, n, h->d_lq.as<int32_t>(), h->d_ncg.as<uint16_t>(), tp, nch);
        CRUMBLE_LAUNCH(k_i64scan_partials, 2, 1024, st, tp, nch);
        CRUMBLE_LAUNCH(k_p0_len_apply, nch, 256, st, n, h->d_lq.as<int32_t>(), h->d_ncg.as<uint16_t>(), (const int64_t *)tp, nch,
                       h->d_seq_off.as<int64_t>(), h->d_cig_off.as<int64_t>());
    }
    if (dev_offsets && n == 0) { static const int64_t z2[1] = {0}; HIPCHK(hipMemcpyAsync(h->d_seq_off.p, z2, 8, hipMemcpyHostToDevice, st)); HIPCHK(hipMemcpyAsync(h->d_cig_off.p, z2, 8, hipMemcpyHostToDevice, st)); }
    if (ds && n > 0)                                    /* CIGAR words, packed bases, qualities out of the records */
        CRUMBLE_LAUNCH(k_st_gather, (unsigned)((n + 63) / 64), 256, st, ds->rec, ds->rec_off, n, h->d_seq_off.as<int64_t>(), h->d_cig_off.as<int64_t>(),
                       ds->noq, h->d_cigar.as<uint32_t>(), h->d_bases4.as<uint8_t>(), h->d_quals.as<uint8_t>() + BASE_FRONT_PAD);
    h->raw_tid = b->tid; h->raw_nbed = n_bed; h->raw_packed4 = packed4;
    h->raw_qbits = qbits; memcpy(h->raw_qlut, b->qual_lut, 16);
    h->n_reads = n; h->n_bases = nb; h->n_cigar = nc;
    h->up_bytes = 0;
    for (const Item &it : items) h->up_bytes += it.bytes;
    HIPCHK(hipEventRecord(h->ev_up[1], st));
    h->have_raw = true;
    return CRUMBLE_OK;
}

/* The device half: everything the column engine needs beyond the raw arrays is derived on the device (p0_kernels.h).
 * One read-back in the middle (the sizes of the expanded tables; the routing / offset / length errors), the rest is
 * asynchronous; `wait`: also wait for the end and report the late errors (coordinate order, tile span). */
static int expand_late_errors(crumble_hip *h)
{
    const int *res = (const int *)((const char *)h->pinned + 256);
    switch (res[0]) {
    case 0: break;
    case P0_ERR_UNSORTED: return CRUMBLE_EUNSORTED;                      /* snp:1075 */
    case P0_ERR_SPAN: g_last_error = "the reads over one 64-column tile span more than 3.7 GB of bases"; return CRUMBLE_EINVAL;
    case P0_ERR_QUAL: g_last_error = "input quality >= 128 (SURVEY.md Q13)"; return CRUMBLE_EINVAL;
    default: g_last_error = "internal: batch expansion failed"; return CRUMBLE_EOVERFLOW;
    }
    if (res[1]) { g_last_error = "internal: tile hit lists inconsistent"; return CRUMBLE_EOVERFLOW; }
    return CRUMBLE_OK;
}
static int expand(crumble_hip *h, bool wait)
{
    hipStream_t st = h->stream;
    const int64_t n = h->n_reads, nb = h->n_bases, nc = h->n_cigar;
    const int n_bed = h->raw_nbed;
    const bool packed4 = h->raw_packed4;
    const int32_t tid = h->raw_tid;
    h->have_batch = false; h->have_run = false; h->have_hits = false; h->have_vrec = false; h->late_pending = false;
    int e;
    HIPCHK(hipEventRecord(h->ev_up[1], st));
    if (h->raw_qbits && nb > 0) {                        /* 0. the qualities, one byte each, from their codes */
        uint32_t l[4];
        memcpy(l, h->raw_qlut, 16);
        const int64_t nch = (nb + 15) >> 4;
        const int g = (int)std::min<int64_t>((nch + 255) / 256, 256 * 32);
        uint8_t *qd = h->d_quals.as<uint8_t>() + BASE_FRONT_PAD;
        if (h->raw_qbits == 2) CRUMBLE_LAUNCH_INDEP((k_p0_quals<2>), g, 256, st, nb, h->d_qualsp.as<uint8_t>(), l[0], l[1], l[2], l[3], qd);
        else                   CRUMBLE_LAUNCH_INDEP((k_p0_quals<4>), g, 256, st, nb, h->d_qualsp.as<uint8_t>(), l[0], l[1], l[2], l[3], qd);
    }
    /* 1. classify + validate + count */
    const int nblocks = (int)((n + P0_CHUNK - 1) / P0_CHUNK);
    if ((e = h->d_inplp.reserve((size_t)(n + 1))) || (e = h->d_end.reserve((size_t)(n + 1) * 4)) ||
        (e = h->d_p0cnt.reserve((size_t)(n + 1) * sizeof(P0Cnt32))) ||
        (e = h->d_p0part.reserve((size_t)(nblocks + 2) * sizeof(P0Part))) || (e = h->d_misc.reserve(256)))
        return e;
    int *d_err = h->d_misc.as<int>() + 8, *d_nslow = d_err + 3, *d_herr = h->d_misc.as<int>() + 15;
    static const int init[8] = {0, INT_MAX, INT_MIN, 0, 0, 0, 0, 0};   /* error, min pos, max end, n_slow, n_general, -, -, hit-list check */
    HIPCHK(hipMemcpyAsync(d_err, init, sizeof init, hipMemcpyHostToDevice, st));
    P0Cnt32 *cnt = h->d_p0cnt.as<P0Cnt32>();
    P0Part *part = h->d_p0part.as<P0Part>();
    P0Part tot;
    memset(&tot, 0, sizeof tot);
    int res[3] = {0, INT_MAX, INT_MIN};
    if (n > 0) {
        CRUMBLE_LAUNCH(k_p0_classify, nblk(n, P0_CHUNK), P0_CHUNK, st, n, (int)tid, h->d_pos.as<int32_t>(),
                       h->d_flag.as<uint16_t>(), h->d_seq_off.as<int64_t>(), h->d_cig_off.as<int64_t>(),
                       h->d_cigar.as<uint32_t>(), h->d_inplp.as<uint8_t>(), h->d_end.as<int32_t>(), cnt, part, d_err, nb, nc);
        CRUMBLE_LAUNCH(k_p0_mid, 1, 1024, st, part, nblocks);
        HIPCHK(hipMemcpyAsync(h->pinned, part + nblocks, sizeof tot, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync((char *)h->pinned + 64, d_err, sizeof res, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));               /* sizes of the expanded tables (and: the caller may free its host buffers) */
        memcpy(&tot, h->pinned, sizeof tot);
        memcpy(res, (char *)h->pinned + 64, sizeof res);
        res[1] = tot.minpos; res[2] = tot.maxend;      /* the batch's extent: the scan's totals (no atomics) */
    } else HIPCHK(hipStreamSynchronize(st));
    switch (res[0]) {
    case 0: break;
    case P0_ERR_QLEN:  g_last_error = "CIGAR query length differs from l_qseq"; return CRUMBLE_EINVAL;
    case P0_ERR_COORD: g_last_error = "coordinate out of range"; return CRUMBLE_EINVAL;
    case P0_ERR_TOTALS: g_last_error = "crumble_batch: n_bases / n_cigar_ops do not match the per-read lengths"; return CRUMBLE_EINVAL;
    default:           g_last_error = "negative or oversized read / cigar length"; return CRUMBLE_EINVAL;
    }
    const int64_t tot_ncix = tot.ncixu * CIX_UNIT;
    if (tot.nseg >= (int64_t)INT_MAX - 64) { g_last_error = "too many CIGAR segments in one batch"; return CRUMBLE_EINVAL; }
    if (tot.ncixu >= (int64_t)0xfffffff0u) { g_last_error = "base->column table too large"; return CRUMBLE_EINVAL; }

    DBatch &B = h->B;
    memset(&B, 0, sizeof(B));
    const bool any = res[1] != INT_MAX;
    B.n_reads = n; B.n_bases = nb; B.n_bed = n_bed;
    B.p0 = any ? res[1] : 0;
    B.L = any ? res[2] - res[1] : 0;
    B.n_tiles = (B.L + 63) / 64;
    B.col_lo = 0; B.col_hi = B.L; B.stat_hi = B.L; B.pad1_ = 0;       /* the whole batch (crumble_hip_set_window narrows it) */
    h->windowed = false; h->phase = 0;
    B.hot_mq = h->raw_hot_mq; B.n_slots = h->raw_nslots;
    B.slot_mq = h->d_mqslot.as<uint8_t>() + 256;
    h->col_perm = h->raw_lowerr;
    if (h->opt_kcol == 1 || h->opt_kcol == 2) h->col_perm = h->opt_kcol == 2;
    h->b_kcol = h->opt_kcol; h->b_pblock = h->opt_pblock;
    if (tot.nslow > 0x7fffffffll) { g_last_error = "crumble_hip_upload: more than 2^31 bases under D/N ops"; return CRUMBLE_EINVAL; }
    h->n_segrec = tot.nseg; h->n_slow = tot.nslow;  /* the list's length: a slot per D/N op with a base to report (empty slots: read -1) */
    h->n_general_max = tot.ngen;                    /* general-CIGAR reads with bases (d_nslow[1] on the device) */

    /* 2. expanded tables */
    const size_t ns = (size_t)tot.nseg;
    const bool multi = tot.nseg > n + n / 2;        /* long reads: many ops per read */
    /* long reads through the packed replay need no base -> column table (4 B per base of every multi-op read) */
    h->batch_flat = h->opt_replay == 1;
    const bool vrec_path = multi && h->packed_ok && !h->batch_flat && tot.nseg + n < (int64_t)INT_MAX - 64;
    /* which replay the run will take (crumble_hip_run_finish): the packed kernels find a general read's columns in its op
     * table; only k_replay_flat looks them up in the base -> column table */
    const bool packed_path = h->packed_ok && !h->batch_flat && (!multi || vrec_path);
    const bool want_cix = !packed_path && !h->hparams.softclip;
    const size_t nblkb = (size_t)(nb / 64 + 1);
    const int64_t nch = (nb + 15) / 16;
    if ((e = h->d_segs.reserve((ns + 1) * sizeof(Seg))) || (e = h->d_sr.reserve((ns + 1) * sizeof(SegRec))) ||
        (e = h->d_rr.reserve((size_t)(n + 1) * sizeof(ReadRec))) || (e = h->d_seg_off.reserve((size_t)(n + 2) * 8)) ||
        (e = h->d_cix.reserve((want_cix ? (size_t)tot_ncix + 1 : 1) * 4)) ||
        (e = h->d_genlist.reserve(((size_t)tot.ngen + 1) * 4)) ||
        (e = h->d_slow_g.reserve(((size_t)tot.nslow + 1) * 8)) ||
        (e = h->d_slow_r.reserve(((size_t)tot.nslow + 1) * 4)) || (e = h->d_kpos.reserve((size_t)(n + 1) * 4)) ||
        (e = h->d_kend.reserve((size_t)(n + 1) * 4)) || (e = h->d_tile_lo.reserve((size_t)(B.n_tiles + 1) * 4)) ||
        (e = h->d_tile_hi.reserve((size_t)(B.n_tiles + 1) * 4)) || (e = h->d_stile_lo.reserve((size_t)(B.n_tiles + 1) * 4)) ||
        (e = h->d_stile_hi.reserve((size_t)(B.n_tiles + 1) * 4)) || (e = h->d_blk_read.reserve((nblkb + 1) * 4)) ||
        (e = h->d_chunk_read.reserve((size_t)(nch + 2) * 4)) || (e = h->d_tile_bed.reserve((size_t)(n_bed ? B.n_tiles + 1 : 1) * 4)) ||
        (e = h->d_vrec.reserve((vrec_path ? (size_t)(tot.nseg + n + 1) : 1) * sizeof(VRec))))
        return e;
    if (n == 0) HIPCHK(hipMemsetAsync(h->d_seg_off.p, 0, 16, st));
    if (n > 0) {
        VRec *vrp = vrec_path ? h->d_vrec.as<VRec>() : (VRec *)nullptr;
        int32_t *gl = (tot.ngen > 0 && !vrec_path) ? h->d_genlist.as<int32_t>() : (int32_t *)nullptr;     /* k_p0_cix, k_replay_reads */
#define FILL_LAUNCH(GW_) CRUMBLE_LAUNCH((k_p0_fill<GW_>), nblk(n, P0_CHUNK), P0_CHUNK, st, n, nb, B.p0, h->d_pos.as<int32_t>(), h->d_mapq.as<uint8_t>(), \
                       h->d_seq_off.as<int64_t>(), h->d_cig_off.as<int64_t>(), h->d_cigar.as<uint32_t>(), h->d_inplp.as<uint8_t>(), \
                       h->d_end.as<int32_t>(), cnt, part, nblocks, h->d_segs.as<Seg>(), h->d_sr.as<SegRec>(), h->d_rr.as<ReadRec>(), \
                       h->d_seg_off.as<int64_t>(), h->d_kpos.as<int32_t>(), h->d_kend.as<int32_t>(), h->d_slow_g.as<int64_t>(), \
                       h->d_slow_r.as<int32_t>(), d_nslow, gl, d_nslow + 1, h->d_mqslot.as<uint8_t>(), h->d_blk_read.as<int32_t>(), \
                       vrp, h->d_chunk_read.as<int32_t>(), h->hparams.min_mqual, d_err)
        if (multi) {
            FILL_LAUNCH(0);
            CRUMBLE_LAUNCH(k_p0_fill_long, nblk(n, 4), 256, st, n, nb, B.p0, h->d_pos.as<int32_t>(), h->d_mapq.as<uint8_t>(),
                           h->d_seq_off.as<int64_t>(), h->d_cig_off.as<int64_t>(), h->d_cigar.as<uint32_t>(), h->d_inplp.as<uint8_t>(),
                           h->d_end.as<int32_t>(), cnt, h->d_segs.as<Seg>(), h->d_sr.as<SegRec>(), h->d_rr.as<ReadRec>(),
                           h->d_seg_off.as<int64_t>(), h->d_slow_g.as<int64_t>(), h->d_slow_r.as<int32_t>(),
                           h->d_mqslot.as<uint8_t>(), h->d_blk_read.as<int32_t>(), vrp, h->d_chunk_read.as<int32_t>(), h->hparams.min_mqual);
        } else FILL_LAUNCH(1);
#undef FILL_LAUNCH
        if (gl && want_cix) {                                /* k_replay_flat: base -> column table of the general-CIGAR reads */
            const unsigned g = (unsigned)(tot.ngen < 4 * 4096 ? (tot.ngen + 3) / 4 : 4096);
            CRUMBLE_LAUNCH(k_p0_cix, g, 256, st, B.p0, h->d_genlist.as<int32_t>(), d_nslow + 1, h->d_rr.as<ReadRec>(),
                           h->d_seg_off.as<int64_t>(), h->d_segs.as<Seg>(), h->d_cix.as<int32_t>());
        }
        if (B.n_tiles > 0)
            CRUMBLE_LAUNCH_INDEP(k_p0_tiles, nblk(B.n_tiles, 256), 256, st, B.n_tiles, B.p0, n, h->d_kpos.as<int32_t>(), h->d_kend.as<int32_t>(),
                                 h->d_seg_off.as<int64_t>(), h->d_sr.as<SegRec>(), h->d_tile_lo.as<int32_t>(), h->d_tile_hi.as<int32_t>(),
                                 h->d_stile_lo.as<int32_t>(), h->d_stile_hi.as<int32_t>(), n_bed, h->d_bed_pmax.as<int32_t>(),
                                 h->d_tile_bed.as<int32_t>(), d_err);
        if (B.n_tiles > 0 && multi) {                       /* the records that reach each tile, in file order (k_columns2<MULTI>) */
            const int rch = (int)((n + P0_CHUNK - 1) / P0_CHUNK), tch = (int)((B.n_tiles + P0_CHUNK - 1) / P0_CHUNK);
            if ((e = h->d_rtoff.reserve((size_t)(n + 2) * 8)) || (e = h->d_tpart.reserve((size_t)((rch > tch ? rch : tch) + 2) * 8)) ||
                (e = h->d_hitoff.reserve((size_t)(B.n_tiles + 2) * 8)))
                return e;
            int64_t *rto = h->d_rtoff.as<int64_t>(), *tp = h->d_tpart.as<int64_t>(), *ho = h->d_hitoff.as<int64_t>();
            CRUMBLE_LAUNCH_INDEP(k_p0_rtcount, nblk(n, 256), 256, st, n, B.p0, h->d_rr.as<ReadRec>(), h->d_seg_off.as<int64_t>(), rto);
            CRUMBLE_LAUNCH(k_i64scan_reduce, rch, 256, st, n, rto, tp, rch);
            CRUMBLE_LAUNCH(k_i64scan_partials, 1, 1024, st, tp, rch);
            CRUMBLE_LAUNCH(k_i64scan_apply, rch, 256, st, n, rto, tp, rch);
            HIPCHK(hipMemcpyAsync(h->pinned, rto + n, 8, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            int64_t npairs = 0;
            memcpy(&npairs, h->pinned, 8);
            if ((e = h->d_first.reserve((size_t)(npairs + 64) * 4))) return e;
            const uint32_t *first = h->d_first.as<uint32_t>();
            CRUMBLE_LAUNCH_INDEP(k_p0_first, nblk(n, 4), 256, st, n, B.p0, h->d_rr.as<ReadRec>(), h->d_seg_off.as<int64_t>(), h->d_sr.as<SegRec>(),
                                 rto, h->d_first.as<uint32_t>());
            const unsigned hgrid = nblk((int64_t)B.n_tiles * P0_HIT_GROUP, 256);
            CRUMBLE_LAUNCH(k_p0_hitcount, hgrid, 256, st, B.n_tiles, B.p0, h->d_tile_lo.as<int32_t>(), h->d_tile_hi.as<int32_t>(),
                           h->d_rr.as<ReadRec>(), h->d_seg_off.as<int64_t>(), rto, first, ho);
            CRUMBLE_LAUNCH(k_i64scan_reduce, tch, 256, st, (int64_t)B.n_tiles, ho, tp, tch);
            CRUMBLE_LAUNCH(k_i64scan_partials, 1, 1024, st, tp, tch);
            CRUMBLE_LAUNCH(k_i64scan_apply, tch, 256, st, (int64_t)B.n_tiles, ho, tp, tch);
            HIPCHK(hipMemcpyAsync(h->pinned, ho + B.n_tiles, 8, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            int64_t nhits = 0;
            memcpy(&nhits, h->pinned, 8);
            if (nhits >= INT_MAX) { g_last_error = "more than 2^31 (record, tile) pairs in one batch"; return CRUMBLE_EINVAL; }
            if ((e = h->d_hits.reserve((size_t)(nhits + 64) * 4))) return e;
            CRUMBLE_LAUNCH(k_p0_hitfill, hgrid, 256, st, B.n_tiles, B.p0, h->d_tile_lo.as<int32_t>(), h->d_tile_hi.as<int32_t>(),
                           h->d_rr.as<ReadRec>(), h->d_seg_off.as<int64_t>(), rto, first, ho, h->d_hits.as<int32_t>(), d_herr);
            h->have_hits = true;
        }
        if (nb > 0) {                                       /* the bases as the kernels read them; chunk -> read table (k_replay_fast) */
            const unsigned g = nblk(nb, P0B_BASES);
#define BASES_LAUNCH(P_, O_) CRUMBLE_LAUNCH((k_p0_bases<P_, O_>), g, P0B_THREADS, st, nb, n, h->d_seq_off.as<int64_t>(), h->d_blk_read.as<int32_t>(), \
                                            packed4 ? h->d_bases4.as<uint8_t>() : (const uint8_t *)nullptr, h->d_bases.as<uint8_t>() + BASE_FRONT_PAD, \
                                            h->d_chunk_read.as<int32_t>())
            if (packed4) { if (vrec_path) BASES_LAUNCH(true, false); else BASES_LAUNCH(true, true); }
            else         { if (vrec_path) BASES_LAUNCH(false, false); else BASES_LAUNCH(false, true); }
#undef BASES_LAUNCH
            if (vrec_path) { B.n_vrec = tot.nseg + n; h->have_vrec = true; }
        }
    }
    B.pos = h->d_pos.as<int32_t>(); B.end = h->d_end.as<int32_t>();
    B.mapq = h->d_mapq.as<uint8_t>(); B.inplp = h->d_inplp.as<uint8_t>();
    B.seq_off = h->d_seq_off.as<int64_t>(); B.cig_off = h->d_cig_off.as<int64_t>();
    B.seg_off = h->d_seg_off.as<int64_t>(); B.cigar = h->d_cigar.as<uint32_t>();
    B.segs = h->d_segs.as<Seg>(); B.rr = h->d_rr.as<ReadRec>(); B.cix = h->d_cix.as<int32_t>();
    B.vr = h->d_vrec.as<VRec>(); if (!h->have_vrec) B.n_vrec = 0;
    B.sr = h->d_sr.as<SegRec>(); B.stile_lo = h->d_stile_lo.as<int32_t>(); B.stile_hi = h->d_stile_hi.as<int32_t>();
    B.unit_tiles = KCOL2_UTILES; B.n_units = (B.n_tiles + KCOL2_UTILES - 1) / KCOL2_UTILES;
    B.hit_off = h->have_hits ? h->d_hitoff.as<int64_t>() : nullptr; B.hits = h->have_hits ? h->d_hits.as<int32_t>() : nullptr;
    B.bases = h->d_bases.as<uint8_t>() + BASE_FRONT_PAD; B.quals = h->d_quals.as<uint8_t>() + BASE_FRONT_PAD;
    B.tile_lo = h->d_tile_lo.as<int32_t>(); B.tile_hi = h->d_tile_hi.as<int32_t>();
    B.bed_start = h->d_bed_start.as<int32_t>(); B.bed_end = h->d_bed_end.as<int32_t>();
    B.bed_pmax = h->d_bed_pmax.as<int32_t>(); B.tile_bed = h->d_tile_bed.as<int32_t>();

    const size_t L = (size_t)(B.L > 0 ? B.L : 1);
    const int nsc = (int)((L + SCAN_CHUNK_OF(L) - 1) / SCAN_CHUNK_OF(L));
    if ((e = h->d_colrec.reserve(L * 4)) || (e = h->d_nplp.reserve(L * 4)) || (e = h->d_score.reserve(L * 4)) ||
        (e = h->d_trigcnt.reserve(L * 4)) || (e = h->d_dpre.reserve(L * 8)) || (e = h->d_col_ord.reserve(L * 4)) ||
        (e = h->d_trig_ord.reserve(L * 4)) || (e = h->d_item_off.reserve(L * 4)) || (e = h->d_ins_ord.reserve(L * 4)) ||
        (e = h->d_pos_mn2.reserve(L * 4)) || (e = h->d_ht.reserve(L * 4)) || (e = h->d_partial.reserve((size_t)(nsc + 2) * sizeof(ScanVal))) ||
        (e = h->d_epochs.reserve(MAX_EPOCHS * sizeof(Epoch))) ||
        (e = h->d_out.reserve((size_t)nb + 64)) || (e = h->d_rflags.reserve((size_t)(n ? n : 1))) ||
        (e = h->d_keepf.reserve((size_t)(n ? n : 1) * 2)) || (e = h->d_kept.reserve((size_t)(n ? n : 1) * 4)))
        return e;
    h->d_b15f.view(h->d_keepf.as<uint8_t>() + (n ? n : 1), (size_t)(n ? n : 1));      /* (one memset clears both) */
    /* the late errors: what k_p0_fill / k_p0_tiles / k_p0_hitfill found */
    HIPCHK(hipMemcpyAsync((char *)h->pinned + 256, d_err, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync((char *)h->pinned + 260, d_herr, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipEventRecord(h->ev_up[2], st));
    h->have_batch = true;
    h->late_pending = true;
    if (wait) {
        HIPCHK(hipStreamSynchronize(st));
        HIPCHK(hipGetLastError());
        h->late_pending = false;
        HIPCHK(hipEventElapsedTime(&h->up_h2d_ms, h->ev_up[0], h->ev_up[1]));
        HIPCHK(hipEventElapsedTime(&h->up_p0_ms, h->ev_up[1], h->ev_up[2]));
        if ((e = expand_late_errors(h))) { h->have_batch = false; return e; }
    }
    return CRUMBLE_OK;
}

extern "C" int crumble_hip_upload(crumble_hip *h, const crumble_batch *b)
{
    if (!h || !b || b->n_reads < 0) return CRUMBLE_EINVAL;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipEventRecord(h->ev_up[0], h->stream));
    int e;
    if ((e = copy_raw(h, b))) return e;
    return expand(h, true);
}

/* the copy half alone: returns when the raw arrays are on the device (the link is free for the next batch's copy while
 * this one's expansion -- crumble_hip_expand -- and run wait their turn on the GPU) */
extern "C" int crumble_hip_upload_raw(crumble_hip *h, const crumble_batch *b)
{
    if (!h || !b || b->n_reads < 0) return CRUMBLE_EINVAL;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipEventRecord(h->ev_up[0], h->stream));
    int e;
    if ((e = copy_raw(h, b))) return e;
    HIPCHK(hipEventSynchronize(h->ev_up[1]));
    HIPCHK(hipEventElapsedTime(&h->up_h2d_ms, h->ev_up[0], h->ev_up[1]));
    h->up_p0_ms = 0;
    return CRUMBLE_OK;
}

extern "C" int crumble_hip_expand(crumble_hip *h)
{
    if (!h || !h->have_raw) { g_last_error = "crumble_hip_expand: no raw batch on the device (crumble_hip_upload first)"; return CRUMBLE_EINVAL; }
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipEventRecord(h->ev_up[0], h->stream));
    return expand(h, true);
}

/* -------------------------------------------------------------------- run */

#define STAGE(name_, bytes_) do { if (h->opt_stages && h->n_stages < MAX_STAGES) { \
    HIPCHK(hipEventRecord(h->ev[h->n_stages], st)); \
    h->stage_name[h->n_stages] = name_; h->stage_bytes[h->n_stages] = (double)(bytes_); h->n_stages++; } } while (0)

/* P-block over a flat quality array (snp:803-834): stop masks, then one lane per 64-base word
 * replays the segments that start in its word (kernels.h). */
static int launch_pblock(crumble_hip *h, hipStream_t st, int64_t n_reads, int64_t nb, const int64_t *seq_off,
                         uint8_t *out, const uint8_t *rflags, const int32_t *blk_read, int mode)
{
    if (nb <= 0 || n_reads <= 0) return CRUMBLE_OK;
    const DParams *P = h->d_params.as<DParams>();
    /* The word-parallel kernels (kernels.h: k_pb_words / k_pb_chains / k_pb_fillw) for every batch: in their second form
     * they beat the per-read walk on short reads too (chr20 -9: 2.26 ms against 4.16 for k_pb_chunkmm / k_pb_reads /
     * k_pb_fill, the 300x panel 0.73 against 0.78) as they did on long reads (config 4: 5.4 ms against 13.6 per read and
     * 5.8 for k_pb_segments).  Option "pblock" = reads | segments | words forces one (tuning / tests: all give the same
     * bytes). */
    const bool per_read = mode == 1;
    const bool words = mode == 0 || mode == 3;
    const int64_t nwords = (nb + 63) / 64;
    int e;
    if (per_read && h->hparams.any_preserve) {          /* -k/-K: the byte-wise kernel knows the skip rule */
        CRUMBLE_LAUNCH_INDEP(k_pblock, nblk(n_reads, 64), 64, st, n_reads, seq_off, out, rflags, P);
        return CRUMBLE_OK;
    }
    if (per_read) {     /* chunk min/max (coalesced) -> runs per read -> fill of the words whose bytes change */
        const int64_t nch = (nb + 15) / 16;
        if ((e = h->d_pbstop.reserve((size_t)(nwords + 1) * 8)) || (e = h->d_pbruns.reserve((size_t)(nwords + 1) * 8)) ||
            (e = h->d_pbval.reserve((size_t)nb + 64)) || (e = h->d_pbwmm.reserve((size_t)(nch + 1) * 2)) ||
            (e = h->d_pbdirty.reserve((size_t)nwords + 64)))
            return e;
        u64 *stop = h->d_pbstop.as<u64>(), *runs = h->d_pbruns.as<u64>();
        HIPCHK(hipMemsetAsync(stop, 0, (size_t)nwords * 8, st));
        HIPCHK(hipMemsetAsync(runs, 0, (size_t)nwords * 8, st));
        HIPCHK(hipMemsetAsync(h->d_pbdirty.p, 0, (size_t)nwords, st));
        CRUMBLE_LAUNCH_INDEP(k_pb_chunkmm, nblk(nch, 256), 256, st, nb, out, h->d_pbwmm.as<uint16_t>());
        CRUMBLE_LAUNCH_INDEP(k_pb_readstarts, nblk(n_reads, 256), 256, st, n_reads, seq_off, stop, (u64 *)nullptr);
        CRUMBLE_LAUNCH_INDEP(k_pb_reads, nblk(n_reads, 256), 256, st, n_reads, seq_off, out, rflags,
                             h->d_pbwmm.as<uint16_t>(), runs, h->d_pbval.as<uint8_t>(), h->d_pbdirty.as<uint8_t>(), P);
        CRUMBLE_LAUNCH_INDEP(k_pb_fill, nblk(nch, 256), 256, st, nb, out, stop, runs, h->d_pbval.as<uint8_t>(),
                             h->d_pbdirty.as<uint8_t>());
        return CRUMBLE_OK;
    }
    if ((e = h->d_pbstop.reserve((size_t)(nwords + 1) * 8)) || (e = h->d_pbrs.reserve((size_t)(nwords + 1) * 8)) ||
        (e = h->d_pbruns.reserve((size_t)(nwords + 1) * 8)) || (e = h->d_pbval.reserve((size_t)nb + 64)) ||
        (e = h->d_pbwmm.reserve((size_t)(nwords + 1) * 2)) || (e = h->d_pbdirty.reserve((size_t)nwords + 64)))
        return e;
    u64 *stop = h->d_pbstop.as<u64>(), *rs = h->d_pbrs.as<u64>(), *runs = h->d_pbruns.as<u64>();
    uint8_t *val = h->d_pbval.as<uint8_t>();
    if (words) {
        /* read starts, then words (stop masks and every run inside a word), chains (runs that cross words), fill */
        if ((e = h->d_pbcarry.reserve((size_t)(nwords + 1) * 8)) || (e = h->d_pbchg.reserve((size_t)(nwords + 1) * 8)) ||
            (e = h->d_pbdirty.reserve((size_t)nwords * 4 + 64)))
            return e;
        u64 *carry = h->d_pbcarry.as<u64>(), *chg = h->d_pbchg.as<u64>();
        uint16_t *mm = h->d_pbwmm.as<uint16_t>();
        uint32_t *dirty = h->d_pbdirty.as<uint32_t>();
        HIPCHK(hipMemsetAsync(dirty, 0, (size_t)nwords * 4, st));
        HIPCHK(hipMemsetAsync(rs, 0, (size_t)nwords * 8, st));
        CRUMBLE_LAUNCH_INDEP(k_pb_readstarts, nblk(n_reads, 256), 256, st, n_reads, seq_off, (u64 *)nullptr, rs);
        CRUMBLE_LAUNCH(k_pb_words, nblk(nwords, 256), 256, st, nb, n_reads, seq_off, out, rflags, blk_read, rs, stop, runs, chg, val, carry, mm, dirty, P);
        CRUMBLE_LAUNCH_INDEP(k_pb_chains, nblk(nwords, 256), 256, st, nb, out, stop, rs, runs, chg, val, carry, mm, dirty, P);
        CRUMBLE_LAUNCH(k_pb_fillw, nblk(nwords, 256), 256, st, nb, out, stop, runs, chg, val, dirty);
        return CRUMBLE_OK;
    }
    HIPCHK(hipMemsetAsync(h->d_pbdirty.p, 0, (size_t)nwords, st));
    CRUMBLE_LAUNCH_INDEP(k_pb_marks, nblk(nwords, 256), 256, st, nb, out, P, stop, rs, runs, h->d_pbwmm.as<uint16_t>());
    CRUMBLE_LAUNCH_INDEP(k_pb_readstarts, nblk(n_reads, 256), 256, st, n_reads, seq_off, stop, rs);
    CRUMBLE_LAUNCH_INDEP(k_pb_segments, nblk(nwords, 256), 256, st, nb, n_reads, seq_off, out, rflags, blk_read, stop, rs, runs, val, h->d_pbwmm.as<uint16_t>(), h->d_pbdirty.as<uint8_t>(), P);
    CRUMBLE_LAUNCH_INDEP(k_pb_fill, nblk((nb + 15) / 16, 256), 256, st, nb, out, stop, runs, val, h->d_pbdirty.as<uint8_t>());
    return CRUMBLE_OK;
}

extern "C" void crumble_carry_init(crumble_carry *c)
{
    if (!c) return;
    c->total_col = c->total_depth = 0;
    c->min_pos = INT_MAX; c->max_pos = 0; c->min_pos2 = INT_MAX; c->max_pos2 = 0;
}

extern "C" int crumble_hip_set_window(crumble_hip *h, const crumble_window *w)
{
    if (!h || !h->have_batch) { g_last_error = "crumble_hip_set_window: no batch uploaded"; return CRUMBLE_EINVAL; }
    DBatch &B = h->B;
    h->phase = 0; h->have_run = false;
    if (!w) { B.col_lo = 0; B.col_hi = B.L; B.stat_hi = B.L; h->windowed = false; return CRUMBLE_OK; }
    if (w->col_start > w->stat_end || w->stat_end > w->col_end || w->read_first < 0 || w->read_first > w->read_end ||
        w->read_end > h->n_reads) { g_last_error = "crumble_hip_set_window: inconsistent window"; return CRUMBLE_EINVAL; }
    /* positions -> column indices of this batch, clipped to what it covers */
    const int64_t lo = (int64_t)w->col_start - B.p0, hi = (int64_t)w->col_end - B.p0, sh = (int64_t)w->stat_end - B.p0;
    B.col_lo = (int32_t)(lo < 0 ? 0 : lo > B.L ? B.L : lo);
    B.col_hi = (int32_t)(hi < 0 ? 0 : hi > B.L ? B.L : hi);
    B.stat_hi = (int32_t)(sh < B.col_lo ? B.col_lo : sh > B.col_hi ? B.col_hi : sh);
    h->win = *w; h->windowed = true;
    /* byte range of the owned reads (two 8-byte reads of the offset table) */
    int64_t o2[2] = {0, 0};
    HIPCHK(hipSetDevice(h->device));
    if (h->n_reads > 0) {
        HIPCHK(copy_sync(h->stream, &o2[0], h->d_seq_off.as<int64_t>() + w->read_first, 8, hipMemcpyDeviceToHost));
        HIPCHK(copy_sync(h->stream, &o2[1], h->d_seq_off.as<int64_t>() + w->read_end, 8, hipMemcpyDeviceToHost));
    }
    h->own_b_lo = o2[0]; h->own_b_hi = o2[1];
    return CRUMBLE_OK;
}

/* The run of a batch is five phases.  crumble_hip_run() does them back to back for a whole contig; a contig cut into
 * windows (include/crumble_hip.h, crumble_hip_set_window) runs phases 1, 3 and 5 of all its windows in parallel -- one
 * engine per device -- and only the two tiny chained phases in window order: 2 hands the running depth average on
 * (snp:1490-1491,1684-1687), 4 the STR window state (snp:1508-1511,1741-1755). */
/* what every phase needs from the handle.  Run flags, behind the batch statistics (d_overflow): [0] overflow flag,
 * [1] n_epochs, [2] n_kept, [3] k_columns work counter, [4] a quality >= 128 was seen (k_stats), [5] fast path aborted,
 * [8..15] the scan's totals (fast path).  d_misc: [8..12] upload, [15] hit-list check, [16..19] depth state out,
 * [24..27] window state out */
#define RUN_LOCALS \
    HIPCHK(hipSetDevice(h->device)); \
    hipStream_t st = h->stream; \
    const DBatch &B = h->B; \
    const DParams *P = h->d_params.as<DParams>(); \
    const DTables *T = h->d_tables.as<DTables>(); \
    u64 *stats = h->d_batch_stats.as<u64>(); \
    int *d_overflow = (int *)(stats + CRUMBLE_ST_N);        /* the run's flags live behind the batch statistics: one memset */ \
    int *d_nep = d_overflow + 1; \
    const bool spec = h->spec; \
    const int *spec_abort = spec ? d_overflow + 5 : (const int *)nullptr; \
    const int L = B.L; \
    const int64_t n = B.n_reads, nb = B.n_bases; \
    uint32_t *colrec = h->d_colrec.as<uint32_t>(); \
    int32_t *nplp = h->d_nplp.as<int32_t>(), *score = h->d_score.as<int32_t>(); \
    int32_t *trigcnt = h->d_trigcnt.as<int32_t>(); \
    const int scan_chunk = SCAN_CHUNK_OF(L); \
    const int nchunks = (L + scan_chunk - 1) / scan_chunk; \
    ScanVal *partial = h->d_partial.as<ScanVal>(); \
    const int n_trig = spec ? h->cap_trig : h->totals.trigcols, n_items = spec ? h->cap_items : h->totals.items, n_ins = spec ? h->cap_ins : h->totals.ins; \
    const int *nd_trig = spec ? &partial[nchunks].trigcols : (const int *)nullptr, *nd_items = spec ? &partial[nchunks].items : (const int *)nullptr, \
              *nd_ins = spec ? &partial[nchunks].ins : (const int *)nullptr; \
    /* what the list kernels' grids are sized for: the exact lengths, or (fast path) the capacities bounded by what a \
     * batch of this size can produce -- blocks beyond the real length leave at once, and a small contig behind a large \
     * one does not launch the large one's grids */ \
    const int gn_trig = spec ? (int)std::min<int64_t>(n_trig, L) : n_trig, gn_ins = spec ? (int)std::min<int64_t>(n_ins, L) : n_ins, \
              gn_items = spec ? (int)std::min<int64_t>(n_items, nb + L) : n_items; \
    const unsigned gcap = spec ? 65536u : 0x7fffffffu;      /* (fast path: a workgroup that finds nothing costs ~0.2 ns; the kernels stride) */ \
    int32_t *trigcol_pos = h->d_trigcol_pos.as<int32_t>(), *trigcol_off = h->d_trigcol_off.as<int32_t>(); \
    int32_t *inscol_pos = h->d_inscol_pos.as<int32_t>(); \
    TrigItem *items = h->d_items.as<TrigItem>(); \
    (void)P; (void)T; (void)stats; (void)d_nep; (void)L; (void)n; (void)nb; (void)colrec; (void)nplp; (void)score; (void)trigcnt; \
    (void)nchunks; (void)scan_chunk; (void)partial; (void)n_trig; (void)n_items; (void)n_ins; (void)trigcol_pos; (void)trigcol_off; (void)inscol_pos; (void)items; \
    (void)spec_abort; (void)nd_trig; (void)nd_items; (void)nd_ins; (void)gn_trig; (void)gn_ins; (void)gn_items; (void)gcap;
static const crumble_carry CARRY0 = {0, 0, INT_MAX, 0, INT_MAX, 0};

extern "C" int crumble_hip_run_columns(crumble_hip *h)
{
    if (!h || !h->have_batch) { g_last_error = "no batch uploaded"; return CRUMBLE_EINVAL; }
    RUN_LOCALS
    h->n_stages = 0; h->have_run = false; h->fold_small = false; h->have_coded = false;
    memset(&h->totals, 0, sizeof(h->totals));
    HIPCHK(hipMemsetAsync(stats, 0, CRUMBLE_ST_N * 8 + 128, st));     /* the counters, the run flags, k_columns2's eight unit counters */

    if (L > 0) {
        uint32_t *ht = h->d_ht.as<uint32_t>();
        const bool modeB = h->hparams.min_qual_B != 0;
        const unsigned kw = KCOL_THREADS / 64;
        unsigned grid = nblk(B.n_tiles, kw);
        const unsigned pgrid = (h->persistent_blocks * 4 * (modeB && h->hparams.min_qual_A ? 3 : KCOL_WAVES) / 4 + kw - 1) / kw;
        if (grid > pgrid) grid = pgrid;                 /* persistent: grid-stride over tiles */
        const bool modeA = h->hparams.min_qual_A != 0;
        const bool multi = h->have_hits;                                  /* long reads: many ops per read (the upload listed every tile's records) */
        DBatch B1 = B;                                  /* k_columns gives a wave a whole unit: smaller ones */
        B1.unit_tiles = KCOL1_UTILES; B1.n_units = (B.n_tiles + KCOL1_UTILES - 1) / KCOL1_UTILES;
#define KCOL_LAUNCH2(M, MU, PE) CRUMBLE_LAUNCH((k_columns<M, MU, PE>), grid, KCOL_THREADS, st, B1, P, T, colrec, nplp, score, trigcnt, ht, stats, (unsigned *)(d_overflow + 3))
#define KCOL_LAUNCH(M) do { if (multi) { if (h->col_perm) KCOL_LAUNCH2(M, true, true); else KCOL_LAUNCH2(M, true, false); } \
                            else       { if (h->col_perm) KCOL_LAUNCH2(M, false, true); else KCOL_LAUNCH2(M, false, false); } } while (0)
        /* k_columns2 (kcol2.h) covers one active consensus; both at once (-q with -Q) and a disabled consensus
         * (min_qual 0) stay with k_columns */
        const bool use_cons = modeB ? true : h->hparams.min_qual_A != 0;
        const bool v2 = (h->b_kcol == 0 || h->b_kcol == 3) && !(modeA && modeB) && use_cons;
        if (!v2) {                                          /* k_columns2 counts heads / tails while it stages its records */
            STAGE("k_headtail", 0);
            HIPCHK(hipMemsetAsync(ht, 0, (size_t)L * 4, st));
            CRUMBLE_LAUNCH_INDEP(k_headtail, nblk(n, 256), 256, st, B, ht);
        }
        STAGE(v2 ? "k_columns2" : "k_columns", 2.0 * nb + 32.0 * n + 8.0 * h->n_cigar);
        if (v2) {
            unsigned g2 = (h->persistent_blocks + 3) / 4;       /* one workgroup of KC2_WAVES waves per CU */
            if (g2 > nblk(B.n_tiles, KC2_WAVES)) g2 = nblk(B.n_tiles, KC2_WAVES);
#define KCOL2_LAUNCH4(MQ_, AP_, MU_, W_) CRUMBLE_LAUNCH((k_columns2<MQ_, AP_, MU_, W_>), g2, KC2_THREADS, st, B, P, T, colrec, nplp, score, trigcnt, ht, stats, (unsigned *)(d_overflow + 16))
#define KCOL2_LAUNCH3(MQ_, AP_, MU_) do { if (h->windowed) KCOL2_LAUNCH4(MQ_, AP_, MU_, true); else KCOL2_LAUNCH4(MQ_, AP_, MU_, false); } while (0)
#define KCOL2_LAUNCH(MQ_, AP_) do { if (multi) KCOL2_LAUNCH3(MQ_, AP_, true); else KCOL2_LAUNCH3(MQ_, AP_, false); } while (0)
            if (modeB) { if (h->hparams.any_preserve) KCOL2_LAUNCH(true, true); else KCOL2_LAUNCH(true, false); }
            else       { if (h->hparams.any_preserve) KCOL2_LAUNCH(false, true); else KCOL2_LAUNCH(false, false); }
#undef KCOL2_LAUNCH
#undef KCOL2_LAUNCH3
#undef KCOL2_LAUNCH4
        }
        else if (modeA && modeB) KCOL_LAUNCH(2);
        else if (modeB)     KCOL_LAUNCH(1);
        else                KCOL_LAUNCH(0);
#undef KCOL_LAUNCH
#undef KCOL_LAUNCH2

        STAGE("scan", 0);
        CRUMBLE_LAUNCH(k_scan_reduce, nchunks, 256, st, L, colrec, nplp, trigcnt, partial, nchunks, scan_chunk);
        if (spec) {
            /* capacities from the batch's shape (measured: trigger columns 0.03-1 % of the columns, triggers 0.003-0.07 %
             * of the bases over the four BASELINE configurations), never below what an earlier batch needed */
            const int64_t ct = std::max<int64_t>(4096, (int64_t)L / 32), ci = std::max<int64_t>(16384, nb / 128);
            if (h->cap_trig < ct) h->cap_trig = (int)std::min<int64_t>(ct, INT_MAX / 2);
            if (h->cap_items < ci) h->cap_items = (int)std::min<int64_t>(ci, INT_MAX / 2);
            if (h->cap_ins < h->cap_trig) h->cap_ins = h->cap_trig;
            if (h->opt_caps) h->cap_trig = h->cap_items = h->cap_ins = h->opt_caps;
            CRUMBLE_LAUNCH(k_scan_partials, 1, 1024, st, partial, nchunks, h->cap_trig, h->cap_items, h->cap_ins, d_overflow + 5, (ScanVal *)(d_overflow + 8));
            h->scan_fused = true;                   /* (a too-deep column aborts the fast path) */
        } else {
            CRUMBLE_LAUNCH(k_scan_partials, 1, 1024, st, partial, nchunks, 0, 0, 0, (int *)nullptr, (ScanVal *)nullptr);
            HIPCHK(hipMemcpyAsync(h->pinned, partial + nchunks, sizeof(ScanVal), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));           /* sizes of the work lists */
            memcpy(&h->totals, h->pinned, sizeof(ScanVal));
            /* a whole contig without a too-deep column needs no per-column prefix arrays (kernels.h, k_scan_overdepth) */
            h->scan_fused = !h->windowed && h->totals.toodeep == 0;
            if (!h->scan_fused)
                CRUMBLE_LAUNCH(k_scan_apply, nchunks, 256, st, L, colrec, nplp, trigcnt, partial, nchunks,
                                     h->d_dpre.as<int64_t>(), h->d_col_ord.as<int32_t>(), h->d_trig_ord.as<int32_t>(),
                                     h->d_item_off.as<int32_t>(), h->d_ins_ord.as<int32_t>(), scan_chunk);
        }
        const int nt = spec ? h->cap_trig : h->totals.trigcols, nit = spec ? h->cap_items : h->totals.items, ni = spec ? h->cap_ins : h->totals.ins;
        int e;
        if ((e = h->d_trigcol_pos.reserve((size_t)(nt + 1) * 4)) || (e = h->d_trigcol_off.reserve((size_t)(nt + 1) * 4)) ||
            (e = h->d_inscol_pos.reserve((size_t)(ni + 1) * 4)) || (e = h->d_items.reserve((size_t)(nit + 1) * sizeof(TrigItem))) ||
            (e = h->d_colhi.reserve((size_t)(nt + 1) * 4)) || (e = h->d_ub.reserve((size_t)(nt + 1) * 4)) ||
            (e = h->d_trig_mx2.reserve((size_t)(nt + 1) * 4)) || (e = h->d_pmax_partial.reserve((size_t)(nt / 256 + 2) * 4)))
            return e;

    } else memset(&h->totals, 0, sizeof(h->totals));
    h->phase = 1;
    return CRUMBLE_OK;
}

extern "C" int crumble_hip_run_depth(crumble_hip *h, const crumble_carry *in, crumble_carry *out)
{
    if (!h || !h->have_batch || h->phase != 1) { g_last_error = "crumble_hip_run_depth: run the columns first"; return CRUMBLE_EINVAL; }
    RUN_LOCALS
    const crumble_carry cin = in ? *in : CARRY0;
    /* (total_col passes 2^20 + 1 only while the columns at the halving point are too deep: snp:1493-1500 skips the halving) */
    if (cin.total_col < 0 || cin.total_col > ((int64_t)1 << 40) || cin.total_depth < 0) { g_last_error = "crumble_hip_run_depth: impossible carried state"; return CRUMBLE_EINVAL; }
    const bool late = cin.total_col > 1048576;         /* the halving is overdue: the general epoch walk */
    h->carry_in.total_col = cin.total_col; h->carry_in.total_depth = cin.total_depth;
    if (out) { out->total_col = cin.total_col; out->total_depth = cin.total_depth; }
    if (L > 0 && h->scan_fused && (out || late)) {    /* the state at a cut is read from the prefix arrays */
        CRUMBLE_LAUNCH(k_scan_apply, nchunks, 256, st, L, colrec, nplp, trigcnt, partial, nchunks,
                             h->d_dpre.as<int64_t>(), h->d_col_ord.as<int32_t>(), h->d_trig_ord.as<int32_t>(),
                             h->d_item_off.as<int32_t>(), h->d_ins_ord.as<int32_t>(), scan_chunk);
        h->scan_fused = false;
    }
    if (L > 0 && h->scan_fused) {
        /* (fast path: the number of counted columns is on the device; at most L of them) */
        const int64_t ncols = spec ? (int64_t)L : h->totals.cols, first = 1048577 - cin.total_col;
        int n = 0;
        if (ncols > 0) n = 1 + (int)((ncols > first) ? (ncols - first + 524288) / 524289 : 0);
        if (n > MAX_EPOCHS - 1) n = MAX_EPOCHS - 1;
        if (n > 1)
            CRUMBLE_LAUNCH(k_epochs_find, n - 1, 64, st, L, colrec, nplp, partial, nchunks, h->d_epochs.as<Epoch>(), (int64_t)cin.total_col, scan_chunk);
        CRUMBLE_LAUNCH_INDEP(k_epochs_td, 1, 1, st, h->d_epochs.as<Epoch>(), n, d_nep, (int64_t)cin.total_col, (int64_t)cin.total_depth,
                             spec ? partial + nchunks : (const ScanVal *)nullptr);
    } else
    if (L > 0) {
        if (h->totals.toodeep == 0 && !late)
            CRUMBLE_LAUNCH(k_epochs_par, 1, 256, st, L, colrec, h->d_col_ord.as<int32_t>(), h->d_dpre.as<int64_t>(),
                           partial + nchunks, h->d_epochs.as<Epoch>(), d_nep, (int64_t)cin.total_col, (int64_t)cin.total_depth);
        else
            CRUMBLE_LAUNCH_INDEP(k_epochs, 1, 1, st, L, colrec, nplp, h->d_col_ord.as<int32_t>(), h->d_dpre.as<int64_t>(),
                                 partial + nchunks, h->d_epochs.as<Epoch>(), d_nep, (int64_t)cin.total_col, (int64_t)cin.total_depth);
        if (out) {
            int64_t *d_state = (int64_t *)(h->d_misc.as<int>() + 16);
            CRUMBLE_LAUNCH_INDEP(k_depth_state_at, 1, 1, st, L, B.stat_hi, colrec, nplp, h->d_col_ord.as<int32_t>(), h->d_dpre.as<int64_t>(),
                                 partial + nchunks, h->d_epochs.as<Epoch>(), d_nep, (int64_t)cin.total_col, (int64_t)cin.total_depth, d_state);
            int64_t res[2];
            HIPCHK(hipMemcpyAsync(res, d_state, sizeof res, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            out->total_col = res[0]; out->total_depth = res[1];
        }
    }
    h->phase = 2;
    return CRUMBLE_OK;
}

extern "C" int crumble_hip_run_triggers(crumble_hip *h)
{
    if (!h || !h->have_batch || h->phase != 2) { g_last_error = "crumble_hip_run_triggers: run the depth phase first"; return CRUMBLE_EINVAL; }
    RUN_LOCALS
    if (L > 0) {
        STAGE("k_overdepth", 0);
        if (h->scan_fused)
            CRUMBLE_LAUNCH(k_scan_overdepth, nchunks, 256, st, L, colrec, nplp, trigcnt, partial, h->d_epochs.as<Epoch>(), d_nep, P,
                           trigcol_pos, trigcol_off, inscol_pos, stats, B.stat_hi, spec_abort, scan_chunk);
        else
        CRUMBLE_LAUNCH_INDEP(k_overdepth, nblk((L + 3) / 4, 256), 256, st, L, colrec, nplp, h->d_col_ord.as<int32_t>(),
                             h->d_dpre.as<int64_t>(), h->d_trig_ord.as<int32_t>(), h->d_item_off.as<int32_t>(),
                             h->d_ins_ord.as<int32_t>(), h->d_epochs.as<Epoch>(), d_nep, P,
                             trigcol_pos, trigcol_off, inscol_pos, stats, B.stat_hi);
        /* (fast path: the lists' lengths are on the device; grids sized for the capacities, bounded, striding) */
        if (n_ins > 0) {
            STAGE("k_inshist", 0);
            {   /* lanes per column: 16 when a column has a few dozen candidate reads, the whole wave on deep pileups
                 * (mean depth of the covered columns; fast path: the kernel takes it from the scan's totals on the device) */
                const int64_t mean_depth = spec ? 0 : (h->totals.cols > 0 ? h->totals.depth / h->totals.cols : 0);
                const int gshift = mean_depth < 40 ? 4 : 6;
                const unsigned g = std::min(gcap, (unsigned)((gn_ins + (64 >> gshift) - 1) / (64 >> gshift)));
                CRUMBLE_LAUNCH(k_inshist, g, 64, st, B, P, n_ins, nd_ins, spec_abort, inscol_pos, colrec, nplp, stats, gshift,
                               spec ? (const ScanVal *)&partial[nchunks] : (const ScanVal *)nullptr);
            }
        }
        if (n_trig > 0) {
            STAGE("k_emit", 0);
            if (h->n_segrec > h->n_reads + h->n_reads / 2)       /* long reads: few candidates per column */
                CRUMBLE_LAUNCH_INDEP(k_emit_lanes, std::min(gcap, nblk(gn_trig, 64)), 64, st, B, P, n_trig, nd_trig, spec_abort, trigcol_pos, trigcol_off, colrec, items);
            else
                CRUMBLE_LAUNCH(k_emit, std::min(gcap / 4, nblk(gn_trig, 4)), 256, st, B, P, n_trig, nd_trig, spec_abort, trigcol_pos, trigcol_off, colrec, items);
            STAGE("k_str", 0);
            CRUMBLE_LAUNCH_INDEP(k_str, std::min(gcap, nblk(gn_items, 64)), 64, st, B, P, n_items, nd_items, spec_abort, items, d_overflow);
        }
    }
    h->phase = 3;
    return CRUMBLE_OK;
}

extern "C" int crumble_hip_run_fold(crumble_hip *h, const crumble_carry *in, crumble_carry *out)
{
    if (!h || !h->have_batch || h->phase != 3) { g_last_error = "crumble_hip_run_fold: run the trigger phase first"; return CRUMBLE_EINVAL; }
    RUN_LOCALS
    const crumble_carry cin = in ? *in : CARRY0;
    h->carry_in.min_pos = cin.min_pos; h->carry_in.max_pos = cin.max_pos; h->carry_in.min_pos2 = cin.min_pos2; h->carry_in.max_pos2 = cin.max_pos2;
    int32_t *d_carry = h->d_misc.as<int>() + 24;
    const int32_t c4[4] = {cin.min_pos, cin.max_pos, cin.min_pos2, cin.max_pos2};   /* no trigger column before the cut: unchanged */
    if (L > 0) {
        if (out) HIPCHK(hipMemcpyAsync(d_carry, c4, sizeof c4, hipMemcpyHostToDevice, st));
        if (n_trig > 0) {
            STAGE("k_fold", 0);
            int32_t *colhi = h->d_colhi.as<int32_t>(), *ub = h->d_ub.as<int32_t>();
            int pe;
            h->fold_small = !h->windowed && !in && !out && n_trig <= FOLD_SMALL_MAX;
            if (h->fold_small)                          /* a short list (a small contig): all of it, and the window marks, in one launch */
                CRUMBLE_LAUNCH(k_fold_small, 1, 1024, st, B, P, n_trig, nd_trig, spec_abort, trigcol_pos, trigcol_off, trigcnt, items, colhi, ub,
                               colrec, h->d_pos_mn2.as<int32_t>(), h->d_trig_mx2.as<int32_t>(), d_carry);
            else {
            const unsigned gt = std::min(gcap, nblk(gn_trig, 64));
            CRUMBLE_LAUNCH_INDEP(k_fold_bounds, gt, 64, st, B, P, n_trig, nd_trig, spec_abort, trigcol_pos, trigcol_off, trigcnt, items, colhi, cin.max_pos);
            if ((pe = prefix_max(h, colhi, gn_trig, nd_trig))) return pe;
            CRUMBLE_LAUNCH_INDEP(k_fold_ub, gt, 64, st, B, P, n_trig, nd_trig, spec_abort, trigcol_pos, colhi, ub, cin.max_pos2);
            if ((pe = prefix_max(h, ub, gn_trig, nd_trig))) return pe;
            CRUMBLE_LAUNCH_INDEP(k_fold, gt, 64, st, B, P, n_trig, nd_trig, spec_abort, trigcol_pos, trigcol_off, trigcnt, items, ub,
                                 colrec, h->d_pos_mn2.as<int32_t>(), h->d_trig_mx2.as<int32_t>(),
                                 cin.min_pos, cin.max_pos, cin.min_pos2, cin.max_pos2, B.p0 + B.stat_hi, d_carry);
            }
        }
        /* the window's first columns are "in window" while the carried state is live: snp:1880 tests
         * min_pos != INT_MAX, after the reset of every column with pos > max_pos2 (snp:1508) */
        if (h->windowed && cin.min_pos != INT_MAX && (int64_t)cin.max_pos2 >= (int64_t)B.p0 + B.col_lo)
            CRUMBLE_LAUNCH_INDEP(k_window_head, 64, 256, st, B, n_trig, trigcol_pos, cin.max_pos2, colrec);
    }
    if (out) {
        int32_t r4[4] = {c4[0], c4[1], c4[2], c4[3]};
        if (L > 0) { HIPCHK(hipMemcpyAsync(r4, d_carry, sizeof r4, hipMemcpyDeviceToHost, st)); HIPCHK(hipStreamSynchronize(st)); }
        out->min_pos = r4[0]; out->max_pos = r4[1]; out->min_pos2 = r4[2]; out->max_pos2 = r4[3];
    }
    h->phase = 4;
    return CRUMBLE_OK;
}

extern "C" int crumble_hip_run_finish(crumble_hip *h)
{
    if (!h || !h->have_batch || h->phase != 4) { g_last_error = "crumble_hip_run_finish: run the fold phase first"; return CRUMBLE_EINVAL; }
    RUN_LOCALS
    if (L > 0 && n_trig > 0 && !h->fold_small) {
        CRUMBLE_LAUNCH_INDEP(k_window, std::min(gcap, nblk(gn_trig, 64)), 64, st, B, n_trig, nd_trig, spec_abort, trigcol_pos, h->d_trig_mx2.as<int32_t>(), colrec);
    }
    if (n > 0) {
        STAGE("k_replay", 3.0 * nb);
        const bool multi = h->n_segrec > h->n_reads + h->n_reads / 2;
        const bool packed = nb > 0 && !h->hparams.softclip && h->packed_ok && !h->batch_flat && (!multi || h->have_vrec);
        /* the statistics come out of the replay when it is the last writer (no P-block) and counts every byte (a window
         * of a contig counts its owned reads only: k_stats) */
        const bool fused_stats = packed && !h->hparams.pblock && !h->windowed;
        if (packed) {
            /* The packed replay.  First what a byte needs to know beyond its own column: the per-read flags
             * (k_readflags, from the columns k_actions marks) and the back-fill ranges of the trigger columns (k_restore's
             * marks, a bit per base).  Then ONE writer per byte: k_replay_fast over 16-base chunks (per read for short
             * reads, per reference-consuming op for long reads), k_replay_reads for the general-CIGAR reads of a
             * short-read batch, k_replay_slow for the bases D/N columns report. */
            int ea;
            const size_t nfw = (size_t)(L > 0 ? L : 0) / 32 + 2, nbw = (size_t)nb / 32 + 4;
            if ((ea = h->d_act.reserve((size_t)L + 2 * ACT_PAD)) || (ea = h->d_marks.reserve((nfw + nbw) * 4 + 64))) return ea;
            uint8_t *act = h->d_act.as<uint8_t>() + ACT_PAD;
            uint32_t *flagbits = h->d_marks.as<uint32_t>(), *forcebits = flagbits + nfw;
            uint8_t *rfl = h->d_rflags.as<uint8_t>();
            HIPCHK(hipMemsetAsync(flagbits, 0, (nfw + nbw) * 4, st));
            if (L > 0) CRUMBLE_LAUNCH_INDEP(k_actions, nblk((L + 3) / 4, 256), 256, st, L, colrec, act, flagbits);
            if (multi) CRUMBLE_LAUNCH(k_readflags_wave, nblk(n, 4), 256, st, B, colrec, flagbits, rfl, forcebits);
            else       CRUMBLE_LAUNCH_INDEP(k_readflags, nblk(n, 256), 256, st, B, colrec, flagbits, rfl, forcebits);
            if (L > 0 && n_trig > 0) {
                const int coop = multi ? 0 : 1;                 /* (long reads: eight columns per wave) */
                const unsigned g = std::min(gcap, (unsigned)(coop ? gn_trig : (gn_trig + 7) / 8));
                CRUMBLE_LAUNCH(k_restore, g, 64, st, B, P, n_trig, nd_trig, spec_abort, h->d_trigcol_pos.as<int32_t>(), h->d_pos_mn2.as<int32_t>(),
                               h->d_trig_mx2.as<int32_t>(), h->d_out.as<uint8_t>(), coop, forcebits);
            }
            /* with the statistics inside, a workgroup flushes its histogram at its end: few enough of them for that, many
             * enough (several rounds of what the chip holds at once) for an even finish */
            const unsigned gfull = nblk((nb + 15) / 16, 256);
            const unsigned gf = (!fused_stats || gfull < h->persistent_blocks * 8) ? gfull : h->persistent_blocks * 8;
#define RFAST_LAUNCH(V_, PV_, S_) CRUMBLE_LAUNCH((k_replay_fast<V_, PV_, S_>), gf, 256, st, B, P, colrec, act, \
                                     h->d_chunk_read.as<int32_t>(), h->d_out.as<uint8_t>(), forcebits, stats, d_overflow + 4)
#define RFAST_LAUNCH2(V_, PV_) do { if (fused_stats) RFAST_LAUNCH(V_, PV_, true); else RFAST_LAUNCH(V_, PV_, false); } while (0)
            const bool pv = h->hparams.n_pv > 0;
            if (h->have_vrec) { if (pv) RFAST_LAUNCH2(true, true); else RFAST_LAUNCH2(true, false); }
            else {
                if (pv) RFAST_LAUNCH2(false, true); else RFAST_LAUNCH2(false, false);
                if (h->n_general_max > 0) {
                    /* sixteen reads per workgroup at a time (RR_GL lanes each); no more workgroups than the chip holds at once */
                    const unsigned gmax = h->persistent_blocks / 4 * 6 > 0 ? h->persistent_blocks / 4 * 6 : 1u;
                    const int64_t rpb = 256 / RR_GL;
                    const unsigned gg = (unsigned)((h->n_general_max + rpb - 1) / rpb < (int64_t)gmax ? (h->n_general_max + rpb - 1) / rpb : gmax);
                    if (fused_stats) CRUMBLE_LAUNCH(k_replay_reads<true>, gg, 256, st, B, P, colrec, h->d_genlist.as<int32_t>(), h->d_misc.as<int>() + 12,
                                                    h->d_out.as<uint8_t>(), forcebits, stats);
                    else             CRUMBLE_LAUNCH(k_replay_reads<false>, gg, 256, st, B, P, colrec, h->d_genlist.as<int32_t>(), h->d_misc.as<int>() + 12,
                                                    h->d_out.as<uint8_t>(), forcebits, stats);
                }
            }
#undef RFAST_LAUNCH2
#undef RFAST_LAUNCH
            if (h->n_slow > 0) {
                if (fused_stats) CRUMBLE_LAUNCH(k_replay_slow<true>, nblk(h->n_slow, 64), 64, st, B, P, colrec, h->d_pos_mn2.as<int32_t>(),
                                                h->d_misc.as<int>() + 11, h->d_slow_g.as<int64_t>(), h->d_slow_r.as<int32_t>(),
                                                h->d_out.as<uint8_t>(), (uint8_t *)nullptr, (uint8_t *)nullptr, forcebits, stats);
                else             CRUMBLE_LAUNCH(k_replay_slow<false>, nblk(h->n_slow, 64), 64, st, B, P, colrec, h->d_pos_mn2.as<int32_t>(),
                                                h->d_misc.as<int>() + 11, h->d_slow_g.as<int64_t>(), h->d_slow_r.as<int32_t>(),
                                                h->d_out.as<uint8_t>(), (uint8_t *)nullptr, (uint8_t *)nullptr, forcebits, stats);
            }
            if (fused_stats) CRUMBLE_LAUNCH_INDEP(k_stats_totals, 1, 64, st, n, nb, stats);
        } else {
            /* the byte-wise replays (-S, parameter sets outside the packed formulas, "replay" = flat): the flags are
             * raised on the way, the back-fill and the kept reads are written afterwards */
            HIPCHK(hipMemsetAsync(h->d_keepf.p, 0, (size_t)n * 2, st));      /* keepf and, behind it, b15f */
            if (nb > 0 && h->hparams.softclip)
                CRUMBLE_LAUNCH_INDEP(k_replay_general, nblk(nb, 256), 256, st, B, P, colrec, h->d_pos_mn2.as<int32_t>(),
                                     h->d_blk_read.as<int32_t>(), h->d_out.as<uint8_t>(), h->d_keepf.as<uint8_t>(),
                                     h->d_b15f.as<uint8_t>());
            else if (nb > 0) {
                CRUMBLE_LAUNCH_INDEP(k_replay_flat, nblk((nb + RPL_NB - 1) / RPL_NB, 256), 256, st, B, P, colrec, h->d_pos_mn2.as<int32_t>(),
                                     h->d_blk_read.as<int32_t>(), h->d_out.as<uint8_t>(), h->d_keepf.as<uint8_t>(),
                                     h->d_b15f.as<uint8_t>());
                if (h->n_slow > 0)
                    CRUMBLE_LAUNCH(k_replay_slow<false>, nblk(h->n_slow, 64), 64, st, B, P, colrec, h->d_pos_mn2.as<int32_t>(),
                                   h->d_misc.as<int>() + 11, h->d_slow_g.as<int64_t>(), h->d_slow_r.as<int32_t>(),
                                   h->d_out.as<uint8_t>(), h->d_keepf.as<uint8_t>(), h->d_b15f.as<uint8_t>(),
                                   (const uint32_t *)nullptr, stats);
            }
            if (L > 0 && n_trig > 0) {
                const int coop = multi ? 0 : 1;
                const unsigned g = std::min(gcap, (unsigned)(coop ? gn_trig : (gn_trig + 7) / 8));
                CRUMBLE_LAUNCH(k_restore, g, 64, st, B, P, n_trig, nd_trig, spec_abort,
                                     h->d_trigcol_pos.as<int32_t>(), h->d_pos_mn2.as<int32_t>(), h->d_trig_mx2.as<int32_t>(), h->d_out.as<uint8_t>(),
                                     coop, (uint32_t *)nullptr);
            }
            CRUMBLE_LAUNCH_INDEP(k_readfix, nblk(n, 256), 256, st, B, colrec, h->d_keepf.as<uint8_t>(),
                                 h->d_b15f.as<uint8_t>(), h->d_rflags.as<uint8_t>(), h->d_kept.as<int32_t>(), d_overflow + 2);
            CRUMBLE_LAUNCH_INDEP(k_keepfix, (nblk(n, 4) < 4096u ? nblk(n, 4) : 4096u), 256, st, B, P, h->d_kept.as<int32_t>(),
                                 d_overflow + 2, h->d_out.as<uint8_t>());
        }
        if (h->hparams.pblock) {
            STAGE("k_pblock", 2.0 * nb);
            const int pe = launch_pblock(h, st, n, nb, B.seq_off, h->d_out.as<uint8_t>(), h->d_rflags.as<uint8_t>(),
                                         h->d_blk_read.as<int32_t>(), h->b_pblock);
            if (pe) return pe;
        }
        if (!fused_stats) {
            STAGE("k_stats", 2.0 * nb);
            if (h->windowed)
                CRUMBLE_LAUNCH(k_stats, 1024, 256, st, h->own_b_lo, h->own_b_hi, h->win.read_end - h->win.read_first, h->d_out.as<uint8_t>(), B.quals, stats, d_overflow + 4);
            else
                CRUMBLE_LAUNCH(k_stats, 1024, 256, st, (int64_t)0, nb, n, h->d_out.as<uint8_t>(), B.quals, stats, d_overflow + 4);
        }
    }
    if (h->opt_stages) HIPCHK(hipEventRecord(h->ev[h->n_stages], st));
    /* fold this batch's counters into the engine's accumulated stats vector, on the device */
    CRUMBLE_LAUNCH_INDEP(k_stats_fold, nblk(CRUMBLE_ST_N, 64), 64, st, h->d_stats.as<u64>(), (const u64 *)stats, CRUMBLE_ST_N, (const int *)d_overflow, spec_abort,
                         (const int *)(h->d_misc.as<int>() + 8), (const int *)(h->d_misc.as<int>() + 15));
    int ovf[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    HIPCHK(hipMemcpyAsync(h->pinned, d_overflow, 64, hipMemcpyDeviceToHost, st));      /* the flags and (fast path) the scan's totals */
    HIPCHK(hipStreamSynchronize(st));
    memcpy(ovf, h->pinned, 64);
    HIPCHK(hipGetLastError());
    if (spec) memcpy(&h->totals, ovf + 8, sizeof(ScanVal));
    if (h->late_pending) {                              /* crumble_hip_contig: the expansion was not waited for */
        h->late_pending = false;
        (void)hipEventElapsedTime(&h->up_h2d_ms, h->ev_up[0], h->ev_up[1]);
        (void)hipEventElapsedTime(&h->up_p0_ms, h->ev_up[1], h->ev_up[2]);
        const int le = expand_late_errors(h);
        if (le) { h->have_batch = false; return le; }
    }
    if (ovf[4]) { g_last_error = "input quality >= 128 (SURVEY.md Q13)"; return CRUMBLE_EINVAL; }
    for (int i = 0; i < h->n_stages; i++)
        HIPCHK(hipEventElapsedTime(&h->stage_ms[i], h->ev[i], h->ev[i + 1]));
    if (spec && ovf[5]) {                               /* the lists did not fit (or a column is too deep): the careful way, with exact sizes */
        h->cap_trig = (int)std::min<int64_t>(INT_MAX / 2, std::max<int64_t>(h->cap_trig, (int64_t)h->totals.trigcols + h->totals.trigcols / 4));
        h->cap_items = (int)std::min<int64_t>(INT_MAX / 2, std::max<int64_t>(h->cap_items, (int64_t)h->totals.items + h->totals.items / 4));
        h->cap_ins = (int)std::min<int64_t>(INT_MAX / 2, std::max<int64_t>(h->cap_ins, (int64_t)h->totals.ins + h->totals.ins / 4));
        h->spec_aborted = true;
        return CRUMBLE_OK;
    }
    if (ovf[0]) { g_last_error = "STR repeat list overflow"; return CRUMBLE_EOVERFLOW; }
    h->have_run = true; h->phase = 5;
    return CRUMBLE_OK;
}


extern "C" int crumble_hip_run(crumble_hip *h)
{
    int e;
    if (!h) return CRUMBLE_EINVAL;
    /* a whole contig: first the fast way (no size read-back in the middle: capacities guessed from the batch's shape,
     * lengths read on the device); should the scan find a list too small or a column deeper than MAX_DEPTH, once more
     * with the read-back.  A window of a contig always takes the second way (its phases hand state on through the host). */
    h->spec = !h->windowed && h->opt_spec;
    h->spec_aborted = false;
    e = CRUMBLE_OK;
    if ((e = crumble_hip_run_columns(h)) || (e = crumble_hip_run_depth(h, nullptr, nullptr)) || (e = crumble_hip_run_triggers(h)) ||
        (e = crumble_hip_run_fold(h, nullptr, nullptr)) || (e = crumble_hip_run_finish(h))) { h->spec = false; return e; }
    if (h->spec && h->spec_aborted) {
        h->spec = false;
        if ((e = crumble_hip_run_columns(h)) || (e = crumble_hip_run_depth(h, nullptr, nullptr)) || (e = crumble_hip_run_triggers(h)) ||
            (e = crumble_hip_run_fold(h, nullptr, nullptr)) || (e = crumble_hip_run_finish(h)))
            return e;
    }
    h->spec = false;
    return CRUMBLE_OK;
}

/* Device -> pageable host memory.  A plain hipMemcpy into pageable memory goes through the runtime's own single
 * staging path (measured 14-15 GB/s on the MI355X box: 134 ms for chr20's 1.9 GB).  Here DL_LANES host threads take a
 * slice of the range each: asynchronous copies into the thread's two page-locked buffers on its own stream (several
 * SDMA engines at once), and while a chunk is in flight the thread copies the previous one on to the caller's memory. */
#ifndef CRUMBLE_EMU
/* device -> page-locked host memory by the GPU's own stores (posted writes over the link): sixteen bytes per lane */
__global__ void __launch_bounds__(256) k_d2h_store(uint4 *__restrict__ dst, const uint4 *__restrict__ src, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = src[i];
}
#endif
static int staged_d2h(crumble_hip *h, uint8_t *dst, const uint8_t *src, size_t bytes)
{
    const size_t CH = h->dl_chunk;
    /* (never the null stream: a copy there would wait for, and hold up, every blocking stream of the device) */
#define D2H_PLAIN() do { HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream)); HIPCHK(hipStreamSynchronize(h->stream)); } while (0)
    if (bytes < 4 * CH) { D2H_PLAIN(); return CRUMBLE_OK; }
#ifndef CRUMBLE_EMU
    {   /* a destination that is page-locked already (crumble_hip_host_alloc, hipHostRegister) needs no staging */
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, dst) == hipSuccess && at.type == hipMemoryTypeHost) {
            void *dd = nullptr;
            if (h->opt_dl_kernel && !(((uintptr_t)dst | (uintptr_t)src) & 15) && hipHostGetDevicePointer(&dd, dst, 0) == hipSuccess && dd) {
                const size_t n16 = bytes >> 4, tail = bytes & 15;
                if (n16) CRUMBLE_LAUNCH(k_d2h_store, 64, 256, h->stream, (uint4 *)dd, (const uint4 *)src, n16);
                if (tail) HIPCHK(hipMemcpyAsync(dst + (n16 << 4), src + (n16 << 4), tail, hipMemcpyDeviceToHost, h->stream));
                HIPCHK(hipStreamSynchronize(h->stream));
                return CRUMBLE_OK;
            }
            (void)hipGetLastError();
            D2H_PLAIN();
            return CRUMBLE_OK;
        }
        (void)hipGetLastError();                        /* (an ordinary pointer: the query's error is not ours) */
    }
#endif
    if (!h->dl_ok || h->dl_bufsz < CH) {
        /* two page-locked buffers of ONE chunk per lane (16 buffers of the configured chunk size, not of the largest one);
         * should the page-locked memory not be there, what was taken is given back and the copy goes the plain way */
        bool ok = true;
        for (auto &l : h->dl) {
            if (!l.st && hipStreamCreateWithFlags(&l.st, hipStreamNonBlocking) != hipSuccess) { l.st = nullptr; ok = false; break; }
            for (int i = 0; i < 2 && ok; i++) {
                if (l.buf[i] && h->dl_bufsz < CH) { (void)hipHostFree(l.buf[i]); l.buf[i] = nullptr; }
                if (!l.buf[i] && hipHostMalloc(&l.buf[i], CH, 0) != hipSuccess) { l.buf[i] = nullptr; ok = false; }
                if (ok && !l.ev[i] && hipEventCreate(&l.ev[i]) != hipSuccess) { l.ev[i] = nullptr; ok = false; }
            }
            if (!ok) break;
        }
        if (!ok) {
            for (auto &l : h->dl) for (int i = 0; i < 2; i++) if (l.buf[i]) { (void)hipHostFree(l.buf[i]); l.buf[i] = nullptr; }
            h->dl_ok = false; h->dl_bufsz = 0;
            (void)hipGetLastError();
            D2H_PLAIN();
            return CRUMBLE_OK;
        }
        h->dl_ok = true; h->dl_bufsz = CH;
    }
    std::atomic<int> failed(0);
    auto lane = [&](int t) {
        if (hipSetDevice(h->device) != hipSuccess) { failed = 1; return; }
        crumble_hip::DlLane &l = h->dl[t];
        /* slices in whole chunks, so that every copy but the last is a whole chunk */
        const size_t nch = (bytes + CH - 1) / CH, c0 = nch * (size_t)t / DL_LANES, c1 = nch * (size_t)(t + 1) / DL_LANES;
        auto issue = [&](size_t c) {
            const size_t off = c * CH, len = off + CH <= bytes ? CH : bytes - off;
            if (hipMemcpyAsync(l.buf[c & 1], src + off, len, hipMemcpyDeviceToHost, l.st) != hipSuccess ||
                hipEventRecord(l.ev[c & 1], l.st) != hipSuccess) failed = 1;
        };
        if (c0 < c1) issue(c0);
        for (size_t c = c0; c < c1 && !failed; c++) {
            if (c + 1 < c1) issue(c + 1);
            if (hipEventSynchronize(l.ev[c & 1]) != hipSuccess) { failed = 1; break; }
            const size_t off = c * CH, len = off + CH <= bytes ? CH : bytes - off;
            memcpy(dst + off, l.buf[c & 1], len);
        }
        (void)hipStreamSynchronize(l.st);
    };
    std::thread th[DL_LANES];
    int started = 1;
    try { for (int t = 1; t < DL_LANES; t++) { th[t] = std::thread(lane, t); started = t + 1; } }
    catch (...) { failed = 2; }                         /* (no thread to be had: the slices nobody took are copied below) */
    lane(0);
    for (int t = 1; t < started; t++) th[t].join();
    if (failed == 2) {
        g_last_error = "staged device -> host copy: could not start a host thread";
        D2H_PLAIN();
        return CRUMBLE_OK;
    }
    if (failed) { g_last_error = "staged device -> host copy failed"; return CRUMBLE_EHIP; }
    return CRUMBLE_OK;
#undef D2H_PLAIN
}

extern "C" int crumble_hip_download(crumble_hip *h, uint8_t *qual_out, uint8_t *read_flags_out, int64_t *stats)
{
    if (!h || !h->have_run) { g_last_error = "nothing to download: call crumble_hip_run first"; return CRUMBLE_EINVAL; }
    HIPCHK(hipSetDevice(h->device));
    h->down_bytes = (qual_out ? (size_t)h->n_bases : 0) + (read_flags_out ? (size_t)h->n_reads : 0) + (stats ? sizeof(int64_t) * CRUMBLE_ST_N : 0);
    if ((size_t)h->n_bases < 4 * h->dl_chunk) {
        /* a small batch: the three copies queued behind one another, one wait (three blocking copies of a few
         * kilobytes each were a tenth of a small contig's fixed cost) */
        int64_t s[CRUMBLE_ST_N];
        if (qual_out && h->n_bases) HIPCHK(hipMemcpyAsync(qual_out, h->d_out.p, (size_t)h->n_bases, hipMemcpyDeviceToHost, h->stream));
        if (read_flags_out && h->n_reads) HIPCHK(hipMemcpyAsync(read_flags_out, h->d_rflags.p, (size_t)h->n_reads, hipMemcpyDeviceToHost, h->stream));
        if (stats) HIPCHK(hipMemcpyAsync(s, h->d_batch_stats.p, sizeof s, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        if (stats) for (int i = 0; i < CRUMBLE_ST_N; i++) stats[i] += s[i];
        return CRUMBLE_OK;
    }
    if (qual_out && h->n_bases) { const int e = staged_d2h(h, qual_out, h->d_out.as<uint8_t>(), (size_t)h->n_bases); if (e) return e; }
    if (read_flags_out && h->n_reads) { const int e = staged_d2h(h, read_flags_out, h->d_rflags.as<uint8_t>(), (size_t)h->n_reads); if (e) return e; }
    if (stats) {
        int64_t s[CRUMBLE_ST_N];
        HIPCHK(hipMemcpyAsync(s, h->d_batch_stats.p, sizeof s, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        for (int i = 0; i < CRUMBLE_ST_N; i++) stats[i] += s[i];
    }
    return CRUMBLE_OK;
}

/* ---- coded download (include/crumble_hip.h): 2 bits per base + the escape bytes ---------------------------------- */
extern "C" int crumble_hip_encode(crumble_hip *h, crumble_coded *c)
{
    if (!h || !c || !h->have_run) { g_last_error = "nothing to encode: call crumble_hip_run first"; return CRUMBLE_EINVAL; }
    HIPCHK(hipSetDevice(h->device));
    hipStream_t st = h->stream;
    const int64_t nb = h->n_bases;
    const int64_t nwords = (nb + 15) / 16, nblocks = (nb + ENC_BLOCK - 1) / ENC_BLOCK;
    static_assert(ENC_BLOCK == CRUMBLE_CODED_BLOCK, "the device's and the header's block size");
    int e;
    if ((e = h->d_codes.reserve((size_t)nwords * 4 + 64)) || (e = h->d_escoff.reserve((size_t)(nblocks + 1) * 8)) ||
        (e = h->d_esc.reserve((size_t)nb + 64)) || (e = h->d_misc.reserve(256)))
        return e;
    /* d_misc [32..33]: the escape total (64 bits), [34]: the two values (bytes) */
    unsigned long long *d_tot = (unsigned long long *)(h->d_misc.as<int>() + 32);
    uint8_t *d_pick = (uint8_t *)(h->d_misc.as<int>() + 34);
    CRUMBLE_LAUNCH(k_encode_pick, 1, 256, st, nb, (const uint8_t *)h->d_out.as<uint8_t>(), h->B.quals, h->d_params.as<DParams>(), d_pick, d_tot);
    if (nb > 0) {
        const unsigned g = (unsigned)std::min<int64_t>(nblocks, (int64_t)h->persistent_blocks * 8);
        CRUMBLE_LAUNCH(k_encode, g, 256, st, nb, (const uint8_t *)h->d_out.as<uint8_t>(), h->B.quals, h->d_params.as<DParams>(),
                       (const uint8_t *)d_pick, h->d_codes.as<uint32_t>(), h->d_escoff.as<int64_t>(), h->d_esc.as<uint8_t>(), d_tot);
    }
    HIPCHK(hipMemcpyAsync((char *)h->pinned + 320, d_tot, 16, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    unsigned long long tot = 0;
    memcpy(&tot, (char *)h->pinned + 320, 8);
    memset(c, 0, sizeof *c);
    c->n_bases = nb; c->n_words = nwords; c->n_blocks = nblocks; c->n_escapes = (int64_t)tot;
    memcpy(c->value, (char *)h->pinned + 328, 2);
    for (int q = 0; q < 256; q++)
        c->cap[q] = (uint8_t)((q > h->hparams.qcap && !h->hparams.preserve_qual[q]) ? h->hparams.qcap : q);   /* cap_quality snp:1325-1332 */
    h->have_coded = true; h->coded_escapes = (int64_t)tot;
    return CRUMBLE_OK;
}

extern "C" int crumble_hip_download_coded(crumble_hip *h, const crumble_coded *c, uint8_t *read_flags_out, int64_t *stats)
{
    if (!h || !c || !h->have_run || !h->have_coded) { g_last_error = "crumble_hip_download_coded: call crumble_hip_encode first"; return CRUMBLE_EINVAL; }
    /* the sizes are the engine's own: a struct whose counts were changed since crumble_hip_encode() is refused, not trusted */
    if (c->n_bases != h->n_bases || c->n_words != (h->n_bases + 15) / 16 || c->n_blocks != (h->n_bases + ENC_BLOCK - 1) / ENC_BLOCK ||
        c->n_escapes != h->coded_escapes) {
        g_last_error = "crumble_hip_download_coded: the crumble_coded is not the one crumble_hip_encode() filled for this run"; return CRUMBLE_EINVAL;
    }
    if ((c->n_words && !c->codes) || (c->n_blocks && !c->esc_off) || (c->n_escapes && !c->esc)) {
        g_last_error = "crumble_hip_download_coded: the coded result's buffers are not set"; return CRUMBLE_EINVAL;
    }
    HIPCHK(hipSetDevice(h->device));
    int e;
    h->down_bytes = (size_t)c->n_words * 4 + (size_t)c->n_blocks * 8 + (size_t)c->n_escapes + (read_flags_out ? (size_t)h->n_reads : 0) + (stats ? sizeof(int64_t) * CRUMBLE_ST_N : 0);
    if (c->n_words && (e = staged_d2h(h, (uint8_t *)c->codes, h->d_codes.as<uint8_t>(), (size_t)c->n_words * 4))) return e;
    hipStream_t st = h->stream;
    int64_t s[CRUMBLE_ST_N];
    if (c->n_blocks) HIPCHK(hipMemcpyAsync(c->esc_off, h->d_escoff.p, (size_t)c->n_blocks * 8, hipMemcpyDeviceToHost, st));
    if (c->n_escapes) {
        if ((size_t)c->n_escapes >= 4 * h->dl_chunk) { if ((e = staged_d2h(h, c->esc, h->d_esc.as<uint8_t>(), (size_t)c->n_escapes))) return e; }
        else HIPCHK(hipMemcpyAsync(c->esc, h->d_esc.p, (size_t)c->n_escapes, hipMemcpyDeviceToHost, st));
    }
    if (read_flags_out && h->n_reads) HIPCHK(hipMemcpyAsync(read_flags_out, h->d_rflags.p, (size_t)h->n_reads, hipMemcpyDeviceToHost, st));
    if (stats) HIPCHK(hipMemcpyAsync(s, h->d_batch_stats.p, sizeof s, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (stats) for (int i = 0; i < CRUMBLE_ST_N; i++) stats[i] += s[i];
    return CRUMBLE_OK;
}

extern "C" int crumble_hip_contig_run(crumble_hip *h, const crumble_batch *b)
{
    int e;
    if (!h || !b || b->n_reads < 0) return CRUMBLE_EINVAL;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipEventRecord(h->ev_up[0], h->stream));
    /* the expansion's tail runs into the run's first kernels: its late errors are looked at with the run's own read-back */
    if ((e = copy_raw(h, b)) || (e = expand(h, false))) return e;
    return crumble_hip_run(h);
}

extern "C" int crumble_hip_contig(crumble_hip *h, const crumble_batch *b, uint8_t *qual_out,
                                  uint8_t *read_flags_out, int64_t *stats)
{
    const int e = crumble_hip_contig_run(h, b);
    if (e) return e;
    return crumble_hip_download(h, qual_out, read_flags_out, stats);
}

extern "C" int crumble_hip_kernel_times(crumble_hip *h, const char **names, float *ms, double *bytes, int max)
{
    if (!h || !h->have_run) return 0;
    int n = h->n_stages < max ? h->n_stages : max;
    for (int i = 0; i < n; i++) {
        if (names) names[i] = h->stage_name[i];
        if (ms) ms[i] = h->stage_ms[i];
        if (bytes) bytes[i] = h->stage_bytes[i];
    }
    return n;
}

extern "C" int crumble_hip_upload_times(crumble_hip *h, float *h2d_ms, float *p0_ms)
{
    if (!h || !h->have_batch) return CRUMBLE_EINVAL;
    if (h2d_ms) *h2d_ms = h->up_h2d_ms;
    if (p0_ms) *p0_ms = h->up_p0_ms;
    return CRUMBLE_OK;
}

extern "C" int crumble_hip_transfer_bytes(crumble_hip *h, int64_t *up, int64_t *down)
{
    if (!h) return CRUMBLE_EINVAL;
    if (up) *up = (int64_t)h->up_bytes;
    if (down) *down = (int64_t)h->down_bytes;
    return CRUMBLE_OK;
}

extern "C" void crumble_hip_alloc_stats(int64_t *calls, double *ms)
{
    if (calls) *calls = g_alloc_calls.load();
    if (ms) *ms = g_alloc_us.load() / 1e3;
}

extern "C" void *crumble_hip_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, 0) != hipSuccess) { g_last_error = "hipHostMalloc failed"; return nullptr; }
    return p;
}
extern "C" void crumble_hip_host_free(void *p) { if (p) (void)hipHostFree(p); }

/* ------------------------------------------------------------------ stats */
extern "C" void *crumble_hip_stats_devptr(crumble_hip *h) { return h ? h->d_stats.p : nullptr; }
extern "C" int crumble_hip_stats_reset(crumble_hip *h)
{
    if (!h) return CRUMBLE_EINVAL;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(fill_sync(h->stream, h->d_stats.p, 0, CRUMBLE_ST_N * 8));
    return CRUMBLE_OK;
}
extern "C" int crumble_hip_stats_add(crumble_hip *h, const int64_t *stats)
{
    if (!h || !stats) return CRUMBLE_EINVAL;
    HIPCHK(hipSetDevice(h->device));
    int64_t cur[CRUMBLE_ST_N];
    HIPCHK(copy_sync(h->stream, cur, h->d_stats.p, sizeof cur, hipMemcpyDeviceToHost));
    for (int i = 0; i < CRUMBLE_ST_N; i++) cur[i] += stats[i];
    HIPCHK(copy_sync(h->stream, h->d_stats.p, cur, sizeof cur, hipMemcpyHostToDevice));
    return CRUMBLE_OK;
}
extern "C" int crumble_hip_stats_read(crumble_hip *h, int64_t *stats)
{
    if (!h || !stats) return CRUMBLE_EINVAL;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(copy_sync(h->stream, stats, h->d_stats.p, CRUMBLE_ST_N * 8, hipMemcpyDeviceToHost));
    return CRUMBLE_OK;
}

/* The single collective of the path: sum the int64 stats vector over ranks (RCCL over xGMI).  librccl is resolved at
 * run time so that the library loads on hosts without it; the call signatures, ncclInt64 and ncclSum come from the
 * build's own <rccl/rccl.h> (no literal ABI constants), and a library whose major version differs from the header's
 * is refused. */
#ifndef CRUMBLE_EMU
#include <rccl/rccl.h>
struct RcclApi {
    decltype(&ncclAllReduce) all_reduce = nullptr;
    decltype(&ncclCommInitAll) comm_init_all = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    bool ok = false;
};
static bool rccl_load(RcclApi &api);
static const RcclApi *rccl_api()
{
    /* (a function-local static: initialised once, by whichever thread comes first; the others wait) */
    static RcclApi api;
    static const bool ok = rccl_load(api);
    return ok ? &api : nullptr;
}
static bool rccl_load(RcclApi &api)
{
    void *lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) { g_last_error = "librccl.so not found"; return false; }
    auto get_version = (decltype(&ncclGetVersion))dlsym(lib, "ncclGetVersion");
    int v = 0;
    if (!get_version || get_version(&v) != ncclSuccess || v / 10000 != NCCL_VERSION_CODE / 10000) {
        g_last_error = "librccl.so: version " + std::to_string(v) + " does not match the build's rccl.h (" + std::to_string(NCCL_VERSION_CODE) + ")";
        return false;
    }
    api.all_reduce = (decltype(&ncclAllReduce))dlsym(lib, "ncclAllReduce");
    api.comm_init_all = (decltype(&ncclCommInitAll))dlsym(lib, "ncclCommInitAll");
    api.comm_destroy = (decltype(&ncclCommDestroy))dlsym(lib, "ncclCommDestroy");
    if (!api.all_reduce || !api.comm_init_all || !api.comm_destroy) { g_last_error = "librccl.so: entry points missing"; return false; }
    api.ok = true;
    return true;
}
#endif

extern "C" int crumble_hip_stats_allreduce(crumble_hip *h, void *nccl_comm)
{
#ifdef CRUMBLE_EMU
    (void)h; (void)nccl_comm;
    return CRUMBLE_ERCCL;
#else
    if (!h || !nccl_comm) return CRUMBLE_EINVAL;
    const RcclApi *api = rccl_api();
    if (!api) return CRUMBLE_ERCCL;
    HIPCHK(hipSetDevice(h->device));
    static_assert(sizeof(int64_t) == 8, "the statistics vector is int64");
    const ncclResult_t rc = api->all_reduce(h->d_stats.p, h->d_stats.p, CRUMBLE_ST_N, ncclInt64, ncclSum, (ncclComm_t)nccl_comm, h->stream);
    if (rc != ncclSuccess) { g_last_error = "ncclAllReduce failed (code " + std::to_string((int)rc) + ")"; return CRUMBLE_ERCCL; }
    HIPCHK(hipStreamSynchronize(h->stream));
    return CRUMBLE_OK;
#endif
}

/* One RCCL communicator per device of ONE process (crumble-hip -j N: an engine and a host thread per device).
 * comms[i] belongs to devices[i]; pass it to crumble_hip_stats_allreduce() of the engine on that device, all
 * engines concurrently (one call per host thread). */
extern "C" int crumble_hip_comm_init_all(int n, const int *devices, void **comms)
{
#ifdef CRUMBLE_EMU
    (void)n; (void)devices; (void)comms;
    return CRUMBLE_ERCCL;
#else
    if (n < 1 || !devices || !comms) return CRUMBLE_EINVAL;
    const RcclApi *api = rccl_api();
    if (!api) return CRUMBLE_ERCCL;
    const ncclResult_t rc = api->comm_init_all((ncclComm_t *)comms, n, devices);
    if (rc != ncclSuccess) { g_last_error = "ncclCommInitAll failed (code " + std::to_string((int)rc) + ")"; return CRUMBLE_ERCCL; }
    return CRUMBLE_OK;
#endif
}
extern "C" void crumble_hip_comm_destroy(void *comm)
{
#ifndef CRUMBLE_EMU
    if (!comm) return;
    const RcclApi *api = rccl_api();
    if (api) (void)api->comm_destroy((ncclComm_t)comm);
#else
    (void)comm;
#endif
}

/* ---------------------------------------------------- operator-level seams */
extern "C" int64_t crumble_hip_columns(crumble_hip *h, crumble_column *cols, int64_t max)
{
    if (!h || !h->have_run) return CRUMBLE_EINVAL;
    if (hipSetDevice(h->device) != hipSuccess) return CRUMBLE_EHIP;
    const int L = h->B.L;
    std::vector<uint32_t> rec(L > 0 ? L : 1);
    std::vector<int32_t> np(L > 0 ? L : 1), sc(L > 0 ? L : 1), mn2(L > 0 ? L : 1);
    if (L > 0) {
        if (copy_sync(h->stream, rec.data(), h->d_colrec.p, (size_t)L * 4, hipMemcpyDeviceToHost) != hipSuccess ||
            copy_sync(h->stream, np.data(), h->d_nplp.p, (size_t)L * 4, hipMemcpyDeviceToHost) != hipSuccess ||
            copy_sync(h->stream, sc.data(), h->d_score.p, (size_t)L * 4, hipMemcpyDeviceToHost) != hipSuccess ||
            copy_sync(h->stream, mn2.data(), h->d_pos_mn2.p, (size_t)L * 4, hipMemcpyDeviceToHost) != hipSuccess)
            return CRUMBLE_EHIP;
    }
    int64_t k = 0;
    for (int i = 0; i < L; i++) {
        const uint32_t r = rec[i];
        if (!(r & CR_EXISTS)) continue;
        if (cols && k < max) {
            crumble_column *c = &cols[k];
            memset(c, 0, sizeof(*c));
            c->pos = h->B.p0 + i; c->n_plp = np[i];
            c->skipped = (r & CR_ALLREFSKIP) ? 1 : (r & CR_TOODEEP) ? 2 : 0;
            if (!c->skipped) {
                c->skipped = (uint8_t)(((r & CR_R_DEEP) ? CRUMBLE_WHY_DEEP : 0) | ((r & CR_R_CLIP) ? CRUMBLE_WHY_CLIP : 0) |
                                       ((r & CR_R_INSLEN) ? CRUMBLE_WHY_INDEL_LEN : 0) | ((r & CR_R_INDELOV) ? CRUMBLE_WHY_INDEL_COVERAGE : 0));
                c->score = sc[i];
                c->call1 = CR_A1(r); c->call2 = CR_A2(r);
                c->preserve = CR_PRESERVE(r); c->perfect = (r & CR_PERFECT) ? 1 : 0;
                c->keep_qual = (r & CR_KEEP) ? 1 : 0;
                c->trigger = (r & CR_TRIG) ? 1 : 0; c->window = (r & CR_WIN) ? 1 : 0;
                c->min_pos2 = (r & CR_TRIG) ? mn2[i] : 0;
            }
        }
        k++;
    }
    return k;
}

/* The columns the reference reports with -b (snp_score.c:1496,1676,1768,1802,1810), in position order. */
extern "C" int64_t crumble_hip_suspect(crumble_hip *h, int32_t *pos, uint8_t *why, int64_t max)
{
    if (!h || !h->have_run) return CRUMBLE_EINVAL;
    if (hipSetDevice(h->device) != hipSuccess) return CRUMBLE_EHIP;
    const int L = h->B.L;
    if (L <= 0) return 0;
    std::vector<uint32_t> rec((size_t)L);
    if (copy_sync(h->stream, rec.data(), h->d_colrec.p, (size_t)L * 4, hipMemcpyDeviceToHost) != hipSuccess) return CRUMBLE_EHIP;
    int64_t k = 0;
    for (int i = 0; i < L; i++) {
        const uint32_t r = rec[i];
        if (!(r & CR_EXISTS) || (r & CR_ALLREFSKIP)) continue;
        const unsigned w = (r & CR_TOODEEP) ? CRUMBLE_WHY_VDEEP
                         : (((r & CR_R_DEEP) ? CRUMBLE_WHY_DEEP : 0) | ((r & CR_R_CLIP) ? CRUMBLE_WHY_CLIP : 0) |
                            ((r & CR_R_INSLEN) ? CRUMBLE_WHY_INDEL_LEN : 0) | ((r & CR_R_INDELOV) ? CRUMBLE_WHY_INDEL_COVERAGE : 0));
        if (!w) continue;
        if (k < max) { if (pos) pos[k] = h->B.p0 + i; if (why) why[k] = (uint8_t)w; }
        k++;
    }
    return k;
}

extern "C" int crumble_hip_mask_lc(crumble_hip *h, int64_t nq, const int64_t *read_idx, const int32_t *qpos1,
                                   const uint8_t *is_indel, int32_t *lo, int32_t *hi)
{
    if (!h || !h->have_batch || nq < 0) return CRUMBLE_EINVAL;
    if (nq == 0) return CRUMBLE_OK;
    HIPCHK(hipSetDevice(h->device));
    std::vector<TrigItem> it((size_t)nq);
    for (int64_t i = 0; i < nq; i++) {
        if (read_idx[i] < 0 || read_idx[i] >= h->n_reads || qpos1[i] < 0) return CRUMBLE_EINVAL;
        it[i].read = (int32_t)read_idx[i];
        it[i].qpos1_kind = qpos1[i] | ((is_indel[i] ? 1 : 0) << 30);
        it[i].lo = INT_MAX; it[i].hi = 0;
    }
    DevBuf d;
    int e = d.reserve((size_t)nq * sizeof(TrigItem));
    if (e) return e;
    int *d_overflow = h->d_misc.as<int>();
    HIPCHK(fill_sync(h->stream, d_overflow, 0, 4));
    HIPCHK(copy_sync(h->stream, d.p, it.data(), (size_t)nq * sizeof(TrigItem), hipMemcpyHostToDevice));
    CRUMBLE_LAUNCH_INDEP(k_str, nblk(nq, 64), 64, h->stream, h->B, h->d_params.as<DParams>(), (int)nq, (const int *)nullptr, (const int *)nullptr, d.as<TrigItem>(), d_overflow);
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(copy_sync(h->stream, it.data(), d.p, (size_t)nq * sizeof(TrigItem), hipMemcpyDeviceToHost));
    int ovf = 0;
    HIPCHK(copy_sync(h->stream, &ovf, d_overflow, 4, hipMemcpyDeviceToHost));
    d.release();
    if (ovf) return CRUMBLE_EOVERFLOW;
    for (int64_t i = 0; i < nq; i++) { lo[i] = it[i].lo; hi[i] = it[i].hi; }
    return CRUMBLE_OK;
}

extern "C" int crumble_hip_pblock(crumble_hip *h, int64_t n_reads, const int64_t *seq_off, uint8_t *quals)
{
    if (!h || n_reads < 0 || !seq_off) return CRUMBLE_EINVAL;
    if (n_reads == 0) return CRUMBLE_OK;
    HIPCHK(hipSetDevice(h->device));
    const int64_t nb = seq_off[n_reads];
    for (int64_t i = 0; i < nb; i++)
        if (quals[i] >= 128) { g_last_error = "quality >= 128 (SURVEY.md Q13)"; return CRUMBLE_EINVAL; }
    DevBuf dq, ds;
    int e;
    if ((e = dq.reserve((size_t)nb + 64)) || (e = ds.reserve((size_t)(n_reads + 1) * 8))) return e;
    if (nb) HIPCHK(copy_sync(h->stream, dq.p, quals, (size_t)nb, hipMemcpyHostToDevice));
    HIPCHK(copy_sync(h->stream, ds.p, seq_off, (size_t)(n_reads + 1) * 8, hipMemcpyHostToDevice));
    if (h->hparams.pblock && nb > 0 && (e = launch_pblock(h, h->stream, n_reads, nb, ds.as<int64_t>(), dq.as<uint8_t>(), nullptr, nullptr, h->opt_pblock)))
        return e;
    HIPCHK(hipStreamSynchronize(h->stream));
    if (nb) HIPCHK(copy_sync(h->stream, quals, dq.p, (size_t)nb, hipMemcpyDeviceToHost));
    dq.release(); ds.release();
    return CRUMBLE_OK;
}

extern "C" int crumble_hip_indel_only(crumble_hip *h, int dist, int q, uint8_t *qual_out)
{
    if (!h || !h->have_batch || dist < 0 || q < 0 || q > 127 || !qual_out) return CRUMBLE_EINVAL;
    HIPCHK(hipSetDevice(h->device));
    const int64_t nb = h->n_bases;
    if (nb == 0) return CRUMBLE_OK;
    {   /* no run here, so no k_stats: the qualities' input check on its own */
        int *d_err = h->d_misc.as<int>() + 8;
        HIPCHK(hipMemsetAsync(d_err, 0, 4, h->stream));
        CRUMBLE_LAUNCH_INDEP(k_p0_qualcheck, nblk((nb + 15) / 16, 256), 256, h->stream, nb, h->B.quals, d_err);
        int bad = 0;
        HIPCHK(hipMemcpyAsync(h->pinned, d_err, 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        memcpy(&bad, h->pinned, 4);
        if (bad) { g_last_error = "input quality >= 128 (SURVEY.md Q13)"; return CRUMBLE_EINVAL; }
    }
    CRUMBLE_LAUNCH_INDEP(k_indel_only, nblk(nb, 256), 256, h->stream, h->B, h->d_blk_read.as<int32_t>(), dist, q,
                         h->d_out.as<uint8_t>());
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(copy_sync(h->stream, qual_out, h->d_out.p, (size_t)nb, hipMemcpyDeviceToHost));
    h->have_run = false;                 /* d_out no longer holds crumble_hip_run's result */
    return CRUMBLE_OK;
}

/* ------------------------------------------------------------------ BGZF codec on the device (include/crumble_hip.h) */
#include "gz_host.h"
#include "stream_host.h"

#ifdef CRUMBLE_EMU
/* test hook of the emulated build (tests/emu): how many tiles took k_columns2's two finalize paths since the last call */
extern "C" void crumble_emu_fin_counts(unsigned long long *out) { out[0] = kc2_emu_fin[0]; out[1] = kc2_emu_fin[1]; kc2_emu_fin[0] = kc2_emu_fin[1] = 0; }
#endif
