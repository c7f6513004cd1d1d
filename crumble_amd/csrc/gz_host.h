/*
 * gz_host.h -- host side of the device BGZF codec (gz_kernels.h): crumble_gz_* of include/crumble_hip.h.
 * Included by engine.hip (one translation unit: the CPU test emulator builds exactly that file).
 *
 * A crumble_gz is independent of any engine: its own non-blocking stream and device buffers (grow-only), so a driver's
 * reader and writer threads can each own one beside the engines that run the contigs.  Calls on one handle are serial.
 */
#ifndef CRUMBLE_GZ_HOST_H
#define CRUMBLE_GZ_HOST_H

#include "gz_kernels.h"

#define GZ_BATCH_RAW ((int64_t)256 << 20)      /* uncompressed bytes per round of launches (the token buffer is 4 x that) */
#define GZ_IN_PAD    4096                       /* bytes behind the compressed buffer: k_gz_tokens stages whole 256-byte chunks ahead */

struct crumble_gz {
    int device = 0;
    hipStream_t stream = nullptr;
    DevBuf d_comp, d_blk, d_tokoff, d_tok, d_ntok, d_status, d_raw, d_slots, d_sizes, d_off, d_out, d_tabs, d_scratch, d_cand;
    int opt_lanes = -1;                         /* crumble_gz_set_option("tokens", ...): -1 by member count, 0 a wave per member, 1 a lane per member */
    void *pinned = nullptr;                     /* 4 KiB page-locked: sizes and totals read back */
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    float ms_h2d = 0, ms_kernels = 0, ms_d2h = 0;
    int opt_lpw = 0;                            /* walking lanes per wave of k_gz_tokens_lanes: 0 = by the number of members, 8 / 16 / 64 */
    int ncu = 256;                              /* compute units (hipGetDeviceProperties once: the call costs milliseconds) */
    int64_t last_in = 0, last_out = 0;
};

static void gz_make_tables(uint32_t *tab /* 256 + 32 */)
{
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c & 1u) ? (c >> 1) ^ GZ_POLY : c >> 1;
        tab[i] = c;
    }
    uint32_t p = 1u << 30;                       /* x^1 */
    tab[256] = p;
    for (int n = 1; n < 32; n++) tab[256 + n] = p = gz_multmodp(p, p);
}

extern "C" crumble_gz *crumble_gz_create(int device, int *err)
{
    int dummy; if (!err) err = &dummy;
    *err = CRUMBLE_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        g_last_error = "no usable HIP device for the BGZF codec"; *err = CRUMBLE_ENODEVICE; return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) { *err = CRUMBLE_EHIP; return nullptr; }
    crumble_gz *g = new (std::nothrow) crumble_gz();
    if (!g) { *err = CRUMBLE_ENOMEM; return nullptr; }
    g->device = device;
    bool ok = hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking) == hipSuccess;
    for (auto &e : g->ev) ok = ok && hipEventCreate(&e) == hipSuccess;
    ok = ok && hipHostMalloc(&g->pinned, 4096, 0) == hipSuccess;
#ifndef CRUMBLE_EMU
    { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, device) == hipSuccess && pr.multiProcessorCount > 0) g->ncu = pr.multiProcessorCount; }
#endif
    uint32_t tabs[288];
    gz_make_tables(tabs);
    ok = ok && g->d_tabs.reserve(sizeof tabs) == CRUMBLE_OK && copy_sync(g->stream, g->d_tabs.p, tabs, sizeof tabs, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) { g_last_error = "BGZF codec: device set-up failed"; *err = CRUMBLE_EHIP; crumble_gz_destroy(g); return nullptr; }
#ifndef CRUMBLE_EMU
    {   /* every kernel once over nothing: a kernel's first launch costs milliseconds (measured: 6.5 ms in front of the first
         * k_gz_tokens_lanes of a run), which belongs to the set-up and not to the first file's first members */
        hipStream_t st = g->stream;
        const uint32_t *tb = g->d_tabs.as<uint32_t>();
        hipLaunchKernelGGL(k_gz_tokens_lanes<8>, dim3(1), dim3(64), 0, st, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
        hipLaunchKernelGGL(k_gz_tokens_lanes<16>, dim3(1), dim3(64), 0, st, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
        hipLaunchKernelGGL(k_gz_tokens, dim3(1), dim3(GZ_WAVES * 64), 0, st, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
        hipLaunchKernelGGL(k_gz_resolve, dim3(1), dim3(GZ_RT), 0, st, 0, nullptr, nullptr, nullptr, nullptr, nullptr, tb, tb + 256, nullptr, 0);
        hipLaunchKernelGGL(k_gz_deflate, dim3(1), dim3(GZ_DT), 0, st, 0, nullptr, (int64_t)0, 1, nullptr, nullptr, nullptr, nullptr, tb, tb + 256);
        if (g->d_off.reserve(64) == CRUMBLE_OK) hipLaunchKernelGGL(k_gz_offsets, dim3(1), dim3(1024), 0, st, 0, nullptr, g->d_off.as<int64_t>());
        hipLaunchKernelGGL(k_gz_gather, dim3(1), dim3(256), 0, st, 0, nullptr, nullptr, nullptr, nullptr);
        {   /* ... and one large copy to page-locked host memory: the first one of a process costs 8 ms (measured in front of the
             * first drained members) */
            void *hp = nullptr;
            const size_t wn = (size_t)4 << 20;
            if (g->d_out.reserve(wn) == CRUMBLE_OK && hipHostMalloc(&hp, wn, 0) == hipSuccess) {
                (void)hipMemcpyAsync(hp, g->d_out.p, wn, hipMemcpyDeviceToHost, st);
                (void)hipStreamSynchronize(st);
                (void)hipHostFree(hp);
            }
        }
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
            g_last_error = "BGZF codec: kernel warm-up failed"; *err = CRUMBLE_EHIP; crumble_gz_destroy(g); return nullptr;
        }
    }
#endif
    return g;
}

extern "C" void crumble_gz_destroy(crumble_gz *g)
{
    if (!g) return;
    dev_free_deferred(0);
    (void)hipSetDevice(g->device);
    if (g->stream) (void)hipStreamSynchronize(g->stream);
    DevBuf *all[] = { &g->d_comp, &g->d_blk, &g->d_tokoff, &g->d_tok, &g->d_ntok, &g->d_status, &g->d_raw, &g->d_slots, &g->d_sizes,
                      &g->d_off, &g->d_out, &g->d_tabs, &g->d_scratch, &g->d_cand };
    for (DevBuf *b : all) b->release();
    for (auto &e : g->ev) if (e) (void)hipEventDestroy(e);
    if (g->pinned) (void)hipHostFree(g->pinned);
    if (g->stream) (void)hipStreamDestroy(g->stream);
    delete g;
}

extern "C" int64_t crumble_gz_deflate_bound(int64_t raw_len)
{
    const int64_t nb = raw_len > 0 ? (raw_len + GZ_MAXBLK - 1) / GZ_MAXBLK : 0;
    return raw_len + nb * (18 + 5 + 8);
}

static const char *gz_status_text(unsigned s)
{
    if (s & GZ_E_CRC) return "CRC-32 mismatch";
    if (s & GZ_E_SIZE) return "inflated size differs from ISIZE";
    if (s & GZ_E_INPUT) return "DEFLATE stream runs past the member";
    if (s & GZ_E_DIST) return "distance beyond the start of the member or output beyond ISIZE";
    if (s & GZ_E_SYMBOL) return "unassigned Huffman code";
    if (s & GZ_E_CODES) return "invalid code lengths";
    return "invalid block header";
}

/* Inflates n_blocks BGZF members: blk[i].coff / clen = where member i's DEFLATE payload sits in comp[0, comp_len) (behind
 * its gzip header, in front of its CRC-32 / ISIZE trailer), ulen / crc = that trailer, uoff = where its bytes go in
 * raw[0, raw_len).  uoff must not decrease and the members must not overlap in raw.  check_crc = 0 skips the CRC-32. */
/* raw: host destination (copied back round by round), or NULL with raw_dev: a DEVICE destination -- member i's bytes go to
 * raw_dev[blk[i].uoff ...] and stay there (crumble_stream, section 9c) */
static int gz_inflate_impl(crumble_gz *g, const uint8_t *comp, int64_t comp_len, const crumble_gz_block *blk, int64_t n_blocks,
                           uint8_t *raw, uint8_t *raw_dev, int64_t raw_len, int check_crc)
{
    if (!g || n_blocks < 0 || comp_len < 0 || raw_len < 0 || (n_blocks && (!comp || !blk || (!raw && !raw_dev)))) return CRUMBLE_EINVAL;
    static_assert(sizeof(crumble_gz_block) == sizeof(GzBlock), "crumble_gz_block layout");
    HIPCHK(hipSetDevice(g->device));
    hipStream_t st = g->stream;
    g->ms_h2d = g->ms_kernels = g->ms_d2h = 0; g->last_in = comp_len; g->last_out = 0;
    if (!n_blocks) return CRUMBLE_OK;
    int64_t prev_end = 0;
    for (int64_t i = 0; i < n_blocks; i++) {
        const crumble_gz_block &b = blk[i];
        if (b.coff < 0 || b.coff + (int64_t)b.clen > comp_len || b.ulen > 65536u || b.uoff < prev_end || b.uoff + (int64_t)b.ulen > raw_len) {
            g_last_error = "crumble_gz_inflate: block table out of range"; return CRUMBLE_EINVAL;
        }
        prev_end = b.uoff + (int64_t)b.ulen;
    }
    int e;
    const int ncu = g->ncu;
    if ((e = g->d_comp.reserve((size_t)comp_len + GZ_IN_PAD))) return e;
    HIPCHK(hipEventRecord(g->ev[0], st));
    HIPCHK(hipMemcpyAsync(g->d_comp.p, comp, (size_t)comp_len, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync((uint8_t *)g->d_comp.p + comp_len, 0, GZ_IN_PAD, st));
    HIPCHK(hipEventRecord(g->ev[1], st));
    float acc_k = 0, acc_d = 0, ms = 0;
    HIPCHK(hipEventSynchronize(g->ev[1]));
    HIPCHK(hipEventElapsedTime(&ms, g->ev[0], g->ev[1])); g->ms_h2d = ms;
    std::vector<int64_t> tokoff;
    std::vector<uint32_t> stat;
    for (int64_t i0 = 0; i0 < n_blocks; ) {
        /* a round: members [i0, i1) with at most GZ_BATCH_RAW bytes of output */
        /* (a lane per member wants every member of the call in flight at once: its time hardly depends on their number) */
        const int64_t round_raw = g->opt_lanes != 0 ? 8 * GZ_BATCH_RAW : GZ_BATCH_RAW;      /* (2 GiB of records = 8 GiB of token slots at most) */
        int64_t i1 = i0, sum = 0;
        while (i1 < n_blocks && (i1 == i0 || sum + (int64_t)blk[i1].ulen <= round_raw)) { sum += blk[i1].ulen; i1++; }
        const int64_t n = i1 - i0;
        const int64_t u0 = blk[i0].uoff, u1 = blk[i1 - 1].uoff + (int64_t)blk[i1 - 1].ulen;
        tokoff.resize((size_t)n + 1);
        int64_t t = 0;
        for (int64_t i = 0; i < n; i++) { tokoff[(size_t)i] = t; t += blk[i0 + i].ulen; }
        tokoff[(size_t)n] = t;
        if ((e = g->d_blk.reserve((size_t)n * sizeof(GzBlock))) || (e = g->d_tokoff.reserve((size_t)(n + 1) * 8)) ||
            (e = g->d_tok.reserve((size_t)(t + 64) * 4)) || (e = g->d_ntok.reserve((size_t)n * 4)) || (e = g->d_status.reserve((size_t)n * 4)) ||
            (!raw_dev && (e = g->d_raw.reserve((size_t)(u1 - u0) + 64))))
            return e;
        HIPCHK(hipMemsetAsync(g->d_status.p, 0xff, (size_t)n * 4, st));      /* (k_gz_tokens writes every member's word) */
        HIPCHK(hipMemcpyAsync(g->d_blk.p, blk + i0, (size_t)n * sizeof(GzBlock), hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(g->d_tokoff.p, tokoff.data(), (size_t)(n + 1) * 8, hipMemcpyHostToDevice, st));
        HIPCHK(hipEventRecord(g->ev[0], st));
        /* the walk: a lane per member (measured on MI355X, tools/inflate_sweep.py: 11 ms against the wave walk's 23 for 500 members,
         * 19 against 108 for 10 000); the wave per member stays as an option (crumble_gz_set_option) */
        const bool lanes = g->opt_lanes != 0;
        if (lanes) {
            /* walking lanes per wave: eight (eight waves per CU) was the fastest at every size measured, 500 to 32 000 members */
            const int lpw = g->opt_lpw ? g->opt_lpw : 8;
#define GZ_LANES_LAUNCH(LPW_) CRUMBLE_LAUNCH(k_gz_tokens_lanes<LPW_>, (unsigned)((n + LPW_ - 1) / LPW_), 64, st, (int)n, g->d_blk.as<GzBlock>(), g->d_comp.as<uint8_t>(), \
                           g->d_tok.as<uint32_t>(), g->d_tokoff.as<int64_t>(), g->d_ntok.as<uint32_t>(), g->d_status.as<uint32_t>())
            if (lpw == 8) GZ_LANES_LAUNCH(8); else if (lpw == 16) GZ_LANES_LAUNCH(16); else GZ_LANES_LAUNCH(64);
#undef GZ_LANES_LAUNCH
        } else
        CRUMBLE_LAUNCH(k_gz_tokens, (unsigned)((n + GZ_WAVES - 1) / GZ_WAVES), GZ_WAVES * 64, st, (int)n, g->d_blk.as<GzBlock>(),
                       g->d_comp.as<uint8_t>(), g->d_tok.as<uint32_t>(), g->d_tokoff.as<int64_t>(), g->d_ntok.as<uint32_t>(), g->d_status.as<uint32_t>());
        /* (uoff is relative to the caller's raw: the round's image starts at u0) */
        CRUMBLE_LAUNCH(k_gz_resolve, (unsigned)(n < 4 * ncu ? n : 4 * ncu), GZ_RT, st, (int)n, g->d_blk.as<GzBlock>(), g->d_tok.as<uint32_t>(), g->d_tokoff.as<int64_t>(),
                       g->d_ntok.as<uint32_t>(), g->d_status.as<uint32_t>(), g->d_tabs.as<uint32_t>(), g->d_tabs.as<uint32_t>() + 256,
                       raw_dev ? raw_dev : g->d_raw.as<uint8_t>() - u0, check_crc);
        HIPCHK(hipGetLastError());                           /* (a launch that the device refuses must not pass as an empty result) */
        HIPCHK(hipEventRecord(g->ev[1], st));
        stat.resize((size_t)n);
        HIPCHK(hipMemcpyAsync(stat.data(), g->d_status.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
        if (!raw_dev) HIPCHK(hipMemcpyAsync(raw + u0, g->d_raw.p, (size_t)(u1 - u0), hipMemcpyDeviceToHost, st));
        HIPCHK(hipEventRecord(g->ev[2], st));
        HIPCHK(hipStreamSynchronize(st));
        HIPCHK(hipEventElapsedTime(&ms, g->ev[0], g->ev[1])); acc_k += ms;
        HIPCHK(hipEventElapsedTime(&ms, g->ev[1], g->ev[2])); acc_d += ms;
        for (int64_t i = 0; i < n; i++)
            if (stat[(size_t)i]) {
                char buf[160];
                snprintf(buf, sizeof buf, "BGZF member %lld: %s (status %u)", (long long)(i0 + i), gz_status_text(stat[(size_t)i]), stat[(size_t)i]);
                g_last_error = buf;
                return CRUMBLE_EINVAL;
            }
        g->last_out += u1 - u0;
        i0 = i1;
    }
    g->ms_kernels = acc_k; g->ms_d2h = acc_d;
    return CRUMBLE_OK;
}

extern "C" int crumble_gz_inflate(crumble_gz *g, const uint8_t *comp, int64_t comp_len, const crumble_gz_block *blk, int64_t n_blocks,
                                  uint8_t *raw, int64_t raw_len, int check_crc)
{
    if (n_blocks && !raw) return CRUMBLE_EINVAL;
    return gz_inflate_impl(g, comp, comp_len, blk, n_blocks, raw, nullptr, raw_len, check_crc);
}

/* Deflates raw[0, raw_len) into BGZF members of 0xff00 input bytes each (the last one shorter), side by side in out; no
 * end-of-file member.  level 0: stored blocks; any other value: the one compression level the device has.
 * (raw: host source, or NULL with raw_dev: the bytes are on the device already) */
static int gz_deflate_impl(crumble_gz *g, const uint8_t *raw, const uint8_t *raw_dev, int64_t raw_len, int level, uint8_t *out, int64_t out_cap, int64_t *out_len)
{
    if (!g || raw_len < 0 || !out_len || (raw_len && ((!raw && !raw_dev) || !out))) return CRUMBLE_EINVAL;
    *out_len = 0;
    HIPCHK(hipSetDevice(g->device));
    hipStream_t st = g->stream;
    g->ms_h2d = g->ms_kernels = g->ms_d2h = 0; g->last_in = raw_len; g->last_out = 0;
    if (!raw_len) return CRUMBLE_OK;
    level = level == 0 ? 0 : 1;                          /* 0: stored; anything else, zlib's -1 "default" included: the device's one level */
    if (out_cap < crumble_gz_deflate_bound(raw_len)) { g_last_error = "crumble_gz_deflate: output buffer below crumble_gz_deflate_bound()"; return CRUMBLE_EINVAL; }
    int e;
    const int ncu = g->ncu;
    const int64_t per_round = (GZ_BATCH_RAW / GZ_MAXBLK) * GZ_MAXBLK;
    float ms = 0;
    int64_t produced = 0;
    for (int64_t r0 = 0; r0 < raw_len; r0 += per_round) {
        const int64_t len = raw_len - r0 < per_round ? raw_len - r0 : per_round;
        const int nb = (int)((len + GZ_MAXBLK - 1) / GZ_MAXBLK);
        if ((!raw_dev && (e = g->d_raw.reserve((size_t)len + 64))) || (e = g->d_tok.reserve((size_t)nb * 65536 * 4)) || (e = g->d_cand.reserve((size_t)nb * 65536 * 4)) || (e = g->d_slots.reserve((size_t)nb * GZ_SLOT)) ||
            (e = g->d_sizes.reserve((size_t)nb * 4)) || (e = g->d_off.reserve((size_t)(nb + 1) * 8)) || (e = g->d_out.reserve((size_t)nb * GZ_SLOT)))
            return e;
        HIPCHK(hipEventRecord(g->ev[0], st));
        if (!raw_dev) HIPCHK(hipMemcpyAsync(g->d_raw.p, raw + r0, (size_t)len, hipMemcpyHostToDevice, st));
        HIPCHK(hipEventRecord(g->ev[1], st));
        CRUMBLE_LAUNCH(k_gz_deflate, (unsigned)(nb < ncu ? nb : ncu), GZ_DT, st, nb, raw_dev ? raw_dev + r0 : g->d_raw.as<const uint8_t>(), len, level, g->d_tok.as<uint32_t>(), g->d_cand.as<uint32_t>(),
                       g->d_slots.as<uint8_t>(), g->d_sizes.as<uint32_t>(), g->d_tabs.as<uint32_t>(), g->d_tabs.as<uint32_t>() + 256);
        CRUMBLE_LAUNCH(k_gz_offsets, 1, 1024, st, nb, g->d_sizes.as<uint32_t>(), g->d_off.as<int64_t>());
        CRUMBLE_LAUNCH(k_gz_gather, (unsigned)nb, 256, st, nb, g->d_slots.as<uint8_t>(), g->d_sizes.as<uint32_t>(), g->d_off.as<int64_t>(), g->d_out.as<uint8_t>());
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(g->ev[2], st));
        HIPCHK(hipMemcpyAsync(g->pinned, g->d_off.as<int64_t>() + nb, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        int64_t total = 0;
        memcpy(&total, g->pinned, 8);
        if (total <= 0 || produced + total > out_cap) { g_last_error = "crumble_gz_deflate: internal size error"; return CRUMBLE_EOVERFLOW; }
        HIPCHK(hipMemcpyAsync(out + produced, g->d_out.p, (size_t)total, hipMemcpyDeviceToHost, st));
        HIPCHK(hipEventRecord(g->ev[3], st));
        HIPCHK(hipStreamSynchronize(st));
        HIPCHK(hipEventElapsedTime(&ms, g->ev[0], g->ev[1])); g->ms_h2d += ms;
        HIPCHK(hipEventElapsedTime(&ms, g->ev[1], g->ev[2])); g->ms_kernels += ms;
        HIPCHK(hipEventElapsedTime(&ms, g->ev[2], g->ev[3])); g->ms_d2h += ms;
        produced += total;
    }
    *out_len = produced; g->last_out = produced;
    return CRUMBLE_OK;
}

/* The same in two halves for a device source of at most one round (crumble_stream's drains): begin enqueues the kernels and
 * returns; end waits, checks and copies the members out.  Between the two the caller may do other work on other streams; it
 * must not use this codec object, and the source bytes must stay as they are. */
static int gz_deflate_begin(crumble_gz *g, const uint8_t *raw_dev, int64_t raw_len, int level)
{
    if (!g || !raw_dev || raw_len <= 0 || raw_len > (GZ_BATCH_RAW / GZ_MAXBLK) * GZ_MAXBLK) return CRUMBLE_EINVAL;
    HIPCHK(hipSetDevice(g->device));
    hipStream_t st = g->stream;
    g->ms_h2d = g->ms_kernels = g->ms_d2h = 0; g->last_in = raw_len; g->last_out = 0;
    level = level == 0 ? 0 : 1;
    const int ncu = g->ncu;
    const int nb = (int)((raw_len + GZ_MAXBLK - 1) / GZ_MAXBLK);
    int e;
    if ((e = g->d_tok.reserve((size_t)nb * 65536 * 4)) || (e = g->d_cand.reserve((size_t)nb * 65536 * 4)) || (e = g->d_slots.reserve((size_t)nb * GZ_SLOT)) ||
        (e = g->d_sizes.reserve((size_t)nb * 4)) || (e = g->d_off.reserve((size_t)(nb + 1) * 8)) || (e = g->d_out.reserve((size_t)nb * GZ_SLOT)))
        return e;
    HIPCHK(hipEventRecord(g->ev[1], st));
    CRUMBLE_LAUNCH(k_gz_deflate, (unsigned)(nb < ncu ? nb : ncu), GZ_DT, st, nb, raw_dev, raw_len, level, g->d_tok.as<uint32_t>(), g->d_cand.as<uint32_t>(),
                   g->d_slots.as<uint8_t>(), g->d_sizes.as<uint32_t>(), g->d_tabs.as<uint32_t>(), g->d_tabs.as<uint32_t>() + 256);
    CRUMBLE_LAUNCH(k_gz_offsets, 1, 1024, st, nb, g->d_sizes.as<uint32_t>(), g->d_off.as<int64_t>());
    CRUMBLE_LAUNCH(k_gz_gather, (unsigned)nb, 256, st, nb, g->d_slots.as<uint8_t>(), g->d_sizes.as<uint32_t>(), g->d_off.as<int64_t>(), g->d_out.as<uint8_t>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(g->ev[2], st));
    HIPCHK(hipMemcpyAsync(g->pinned, g->d_off.as<int64_t>() + nb, 8, hipMemcpyDeviceToHost, st));
    return CRUMBLE_OK;
}
static int gz_deflate_end(crumble_gz *g, uint8_t *out, int64_t out_cap, int64_t *out_len)
{
    *out_len = 0;
    HIPCHK(hipSetDevice(g->device));
    hipStream_t st = g->stream;
    HIPCHK(hipStreamSynchronize(st));
    int64_t total = 0;
    memcpy(&total, g->pinned, 8);
    if (total <= 0 || total > out_cap) { g_last_error = "crumble_gz_deflate: internal size error"; return CRUMBLE_EOVERFLOW; }
    HIPCHK(hipMemcpyAsync(out, g->d_out.p, (size_t)total, hipMemcpyDeviceToHost, st));
    HIPCHK(hipEventRecord(g->ev[3], st));
    HIPCHK(hipStreamSynchronize(st));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, g->ev[1], g->ev[2])); g->ms_kernels = ms;
    HIPCHK(hipEventElapsedTime(&ms, g->ev[2], g->ev[3])); g->ms_d2h = ms;
    *out_len = total; g->last_out = total;
    return CRUMBLE_OK;
}

extern "C" int crumble_gz_deflate(crumble_gz *g, const uint8_t *raw, int64_t raw_len, int level, uint8_t *out, int64_t out_cap, int64_t *out_len)
{
    if (raw_len && !raw) return CRUMBLE_EINVAL;
    return gz_deflate_impl(g, raw, nullptr, raw_len, level, out, out_cap, out_len);
}

/* implementation choices that never change a result (the tests force both): "tokens" = auto | wave | lane */
extern "C" int crumble_gz_set_option(crumble_gz *g, const char *name, const char *value)
{
    if (!g || !name || !value) return CRUMBLE_EINVAL;
    if (!strcmp(name, "tokens")) {
        if (!strcmp(value, "auto")) g->opt_lanes = -1; else if (!strcmp(value, "wave")) g->opt_lanes = 0; else if (!strcmp(value, "lane")) g->opt_lanes = 1;
        else return CRUMBLE_EINVAL;
        return CRUMBLE_OK;
    }
    if (!strcmp(name, "lanes_per_wave")) {
        const int v = !strcmp(value, "auto") ? 0 : atoi(value);
        if (v != 0 && v != 8 && v != 16 && v != 64) return CRUMBLE_EINVAL;
        g->opt_lpw = v;
        return CRUMBLE_OK;
    }
    return CRUMBLE_EINVAL;
}

/* stream time of the last crumble_gz_inflate / _deflate: host -> device copies, kernels, device -> host copies */
extern "C" int crumble_gz_times(crumble_gz *g, float *h2d_ms, float *kernels_ms, float *d2h_ms)
{
    if (!g) return CRUMBLE_EINVAL;
    if (h2d_ms) *h2d_ms = g->ms_h2d;
    if (kernels_ms) *kernels_ms = g->ms_kernels;
    if (d2h_ms) *d2h_ms = g->ms_d2h;
    return CRUMBLE_OK;
}

#if defined(GZ_STAMP) && !defined(CRUMBLE_EMU)
extern "C" int crumble_gz_debug_stamps(unsigned long long *out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(gz_stamp), sizeof(unsigned long long) * 16) != hipSuccess) return CRUMBLE_EHIP;
    if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(gz_stamp), z, sizeof z); }
    return CRUMBLE_OK;
}
#endif

#endif /* CRUMBLE_GZ_HOST_H */
