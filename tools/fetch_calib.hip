// Calibration of rocprofv3's FETCH_SIZE for the load shapes of this engine (MI355X_MICROARCH.md, "HBM": the
// counter is TCC_EA0_RDREQ x 64 B and reads 1/2 of a wide coalesced stream; other shapes are uncalibrated).
// Every kernel reads a buffer of KNOWN size (far beyond L2 and the 256 MiB Infinity Cache) exactly once in unique
// bytes; run under `rocprofv3 --pmc FETCH_SIZE` (tools/fetch_calib.sh) and divide the counter by the bytes.
//
//   k_stream16   16 B per lane, coalesced                          (k_stats, k_replay_fast: the guide's x2 case)
//   k_rec32      one 32-byte record per lane, two 16-byte loads    (k_columns2's stage phase: SegRec)
//   k_seg20      4-byte-aligned 16 B + 4 B per lane, four lanes per 150-byte read segment of 64 bytes, rows 146
//                bytes apart, tile after tile (k_columns2's produce phase: bases / qualities of a 30x pileup)
//
// Build: hipcc --offload-arch=gfx950 -O2 tools/fetch_calib.hip -o build/fetch_calib     Usage: build/fetch_calib [GiB]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct __attribute__((aligned(4))) U4 { uint32_t x, y, z, w; };

__global__ void k_stream16(const uint4 *__restrict__ p, size_t n, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

__global__ void k_rec32(const uint4 *__restrict__ p, size_t nrec, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrec; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 a = p[2 * i], b = p[2 * i + 1];
        acc ^= a.x ^ a.w ^ b.y ^ b.z;
    }
    if (acc == 0x12345678u) *sink = acc;
}

// reads of 150 bytes back to back (read r at 150 r); read r starts at column 4 r (depth 37.5).  A wave takes a
// 64-column tile t at a time, walks the reads that overlap it 16 per step (lane = (row, 16-column segment)) and loads
// the 16 bytes under its segment the way k_columns2 does: 16 B from the 4-byte aligned address below + the next 4 B.
__global__ void k_seg20(const uint8_t *__restrict__ p, size_t nbytes, int64_t ntiles, uint32_t *sink, unsigned long long *requested)
{
    const int lane = threadIdx.x & 63, rsub = lane >> 2, seg = lane & 3;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    uint32_t acc = 0;
    unsigned long long req = 0;
    // consecutive tiles in units of 5 per wave (the kernel's work units), units dealt round-robin
    for (int64_t u = wave; u * 5 < ntiles; u += nwaves) {
        for (int64_t t = u * 5; t < u * 5 + 5 && t < ntiles; t++) {
            const int64_t c0 = t * 64;
            // reads overlapping [c0, c0 + 64): start column 4 r in (c0 - 150, c0 + 64)
            int64_t r_lo = (c0 - 149 + 3) / 4; if (r_lo < 0) r_lo = 0;
            const int64_t r_hi = (c0 + 63) / 4;                      // inclusive
            for (int64_t r0 = r_lo; r0 <= r_hi; r0 += 16) {
                const int64_t r = r0 + rsub;
                if (r > r_hi) continue;
                const int64_t col = c0 + 16 * seg, q = col - 4 * r;      // query offset of the segment's first column
                int64_t a = q < 0 ? 0 : q, b = q + 16 > 150 ? 150 : q + 16;
                if (a >= b) continue;
                const int64_t byte = 150 * r + q;                    // may start before the read (masked bytes), as in the kernel
                if (byte < 0 || (size_t)(byte + 24) > nbytes) continue;
                const int64_t va = byte & ~(int64_t)3;
                const U4 v = *(const U4 *)(p + va);
                const uint32_t w = *(const uint32_t *)(p + va + 16);
                acc ^= v.x ^ v.y ^ v.z ^ v.w ^ w;
                req += 20;
            }
        }
    }
    if (acc == 0x12345678u) *sink = acc;
    atomicAdd(requested, req);
}

int main(int argc, char **argv)
{
    const double gib = argc > 1 ? atof(argv[1]) : 2.0;
    const size_t nbytes = (size_t)(gib * (1ull << 30)) / 4800 * 4800;     // multiple of 150 and of 32
    uint8_t *buf; uint32_t *sink; unsigned long long *req, hreq = 0;
    CHECK(hipMalloc(&buf, nbytes + 256));
    CHECK(hipMalloc(&sink, 4)); CHECK(hipMalloc(&req, 8));
    CHECK(hipMemset(buf, 0x5a, nbytes + 256)); CHECK(hipMemset(req, 0, 8));
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms;
    for (int rep = 0; rep < 2; rep++) {
        CHECK(hipEventRecord(e0));
        k_stream16<<<4096, 256>>>((const uint4 *)buf, nbytes / 16, sink);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("k_stream16 bytes %zu ms %.3f GB/s %.0f\n", nbytes, ms, nbytes / ms / 1e6);
        CHECK(hipEventRecord(e0));
        k_rec32<<<4096, 256>>>((const uint4 *)buf, nbytes / 32, sink);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("k_rec32 bytes %zu ms %.3f GB/s %.0f\n", nbytes, ms, nbytes / ms / 1e6);
        CHECK(hipMemset(req, 0, 8));
        const int64_t nreads = (int64_t)(nbytes / 150), ntiles = (4 * nreads + 150) / 64;
        CHECK(hipEventRecord(e0));
        k_seg20<<<256 * 3, 256>>>(buf, nbytes, ntiles, sink, req);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemcpy(&hreq, req, 8, hipMemcpyDeviceToHost));
        if (rep) printf("k_seg20 bytes %zu requested %llu ms %.3f GB/s(unique) %.0f\n", nbytes, hreq, ms, nbytes / ms / 1e6);
    }
    return 0;
}
