/* hip_emu.cpp -- TEST INFRASTRUCTURE ONLY (see hip_emu.h). */
#include "hip_emu.h"
#include <pthread.h>

thread_local dim3 threadIdx, blockIdx;
dim3 blockDim, gridDim;

static pthread_barrier_t g_block_barrier;
static pthread_barrier_t g_wave_barrier[16];
static uint64_t g_wave_slots[16][64];
static bool g_coop = false;

void __syncthreads()
{
    if (!g_coop) { fprintf(stderr, "hip_emu: __syncthreads() in a lane-independent launch\n"); abort(); }
    pthread_barrier_wait(&g_block_barrier);
}
void emu_wave_sync()
{
    if (!g_coop) { fprintf(stderr, "hip_emu: wave op in a lane-independent launch\n"); abort(); }
    pthread_barrier_wait(&g_wave_barrier[threadIdx.x >> 6]);
}
uint64_t *emu_wave_slots() { return g_wave_slots[threadIdx.x >> 6]; }

void emu_run_blocks(dim3 grid, dim3 block, bool independent, const std::function<void()> &body)
{
    const uint64_t nblocks = (uint64_t)grid.x * grid.y * grid.z;
    const unsigned nthreads = block.x * block.y * block.z;
    if (block.y != 1 || block.z != 1 || nthreads > 1024) { fprintf(stderr, "hip_emu: 1-D blocks only\n"); abort(); }
    blockDim = block; gridDim = grid;
    if (!nblocks || !nthreads) return;

    if (independent) {
        unsigned T = std::thread::hardware_concurrency(); if (T < 1) T = 1; if (T > 8) T = 8;
        if (nblocks < T) T = (unsigned)nblocks;
        g_coop = false;
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < T; t++)
            pool.emplace_back([&, t]() {
                for (uint64_t b = t; b < nblocks; b += T) {
                    blockIdx = dim3((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((uint64_t)grid.x * grid.y)));
                    for (unsigned i = 0; i < nthreads; i++) { threadIdx = dim3(i); body(); }
                }
            });
        for (auto &th : pool) th.join();
        return;
    }

    g_coop = true;
    pthread_barrier_init(&g_block_barrier, nullptr, nthreads);
    unsigned nwaves = (nthreads + 63) / 64;
    for (unsigned w = 0; w < nwaves; w++) {
        unsigned n = nthreads - w * 64; if (n > 64) n = 64;
        pthread_barrier_init(&g_wave_barrier[w], nullptr, n);
    }
    std::vector<std::thread> pool;
    for (unsigned i = 0; i < nthreads; i++)
        pool.emplace_back([&, i]() {
            threadIdx = dim3(i);
            for (uint64_t b = 0; b < nblocks; b++) {
                blockIdx = dim3((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((uint64_t)grid.x * grid.y)));
                body();
                pthread_barrier_wait(&g_block_barrier);   /* block boundary */
            }
        });
    for (auto &th : pool) th.join();
    pthread_barrier_destroy(&g_block_barrier);
    for (unsigned w = 0; w < nwaves; w++) pthread_barrier_destroy(&g_wave_barrier[w]);
    g_coop = false;
}
