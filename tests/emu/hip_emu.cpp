/* hip_emu.cpp -- TEST INFRASTRUCTURE ONLY (see hip_emu.h).
 * Blocks are spread over host threads; inside a block every GPU thread is a
 * ucontext fiber on that host thread, so __syncthreads() / wave shuffles are
 * cooperative yields (microseconds, not OS-scheduler milliseconds). */
#include "hip_emu.h"
#include <ucontext.h>
#include <atomic>

thread_local dim3 threadIdx, blockIdx;
dim3 blockDim, gridDim;

namespace {
struct Fiber { ucontext_t ctx; char *stack = nullptr; bool done = true; };
struct Barrier { unsigned count = 0; unsigned gen = 0; };
struct Worker {
    std::vector<Fiber> fib;
    ucontext_t sched;
    int cur = -1;
    unsigned nthreads = 0;
    Barrier block_bar, wave_bar[16];
    uint64_t wave_slots[16][64];
    const std::function<void()> *body = nullptr;
    bool coop = false;
};
thread_local Worker *g_w = nullptr;
const size_t STACK = 256 * 1024;

void fiber_entry()
{
    Worker *w = g_w;
    (*w->body)();
    w->fib[w->cur].done = true;
    swapcontext(&w->fib[w->cur].ctx, &w->sched);
}
inline void yield_fiber()
{
    Worker *w = g_w;
    int me = w->cur;
    swapcontext(&w->fib[me].ctx, &w->sched);
}
unsigned alive_in(Worker *w, unsigned lo, unsigned hi)
{
    unsigned n = 0;
    for (unsigned i = lo; i < hi && i < w->nthreads; i++) if (!w->fib[i].done) n++;
    return n;
}
void barrier_wait(Barrier &b, unsigned lo, unsigned hi)
{
    Worker *w = g_w;
    unsigned my = b.gen;
    b.count++;
    for (;;) {
        if (b.gen != my) return;
        if (b.count >= alive_in(w, lo, hi)) { b.count = 0; b.gen++; return; }
        yield_fiber();
    }
}
} // namespace

void __syncthreads()
{
    if (!g_w || !g_w->coop) { fprintf(stderr, "hip_emu: __syncthreads() in a lane-independent launch\n"); abort(); }
    barrier_wait(g_w->block_bar, 0, g_w->nthreads);
}
void emu_wave_sync()
{
    if (!g_w || !g_w->coop) { fprintf(stderr, "hip_emu: wave op in a lane-independent launch\n"); abort(); }
    unsigned wv = threadIdx.x >> 6;
    barrier_wait(g_w->wave_bar[wv], wv * 64, wv * 64 + 64);
}
uint64_t *emu_wave_slots() { return g_w->wave_slots[threadIdx.x >> 6]; }

void emu_run_blocks(dim3 grid, dim3 block, bool independent, const std::function<void()> &body)
{
    /* blockDim / gridDim are process-wide here: launches from several host threads (crumble-hip -j N runs one
     * engine per thread) take turns */
    static std::mutex launch_mu;
    std::lock_guard<std::mutex> launch_lock(launch_mu);
    const uint64_t nblocks = (uint64_t)grid.x * grid.y * grid.z;
    const unsigned nthreads = block.x * block.y * block.z;
    if (block.y != 1 || block.z != 1 || nthreads > 1024) { fprintf(stderr, "hip_emu: 1-D blocks only\n"); abort(); }
    blockDim = block; gridDim = grid;
    if (!nblocks || !nthreads) return;
    unsigned T = std::thread::hardware_concurrency(); if (T < 1) T = 1; if (T > 8) T = 8;
    if (nblocks < T) T = (unsigned)nblocks;
    std::atomic<uint64_t> next(0);
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < T; t++)
        pool.emplace_back([&]() {
            Worker w;
            g_w = &w;
            w.coop = !independent; w.body = &body; w.nthreads = nthreads;
            if (!independent) {
                w.fib.resize(nthreads);
                for (auto &f : w.fib) f.stack = (char *)malloc(STACK);
            }
            for (;;) {
                uint64_t b = next.fetch_add(1);
                if (b >= nblocks) break;
                blockIdx = dim3((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((uint64_t)grid.x * grid.y)));
                if (independent) {
                    for (unsigned i = 0; i < nthreads; i++) { threadIdx = dim3(i); body(); }
                    continue;
                }
                w.block_bar = Barrier();
                for (auto &wb : w.wave_bar) wb = Barrier();
                for (unsigned i = 0; i < nthreads; i++) {
                    Fiber &f = w.fib[i];
                    getcontext(&f.ctx);
                    f.ctx.uc_stack.ss_sp = f.stack; f.ctx.uc_stack.ss_size = STACK; f.ctx.uc_link = nullptr;
                    f.done = false;
                    makecontext(&f.ctx, fiber_entry, 0);
                }
                unsigned remaining = nthreads;
                while (remaining) {
                    remaining = 0;
                    for (unsigned i = 0; i < nthreads; i++) {
                        if (w.fib[i].done) continue;
                        w.cur = (int)i; threadIdx = dim3(i);
                        swapcontext(&w.sched, &w.fib[i].ctx);
                        if (!w.fib[i].done) remaining++;
                    }
                }
            }
            for (auto &f : w.fib) free(f.stack);
            g_w = nullptr;
        });
    for (auto &th : pool) th.join();
}
