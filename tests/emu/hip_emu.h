/*
 * hip_emu.h -- TEST INFRASTRUCTURE ONLY: a minimal CPU emulator of the subset
 * of HIP that crumble_amd/csrc uses, so that the *literal kernel sources* can
 * be unit-tested in a container without a GPU (pytest -m "not gpu").
 *
 * It is force-included (g++ -include) when building tests/emu/libcrumble_emu.so
 * from the very same engine.hip / kernels.h that hipcc compiles for gfx950.
 * The product library never contains or links this file, has no CPU fallback,
 * and the -m gpu parity tests always run the real HIP build.
 *
 * Model: blocks are spread over host threads; the threads of a block are fibers
 * of one host thread (so __syncthreads(), LDS atomics and wave shuffles behave), unless a
 * launch is declared lane-independent (CRUMBLE_LAUNCH_INDEP), in which case
 * lanes run sequentially and blocks are spread over host threads.
 */
#ifndef HIP_EMU_H
#define HIP_EMU_H
#define CRUMBLE_EMU 1

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <chrono>
#include <functional>
#include <thread>
#include <vector>
#include <mutex>
#include <condition_variable>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#define __shared__ static thread_local
#ifndef __restrict__
#define __restrict__
#endif

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint4 { unsigned x, y, z, w; };
struct uint2 { unsigned x, y; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct longlong2 { long long x, y; };
extern thread_local dim3 threadIdx, blockIdx;
extern dim3 blockDim, gridDim;
static const int warpSize = 64;

/* ---- block / wave cooperation ---- */
void __syncthreads();
void emu_wave_sync();                       /* rendezvous of the 64 lanes of a wave */
uint64_t *emu_wave_slots();                 /* 64 x 8-byte exchange slots of this wave */

template <class T> static inline T __shfl(T v, int src, int width = 64)
{
    uint64_t *s = emu_wave_slots();
    int lane = threadIdx.x & 63;
    uint64_t raw = 0; memcpy(&raw, &v, sizeof(T));
    s[lane] = raw; emu_wave_sync();
    int base = lane & ~(width - 1);
    uint64_t r = s[base + (src & (width - 1))];
    emu_wave_sync();
    T out; memcpy(&out, &r, sizeof(T)); return out;
}
template <class T> static inline T __shfl_down(T v, unsigned d, int width = 64)
{ int lane = threadIdx.x & 63; int src = (lane & (width - 1)) + (int)d; return __shfl(v, src < width ? (lane & ~(width - 1)) + src : lane, 64); }
template <class T> static inline T __shfl_up(T v, unsigned d, int width = 64)
{ int lane = threadIdx.x & 63; int src = (lane & (width - 1)) - (int)d; return __shfl(v, src >= 0 ? (lane & ~(width - 1)) + src : lane, 64); }
template <class T> static inline T __shfl_xor(T v, int m, int width = 64)
{ int lane = threadIdx.x & 63; return __shfl(v, lane ^ m, 64); }
static inline unsigned long long __ballot(int pred)
{
    uint64_t *s = emu_wave_slots();
    int lane = threadIdx.x & 63;
    s[lane] = pred ? 1 : 0; emu_wave_sync();
    unsigned long long m = 0;
    int n = (int)(blockDim.x - (threadIdx.x & ~63u)); if (n > 64) n = 64;
    for (int i = 0; i < n; i++) if (s[i]) m |= 1ULL << i;
    emu_wave_sync();
    return m;
}
static inline int __any(int p) { return __ballot(p) != 0; }
static inline int __all(int p) { int n = (int)(blockDim.x - (threadIdx.x & ~63u)); if (n > 64) n = 64;
    unsigned long long full = n == 64 ? ~0ULL : ((1ULL << n) - 1); return __ballot(p) == full; }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }

/* ---- atomics (global and "LDS") ---- */
template <class T> static inline T atomicAdd(T *p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicOr(T *p, T v)  { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicAnd(T *p, T v) { return __atomic_fetch_and(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicExch(T *p, T v){ return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicCAS(T *p, T c, T v) { __atomic_compare_exchange_n(p, &c, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED); return c; }
template <class T> static inline T atomicMin(T *p, T v) { T o = __atomic_load_n(p, __ATOMIC_RELAXED); while (v < o && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)); return o; }
template <class T> static inline T atomicMax(T *p, T v) { T o = __atomic_load_n(p, __ATOMIC_RELAXED); while (v > o && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)); return o; }
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }

/* ---- runtime API subset ---- */
typedef int hipError_t;
typedef void *hipStream_t;
struct emu_event { std::chrono::steady_clock::time_point t; };
typedef emu_event *hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNoDevice = 100 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
/* device memory arrives dirty, as on the GPU: a kernel that reads what nothing wrote must not pass here by luck */
static inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); if (*p) memset(*p, 0xA5, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <class T> static inline hipError_t hipMalloc(T **p, size_t n) { return hipMalloc((void **)p, n); }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
static inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { if (n) memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind k, hipStream_t) { return hipMemcpy(d, s, n, k); }
static inline hipError_t hipMemset(void *d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { return hipMemset(d, v, n); }
static inline hipError_t hipStreamCreate(hipStream_t *s) { *s = nullptr; return hipSuccess; }
#define hipStreamNonBlocking 1u
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char *hipGetErrorString(hipError_t) { return "emulated"; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new emu_event(); return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b)
{ *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return hipSuccess; }

/* ---- launches ---- */
void emu_run_blocks(dim3 grid, dim3 block, bool independent, const std::function<void()> &body);
template <class K, class... A>
static inline void emu_launch(bool indep, K kern, dim3 grid, dim3 block, A... args)
{ emu_run_blocks(grid, block, indep, [&]() { kern(args...); }); }

#define CRUMBLE_LAUNCH(kern, grid, block, stream, ...)       emu_launch(false, kern, dim3(grid), dim3(block), __VA_ARGS__)
#define CRUMBLE_LAUNCH_INDEP(kern, grid, block, stream, ...) emu_launch(true,  kern, dim3(grid), dim3(block), __VA_ARGS__)

#endif
