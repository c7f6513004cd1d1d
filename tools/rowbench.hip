// rowbench.hip -- the consumer's row step of k_columns2 (kcol2.h, KC2_ROW) in isolation: how many cycles does a wave
// take per row step when every wave of the CU does nothing else?  The kernel proper measures ~507 wave-cycles per row step
// on chr20 (KC2_EXP_STAMP) against ~112 of issue; this bench separates the block itself from its surroundings.
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/rowbench.hip -o build/rowbench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

#define NROWTAB (128 + 101)
#define ROWMASK 0x1fe0u
#define CLS(v_) (((v_) >> 1) & 7u)
#define MASKED asm volatile("")

template <int PADW> struct __attribute__((aligned(16))) WaveT {
    uint16_t plane[64][64];
    uint32_t pad[PADW];                 // the rest of KC2Wave (records, counts): same LDS footprint per wave at 12 waves
};
template <int NW> struct __attribute__((aligned(16))) Lds {
    double row[NROWTAB * 4];
    WaveT<(NW <= 12 ? 836 : 256)> wave[NW];
    char other_tables[NW <= 12 ? 16500 : 16];    // adjustment / exp tables of the kernel proper
};

#define SUMS_DECL double s00 = 0, s01 = 0, s02 = 0, s03 = 0, s04 = 0, s11 = 0, s12 = 0, s13 = 0, s14 = 0, s22 = 0, s23 = 0, s24 = 0, \
                         s33 = 0, s34 = 0, s44 = 0, d0 = 0, d1 = 0, d2 = 0, d3 = 0, d4 = 0; int nN = 0

#define DRAIN_V(x_, CLSOF) do { \
    const unsigned pe_ = (x_); \
    if (pe_) { \
        const double *rw_ = (const double *)(rowbase + ROWOF(pe_)); \
        const double pMM = rw_[0], p_M = rw_[1], pom = rw_[2]; \
        const unsigned rc_ = CLSOF(pe_); \
        const unsigned k_ = rc_ < 4u ? (rc_ ^ key) : rc_; \
        if (k_ == 1u) { MASKED; d1 += pom; s11 += pMM; s12 += p_M; s13 += p_M; s14 += p_M; } \
        if (k_ == 2u) { MASKED; d2 += pom; s12 += p_M; s22 += pMM; s23 += p_M; s24 += p_M; } \
        if (k_ == 3u) { MASKED; d3 += pom; s13 += p_M; s23 += p_M; s33 += pMM; s34 += p_M; } \
        if (k_ == 4u) { MASKED; d4 += pom; s14 += p_M; s24 += p_M; s34 += p_M; s44 += pMM; } \
        if (k_ == 5u) { MASKED; nN++; s11 += pMM; s12 += pMM; s13 += pMM; s14 += p_M; s22 += pMM; s23 += pMM; s24 += p_M; \
                        s33 += pMM; s34 += p_M; } \
    } } while (0)
#define DRAIN(CLSOF) do { DRAIN_V(pend >> 16, CLSOF); DRAIN_V(pend & 0xffffu, CLSOF); pend = 0; } while (0)

// ---- variant 0: the kernel's block as it stands
#define ROW_BASE(v_, MM_, _M_, om_) do { \
    unsigned long long mis_, odd_, coll_, rare_, m0_, m1_, m2_, m3_, t_, sv_; \
    unsigned cls_; \
    asm volatile( \
        "v_bfe_u32 %[cls], %[v], 1, 3\n\t" \
        "v_cmp_eq_u32_e64 %[m0], %[cls], %[k0]\n\t" \
        "v_cmp_eq_u32_e64 %[m1], %[cls], %[k1]\n\t" \
        "v_cmp_eq_u32_e64 %[m2], %[cls], %[k2]\n\t" \
        "v_cmp_eq_u32_e64 %[m3], %[cls], %[k3]\n\t" \
        "v_cmp_ne_u32_e64 %[t], 7, %[cls]\n\t" \
        "s_mov_b64 %[sv], exec\n\t" \
        "s_andn2_b64 %[mis], %[t], %[m0]\n\t" \
        "s_or_b64 %[t], %[m1], %[m2]\n\t" \
        "s_or_b64 %[t], %[t], %[m3]\n\t" \
        "s_andn2_b64 %[odd], %[mis], %[t]\n\t" \
        "s_mov_b64 exec, %[m0]\n\t" \
        "v_add_f64 %[s00], %[s00], %[MM]\n\t" \
        "v_add_f64 %[d0], %[d0], %[om]\n\t" \
        "v_add_f64 %[s04], %[s04], %[_M]\n\t" \
        "s_or_b64 exec, %[m0], %[m1]\n\t" \
        "v_add_f64 %[s01], %[s01], %[_M]\n\t" \
        "s_or_b64 exec, %[m0], %[m2]\n\t" \
        "v_add_f64 %[s02], %[s02], %[_M]\n\t" \
        "s_or_b64 exec, %[m0], %[m3]\n\t" \
        "v_add_f64 %[s03], %[s03], %[_M]\n\t" \
        "s_mov_b64 exec, %[mis]\n\t" \
        "v_lshl_or_b32 %[pend], %[pend], 16, %[v]\n\t" \
        "v_cmp_lt_u32_e64 %[coll], %[ffff], %[pend]\n\t" \
        "s_or_b64 %[rare], %[odd], %[coll]\n\t" \
        "s_mov_b64 exec, %[sv]" \
        : [cls] "=&v"(cls_), [m0] "=&s"(m0_), [m1] "=&s"(m1_), [m2] "=&s"(m2_), [m3] "=&s"(m3_), \
          [t] "=&s"(t_), [sv] "=&s"(sv_), [mis] "=&s"(mis_), [odd] "=&s"(odd_), [coll] "=&s"(coll_), [rare] "=&s"(rare_), \
          [s00] "+v"(s00), [s01] "+v"(s01), [s02] "+v"(s02), [s03] "+v"(s03), [s04] "+v"(s04), [d0] "+v"(d0), \
          [pend] "+v"(pend) \
        : [v] "v"(v_), [k0] "v"(k0), [k1] "v"(k1), [k2] "v"(k2), [k3] "v"(k3), [MM] "v"(MM_), [_M] "v"(_M_), [om] "v"(om_), \
          [ffff] "s"(0xffffu) \
        : "scc"); \
    if (__builtin_expect(rare_ != 0ull, 0)) { \
        if (odd_ != 0ull) { \
            if (cls_ == 4u) { MASKED; s04 += (_M_); } \
            if (cls_ == 5u) { MASKED; s00 += (MM_); s01 += (MM_); s02 += (MM_); s03 += (MM_); s04 += (_M_); } \
        } \
        if (coll_ != 0ull) DRAIN(CLS); \
    } } while (0)

// ---- variant 2: classes as f16 number classes, EXEC written by v_cmpx_class (no VALU -> SGPR -> SALU -> EXEC hop)
// entry = sign << 15 | exp << 10 | quiet << 9 | (table row + 1):  A +normal(exp 1)  C -normal  G +denormal  T -denormal
//         '*' signalling NaN  N quiet NaN  none +0
#define FCLS(v_) fcls_of(v_)
__device__ __forceinline__ unsigned fcls_of(unsigned e)
{
    const unsigned ex = (e >> 10) & 31u, sg = e >> 15;
    if ((e & 0x7fffu) == 0u) return 7u;
    if (ex == 31u) return (e & 0x200u) ? 5u : 4u;
    return ex ? sg : 2u + sg;           // A = 0, C = 1 (normals), G = 2, T = 3 (denormals)
}
#define ROW_FCLS(v_, MM_, _M_, om_) do { \
    unsigned long long coll_, odd_, sv_; \
    asm volatile( \
        "s_mov_b64 %[sv], exec\n\t" \
        "v_cmp_class_f16_e64 %[odd], %[v], 3\n\t" \
        "v_cmpx_class_f16_e32 vcc, %[v], %[M01]\n\t" \
        "v_add_f64 %[s01], %[s01], %[_M]\n\t" \
        "s_mov_b64 exec, %[sv]\n\t" \
        "v_cmpx_class_f16_e32 vcc, %[v], %[M02]\n\t" \
        "v_add_f64 %[s02], %[s02], %[_M]\n\t" \
        "s_mov_b64 exec, %[sv]\n\t" \
        "v_cmpx_class_f16_e32 vcc, %[v], %[M03]\n\t" \
        "v_add_f64 %[s03], %[s03], %[_M]\n\t" \
        "s_mov_b64 exec, %[sv]\n\t" \
        "v_cmpx_class_f16_e32 vcc, %[v], %[M0]\n\t" \
        "v_add_f64 %[s00], %[s00], %[MM]\n\t" \
        "v_add_f64 %[d0], %[d0], %[om]\n\t" \
        "v_add_f64 %[s04], %[s04], %[_M]\n\t" \
        "s_mov_b64 exec, %[sv]\n\t" \
        "v_cmpx_class_f16_e32 vcc, %[v], %[Mmis]\n\t" \
        "v_lshl_or_b32 %[pend], %[pend], 16, %[v]\n\t" \
        "v_cmp_lt_u32_e64 %[coll], %[ffff], %[pend]\n\t" \
        "s_mov_b64 exec, %[sv]" \
        : [sv] "=&s"(sv_), [odd] "=&s"(odd_), [coll] "=&s"(coll_), \
          [s00] "+v"(s00), [s01] "+v"(s01), [s02] "+v"(s02), [s03] "+v"(s03), [s04] "+v"(s04), [d0] "+v"(d0), \
          [pend] "+v"(pend) \
        : [v] "v"(v_), [M0] "v"(M0), [M01] "v"(M01), [M02] "v"(M02), [M03] "v"(M03), [Mmis] "v"(Mmis), \
          [MM] "v"(MM_), [_M] "v"(_M_), [om] "v"(om_), [ffff] "s"(0xffffu) \
        : "vcc"); \
    if (__builtin_expect((odd_ | coll_) != 0ull, 0)) { \
        if (odd_ != 0ull) { \
            const unsigned cls_ = FCLS(v_); \
            if (cls_ == 4u) { MASKED; s04 += (_M_); } \
            if (cls_ == 5u) { MASKED; s00 += (MM_); s01 += (MM_); s02 += (MM_); s03 += (MM_); s04 += (_M_); } \
        } \
        if (coll_ != 0ull) DRAIN(FCLS); \
    } } while (0)

// class -> bit of the v_cmp_class mask
__device__ __forceinline__ unsigned clsbit(unsigned c)   // c = absolute class 0..3
{
    return c == 0u ? (1u << 8) : c == 1u ? (1u << 3) : c == 2u ? (1u << 7) : (1u << 4);
}

// VAR 0: baseline   1: baseline, addends from registers (no table reads)   2: f16-class / v_cmpx form
// 3: baseline without the four-rows-ahead software prefetch of entries (plain loop)
template <int VAR, int NW>
__global__ void __launch_bounds__(NW * 64)
k_rb(double *__restrict__ out, long long *__restrict__ cyc, const uint16_t *__restrict__ ent, const double *__restrict__ tab,
     int steps4, int passes)
{
    __shared__ Lds<NW> L;
    for (int i = threadIdx.x; i < NROWTAB * 4; i += blockDim.x) L.row[i] = tab[i];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    auto &W = L.wave[w];
    for (int i = lane; i < 64 * 64; i += 64) W.plane[i >> 6][i & 63] = ent[(size_t)((blockIdx.x * NW + w) & 63) * 4096 + i];
    __syncthreads();
    const char *rowbase = (const char *)L.row;
    SUMS_DECL;
    unsigned pend = 0;
    const unsigned key = (unsigned)(lane & 3);
    const unsigned k0 = key, k1 = key ^ 1u, k2 = key ^ 2u, k3 = key ^ 3u;
    const unsigned M0 = clsbit(k0), M01 = M0 | clsbit(k1), M02 = M0 | clsbit(k2), M03 = M0 | clsbit(k3);
    const unsigned Mmis = 0x19bu & ~M0;
    (void)M0; (void)M01; (void)M02; (void)M03; (void)Mmis; (void)k0; (void)k1; (void)k2; (void)k3;
    const int rowL = (lane * 5) & 7;
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    for (int p = 0; p < passes; p++) {
        const uint16_t *basep = &W.plane[0][0] + ((rowL + p) & 15) * 64 + lane;
        unsigned vn[4];
        #pragma unroll
        for (int u = 0; u < 4; u++) vn[u] = basep[u * 64];
        for (int j0 = 0; j0 < steps4; j0 += 4) {
            unsigned ve[4];
            double rMM[4], r_M[4], rom[4];
            #pragma unroll
            for (int u = 0; u < 4; u++) {
                ve[u] = vn[u];
                if (VAR == 1) { rMM[u] = -0.0001 * (double)(p + 1); r_M[u] = -0.69 - 1e-9 * (double)j0; rom[u] = 0.9999; asm volatile("" : "+v"(rMM[u]), "+v"(r_M[u]), "+v"(rom[u])); }
                else if (VAR == 2) {
                    const double *rw = (const double *)(rowbase + ((ve[u] & 0xffu) << 5) - 32);
                    rMM[u] = rw[0]; r_M[u] = rw[1]; rom[u] = rw[2];
                } else {
                    const double *rw = (const double *)(rowbase + (ve[u] & ROWMASK));
                    rMM[u] = rw[0]; r_M[u] = rw[1]; rom[u] = rw[2];
                }
            }
            if (j0 + 4 < steps4) {
                #pragma unroll
                for (int u = 0; u < 4; u++) vn[u] = basep[(j0 + 4 + u) * 64];
            }
            #pragma unroll
            for (int u = 0; u < 4; u++) {
                if (VAR == 2) {
#define ROWOF(pe_) ((((pe_) & 0xffu) << 5) - 32)
                    ROW_FCLS(ve[u], rMM[u], r_M[u], rom[u]);
#undef ROWOF
                } else {
#define ROWOF(pe_) ((pe_) & ROWMASK)
                    ROW_BASE(ve[u], rMM[u], r_M[u], rom[u]);
#undef ROWOF
                }
            }
        }
    }
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    if (VAR == 2) {
#define ROWOF(pe_) ((((pe_) & 0xffu) << 5) - 32)
        DRAIN(FCLS);
#undef ROWOF
    } else {
#define ROWOF(pe_) ((pe_) & ROWMASK)
        DRAIN(CLS);
#undef ROWOF
    }
    const size_t o = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 21;
    out[o] = s00; out[o + 1] = s01; out[o + 2] = s02; out[o + 3] = s03; out[o + 4] = s04; out[o + 5] = s11; out[o + 6] = s12;
    out[o + 7] = s13; out[o + 8] = s14; out[o + 9] = s22; out[o + 10] = s23; out[o + 11] = s24; out[o + 12] = s33; out[o + 13] = s34;
    out[o + 14] = s44; out[o + 15] = d0; out[o + 16] = d1; out[o + 17] = d2; out[o + 18] = d3; out[o + 19] = d4; out[o + 20] = (double)nN;
    if (lane == 0) cyc[blockIdx.x * NW + w] = t1 - t0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static uint16_t enc_base(unsigned rowidx, unsigned cls) { return (uint16_t)((rowidx << 5) | (cls << 1) | 1u); }
static uint16_t enc_fcls(unsigned rowidx, unsigned cls)
{
    const unsigned pl = rowidx + 1;
    switch (cls) {
    case 0: return (uint16_t)((1u << 10) | pl);
    case 1: return (uint16_t)(0x8000u | (1u << 10) | pl);
    case 2: return (uint16_t)pl;
    case 3: return (uint16_t)(0x8000u | pl);
    case 4: return (uint16_t)((31u << 10) | pl);
    case 5: return (uint16_t)((31u << 10) | 0x200u | pl);
    default: return 0;
    }
}

template <int VAR, int NW>
static void run(const char *name, const std::vector<uint16_t> &ent, const double *d_tab, int steps4, int passes, std::vector<double> *sums)
{
    const int nwg = 256;
    double *d_out; long long *d_cyc; uint16_t *d_ent;
    CK(hipMalloc(&d_out, (size_t)nwg * NW * 64 * 21 * 8));
    CK(hipMalloc(&d_cyc, (size_t)nwg * NW * 8));
    CK(hipMalloc(&d_ent, ent.size() * 2));
    CK(hipMemcpy(d_ent, ent.data(), ent.size() * 2, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_rb<VAR, NW>), dim3(nwg), dim3(NW * 64), 0, 0, d_out, d_cyc, d_ent, d_tab, steps4, passes);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    std::vector<long long> cyc((size_t)nwg * NW);
    CK(hipMemcpy(cyc.data(), d_cyc, cyc.size() * 8, hipMemcpyDeviceToHost));
    std::sort(cyc.begin(), cyc.end());
    const double med = (double)cyc[cyc.size() / 2];
    const double rows = (double)steps4 * passes;
    // s_memtime ticks at 100 MHz on gfx9 (constant clock), so convert through wall time as well
    printf("%-28s NW=%2d  ms=%8.3f  memtime/rowstep=%8.2f  ns/rowstep(wall)=%8.2f  rowsteps/s/CU=%.3e\n", name, NW, best, med / rows,
           best * 1e6 / rows, rows * NW / (best * 1e-3));
    if (sums) { sums->resize((size_t)nwg * NW * 64 * 21); CK(hipMemcpy(sums->data(), d_out, sums->size() * 8, hipMemcpyDeviceToHost)); }
    CK(hipFree(d_out)); CK(hipFree(d_cyc)); CK(hipFree(d_ent));
}

int main(int argc, char **argv)
{
    const int steps4 = argc > 1 ? atoi(argv[1]) : 36, passes = argc > 2 ? atoi(argv[2]) : 2000;
    const double mism = argc > 3 ? atof(argv[3]) : 0.01;
    std::vector<double> tab(NROWTAB * 4);
    for (int i = 0; i < NROWTAB; i++) { tab[i * 4] = -1e-4 * (i + 1); tab[i * 4 + 1] = -0.69 - 1e-3 * i; tab[i * 4 + 2] = 1.0 - 1e-3 * i; tab[i * 4 + 3] = 0; }
    double *d_tab; CK(hipMalloc(&d_tab, tab.size() * 8)); CK(hipMemcpy(d_tab, tab.data(), tab.size() * 8, hipMemcpyHostToDevice));
    // 64 planes of 64 x 64 entries; lane's key = lane & 3, so class = key with probability 1 - mism
    std::vector<uint16_t> eb(64 * 4096), ef(64 * 4096);
    srand(12345);
    const unsigned quals[4] = {2, 11, 25, 37};
    for (size_t i = 0; i < eb.size(); i++) {
        const unsigned lane = (unsigned)(i & 63);
        const double r = rand() / (double)RAND_MAX;
        unsigned cls = lane & 3u;
        if (r < mism) cls = (unsigned)(rand() % 4);
        else if (r < mism + 0.0005) cls = 4u + (unsigned)(rand() & 1);
        else if (r < mism + 0.0005 + 0.15) cls = 7u;
        const int qr = rand() % 100;
        const unsigned q = quals[qr < 1 ? 0 : qr < 5 ? 1 : qr < 20 ? 2 : 3];
        eb[i] = enc_base(q, cls); ef[i] = enc_fcls(q, cls);
    }
    std::vector<double> s0, s2;
    printf("steps4=%d passes=%d mismatch=%.4f\n", steps4, passes, mism);
    run<0, 12>("baseline", eb, d_tab, steps4, passes, &s0);
    run<2, 12>("fclass/cmpx", ef, d_tab, steps4, passes, &s2);
    size_t bad = 0;
    for (size_t i = 0; i < s0.size(); i++) if (memcmp(&s0[i], &s2[i], 8)) bad++;
    printf("fclass sums vs baseline: %zu of %zu differ\n", bad, s0.size());
    run<1, 12>("baseline, no table reads", eb, d_tab, steps4, passes, nullptr);
    run<0, 4>("baseline", eb, d_tab, steps4, passes, nullptr);
    run<0, 8>("baseline", eb, d_tab, steps4, passes, nullptr);
    run<0, 16>("baseline", eb, d_tab, steps4, passes, nullptr);
    run<2, 4>("fclass/cmpx", ef, d_tab, steps4, passes, nullptr);
    run<2, 8>("fclass/cmpx", ef, d_tab, steps4, passes, nullptr);
    run<2, 16>("fclass/cmpx", ef, d_tab, steps4, passes, nullptr);
    return 0;
}
