/*
 * kcol2.h -- k_columns2: the column kernel (calculate_consensus_pileup snp:533-797 + the column heuristics
 * snp:1438-1773), restructured around a tile-sized pileup matrix staged in LDS.
 *
 * k_columns (kernels.h) walks a tile's candidate records one by one: a scalar 32-byte record load, a scalar
 * decode, two dependent byte gathers and four exec-masked blocks of FP64 adds per (tile, record) -- 65 VALU +
 * 62 SALU wave-instructions per record.  Here a wave handles a 64-column tile, 64 candidate records at a time,
 * in three phases:
 *
 *   stage     lanes = candidate records: one coalesced 32-byte record load per lane (long-read batches, where
 *             most ops of a read lie elsewhere: 8 bytes first, the rest only for the hits); the records that
 *             reach the tile are compacted in file order (ballot + prefix popcount) into the wave's LDS.
 *   produce   lanes = (record, 16-column segment), 16 records per step, the global loads of two steps in flight
 *             together.  A lane loads the 16 bases and 16 qualities of its segment (aligned dwords + byte
 *             funnel) and writes 16 two-byte entries of the wave's LDS matrix plane[row][column]:
 *                 entry = quality << 5 | class          (class 7 = the row does not reach the column)
 *             class: A C G T = 0..3 (the high nibble k_p0_basequal left in the base byte), 4 = '*' (D op), 5 = N.
 *   consume   lanes = columns.  Row by row (file order): one ds_read_u16 and one 24-byte table row per lane
 *             (both issued a group of four rows ahead), no global loads, no scalar decode.  The LDS table is
 *             indexed by the RAW quality: rows composed for the batch's dominant mapq (cap_quality, the
 *             (mapq, qual) adjustment and the 1..100 clamp folded in, DTables.adjc).  Rows of reads with another
 *             mapq (flagged per row) take one more LDS hop through a compact per-mapq table; rows with a D/N
 *             op, an indel behind the op or no SEQ ("rare", ~3 %) look at their record (in LDS) again.
 *
 * FP64 adds.  Base classes are numbered PER COLUMN relative to a key base (class 0; majority vote of a few rows
 * that span the tile): class k = idx(base) ^ idx(key).  XOR by a constant permutes {A,C,G,T}, the genotype
 * table (snp:656-683) is symmetric under it, col_unpermute() undoes it.  With sums indexed
 *   S: 00 01 02 03 04 | 11 12 13 14 | 22 23 24 | 33 34 | 44      (4 = '*'),  sC: 0 1 2 3 4
 * an entry of class 0 touches S00 S01 S02 S03 S04 sC0; an entry of class k = 1..4 touches S0k -- and sCk, Skk,
 * Skj (j != 0, k), which NO class-0 entry ever touches.  So every row does its "inline" part at once, in file
 * order, with per-add exec masks:
 *   S00 += MM [class 0]   sC0 += om [0]   S0k += _M [class 0 or k], k = 1..4   (N: MM MM MM MM _M)
 * and a mismatching lane parks the rest of its entry in a one-deep pending slot (a VGPR).  The parked part only
 * touches sums that inline parts never touch, so deferring it reorders no sum; parked parts are applied in their
 * own order: a lane that needs its slot again forces a wave-level drain first.  A drain costs about two rows and
 * happens every ~7 rows at 3 % mismatches (once per row only under a real heterozygous site).  Every sum still
 * receives exactly its own addends in file order: bit-identical to the reference, 6 FP64
 * wave-instructions per row instead of 24.
 */
#ifndef CRUMBLE_KCOL2_H
#define CRUMBLE_KCOL2_H

#ifndef KC2_R
#define KC2_R        64            /* candidates screened per pass = records staged in LDS (56 with 14 waves: see profiles/NOTES_r04.md) */
#endif
#ifndef KC2_PR
#define KC2_PR       64            /* rows of the matrix (a pass with more hits is produced / consumed in groups) */
#endif
#ifndef KC2_WAVES
#define KC2_WAVES    12            /* one workgroup per CU: 3 waves per SIMD, <= 168 VGPRs */
#endif
#define KC2_THREADS  (KC2_WAVES * 64)
#ifndef KC2_SLOTS
#define KC2_SLOTS    64            /* mapq values with a row of the compact adjustment table in LDS */
#endif
#ifndef KC2_LDS_EXP
#define KC2_LDS_EXP  1             /* cons_finalize's exp tables in LDS */
#endif
#ifndef KC2_WGDEAL
#define KC2_WGDEAL   1             /* the waves of a workgroup take the tiles of ONE unit between them, tile by tile: tiles in
                                    * flight on a CU are neighbours, their reads' cache lines are shared while they are hot */
#endif
#ifndef KC2_FASTFIN
#define KC2_FASTFIN  1             /* tiles whose columns are all "common" take cons_finalize_fast (needs KC2_LDS_EXP) */
#endif
#if KC2_PR != KC2_R
#error "the consumer's row windows need the whole pass in the matrix (KC2_PR == KC2_R)"
#endif
#define KC2_RARE     0x10000u      /* row flags in rowinfo[]: bits 0-7 mapq, 8-15 slot */
#define KC2_LOWMQ    0x20000u
#define KC2_FULL     0x40000u      /* a match op spanning the whole tile: takes part in the key vote */
#define KC2_NONHOT   0x80000u      /* mapq != the batch's dominant one (CONS_MQUAL only) */
#define KC2_NROWTAB  (128 + 101)
#define KC2_ROWMASK  0x1fe0u       /* matrix entry: table row byte offset (bits 5-12) | class (bits 1-3; 7 = the record does
                                    * not reach the column) | bit 0 set (never zero) | call_preserve input of a translated
                                    * row (bits 13-14) */
#define KC2_CLS(v_)  (((v_) >> 1) & 7u)

#ifdef CRUMBLE_EMU
static unsigned long long kc2_emu_fin[2];      /* test hook (emulator only): tiles through cons_finalize_fast / the general finalize */
#define KC2_EMU_COUNT(i_) do { if ((threadIdx.x & 63) == 0) __atomic_fetch_add(&kc2_emu_fin[i_], 1ull, __ATOMIC_RELAXED); } while (0)
#define KC2_SLEEP() do { } while (0)       /* (the emulator's fibers switch at wave rendezvous only: a lock is never seen held) */
#define KC2_FENCE_WG() __atomic_thread_fence(__ATOMIC_SEQ_CST)
#define WAVE_SYNC() emu_wave_sync()
__device__ __forceinline__ uint32_t kc2_alignbyte(uint32_t hi, uint32_t lo, unsigned sh)
{ return sh ? (lo >> (8u * sh)) | (hi << (32u - 8u * sh)) : lo; }
__device__ __forceinline__ uint32_t kc2_perm(uint32_t s0, uint32_t s1, uint32_t sel)      /* v_perm_b32 */
{
    const uint64_t src = ((uint64_t)s0 << 32) | s1;
    uint32_t d = 0;
    for (int i = 0; i < 4; i++) {
        const unsigned s = (sel >> (8 * i)) & 255u;
        const unsigned byte = s < 8 ? (unsigned)((src >> (8 * s)) & 255u) : (s == 0x0c ? 0u : 0xffu);   /* (8..11 unused here) */
        d |= byte << (8 * i);
    }
    return d;
}
#define KC2_MASKED
#define KC2_PIN(x)
#define KC2_UNDEF(x) ((x) = 0)
/* inclusive scans across the wave's lanes */
#define KC2_EMU_SCAN(name, OP) \
__device__ __forceinline__ int name(int v) \
{ const int ln = threadIdx.x & 63; for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(v, d); if (ln >= d) v = OP; } return v; }
KC2_EMU_SCAN(kc2_scan_add, v + o)
KC2_EMU_SCAN(kc2_scan_max, (o > v ? o : v))
KC2_EMU_SCAN(kc2_scan_min, (o < v ? o : v))
#else
#define KC2_EMU_COUNT(i_) do { } while (0)
#define KC2_SLEEP() __builtin_amdgcn_s_sleep(1)
#define KC2_FENCE_WG() __threadfence_block()
/* LDS traffic between the lanes of ONE wave: the hardware keeps a wave's DS operations in order; the fence only
 * stops the compiler from moving them across */
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                         __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
__device__ __forceinline__ uint32_t kc2_alignbyte(uint32_t hi, uint32_t lo, unsigned sh)
{ return __builtin_amdgcn_alignbyte(hi, lo, sh); }
__device__ __forceinline__ uint32_t kc2_perm(uint32_t s0, uint32_t s1, uint32_t sel)
{ return __builtin_amdgcn_perm(s0, s1, sel); }
/* hipcc turns `if (m) s += x;` into an unconditional add plus two v_cndmask_b32 per double; an empty volatile asm
 * in the block keeps it an exec-masked block (one s_and_saveexec for several adds, skipped when no lane is in it) */
#define KC2_MASKED asm volatile("")
#define KC2_PIN(x) asm volatile("" : "+v"(x))      /* the value exists HERE (a load above it cannot sink past the block) */
#define KC2_UNDEF(x) asm volatile("" : "=v"(x))    /* any value (no instruction): lanes that load nothing compute on it, masked out later */
/* inclusive scans across the wave's lanes in six DPP steps (row_shr 1/2/4/8 inside the rows of 16, then row_bcast:15
 * into rows 1 and 3 and row_bcast:31 into rows 2 and 3): VALU only, where __shfl_up is a chain of ds_bpermute
 * round trips.  Lanes a step has no source for take the operation's identity. */
#define KC2_DPP(id_, v_, ctrl_, rows_) __builtin_amdgcn_update_dpp((int)(id_), (v_), (ctrl_), (rows_), 0xf, false)
#define KC2_DPP_SCAN(name, ID, OP) \
__device__ __forceinline__ int name(int v) \
{ \
    int o; \
    o = KC2_DPP(ID, v, 0x111, 0xf); v = OP; \
    o = KC2_DPP(ID, v, 0x112, 0xf); v = OP; \
    o = KC2_DPP(ID, v, 0x114, 0xf); v = OP; \
    o = KC2_DPP(ID, v, 0x118, 0xf); v = OP; \
    o = KC2_DPP(ID, v, 0x142, 0xa); v = OP; \
    o = KC2_DPP(ID, v, 0x143, 0xc); v = OP; \
    return v; \
}
KC2_DPP_SCAN(kc2_scan_add, 0, v + o)
KC2_DPP_SCAN(kc2_scan_max, (-0x7fffffff - 1), (o > v ? o : v))
KC2_DPP_SCAN(kc2_scan_min, 0x7fffffff, (o < v ? o : v))
#endif

struct __attribute__((aligned(4))) KC2U4 { uint32_t x, y, z, w; };     /* 16 bytes at a 4-byte aligned address */

struct __attribute__((aligned(16))) KC2Wave {      /* one wave's LDS */
    uint16_t plane[KC2_PR][64];
    uint4    recs[KC2_R][2];                       /* SegRec: {pos, end, so.lo, so.hi} {lq, qoff, indel_after, info} */
    uint32_t rowinfo[KC2_R];
    uint32_t cnt[68];                              /* difference array over the tile's columns: all records (low half) and
                                                    * low-mapq records (high half) in one word -- the prefix sums of both are
                                                    * counts in 0..64, so the packed sum splits again after the scan */
    uint32_t ht[64];                               /* per column: head/tail entries (low half), clipped ones (high half) */
    int32_t  rowhi[64], rowlo[64];                 /* per column: 1 + the last row that starts there / the first row that
                                                    * ends there: the lanes' row windows */
};

struct __attribute__((aligned(16))) KC2Lds {
    double   row[KC2_NROWTAB * 4];
    double   e[KC2_LDS_EXP ? CONS_NEXP : 2];
    unsigned stat[32];
    uint8_t  adjc[KC2_SLOTS * 128];
    uint8_t  pv[128];
    uint4    mask_lt[17], mask_ge[17];             /* producer: byte k of [a] = 0x08 | (k < a ? 0x70 : 0); of [b] = (k >= b ? 0x70 : 0) */
    unsigned long long deal;                       /* KC2_WGDEAL: the workgroup's current unit, end tile << 32 | next tile */
    unsigned deal_lock, deal_done;
    KC2Wave  wave[KC2_WAVES];
};

/* The 15 genotype sums and 5 discrepancy sums are NAMED LOCAL SCALARS of the kernel (permuted numbering, see the
 * header), and the two routines below are macros over them.  Not arrays, not a struct passed by reference: hipcc
 * merges the parked part's four look-alike class blocks into one block with computed addresses, and whatever is
 * addressed that way lives in scratch memory (first builds: 150 scratch instructions, 15.5 ms). */
#define KC2_SUMS_DECL \
    double s00 = 0, s01 = 0, s02 = 0, s03 = 0, s04 = 0, s11 = 0, s12 = 0, s13 = 0, s14 = 0, s22 = 0, s23 = 0, s24 = 0, \
           s33 = 0, s34 = 0, s44 = 0, d0 = 0, d1 = 0, d2 = 0, d3 = 0, d4 = 0; \
    int nN = 0

/* the parked part of a mismatching entry x_ = its matrix entry (bit 15 | table row byte offset | class), 0 = none */
#define KC2_DRAIN_V(x_) do { \
    const unsigned pe_ = (x_); \
    if (pe_) { \
        const double *rw_ = (const double *)(rowbase + (pe_ & KC2_ROWMASK)); \
        const double pMM = rw_[0], p_M = rw_[1], pom = rw_[2]; \
        const unsigned rc_ = KC2_CLS(pe_); \
        const unsigned k_ = rc_ < 4u ? (rc_ ^ key) : rc_; \
        if (k_ == 1u) { KC2_MASKED; d1 += pom; s11 += pMM; s12 += p_M; s13 += p_M; s14 += p_M; } \
        if (k_ == 2u) { KC2_MASKED; d2 += pom; s12 += p_M; s22 += pMM; s23 += p_M; s24 += p_M; } \
        if (k_ == 3u) { KC2_MASKED; d3 += pom; s13 += p_M; s23 += p_M; s33 += pMM; s34 += p_M; } \
        if (k_ == 4u) { KC2_MASKED; d4 += pom; s14 += p_M; s24 += p_M; s34 += p_M; s44 += pMM; } \
        if (k_ == 5u) { KC2_MASKED; nN++; s11 += pMM; s12 += pMM; s13 += pMM; s14 += p_M; s22 += pMM; s23 += pMM; s24 += p_M; \
                        s33 += pMM; s34 += p_M; } \
    } } while (0)
/* a lane parks up to two entries: the older one in the high half (file order of the adds) */
#define KC2_DRAIN() do { KC2_DRAIN_V(pend >> 16); KC2_DRAIN_V(pend & 0xffffu); pend = 0; } while (0)

/* One row of the matrix for this lane's column: entry v_ (table row byte offset | class; class 7 = not reached),
 * the table row's MM_, _M_, om_.  The inline part of the entry (header) plus the pending-slot bookkeeping.
 * On the GPU the common path is one hand-written block: hipcc's version of the same statements came to ~25 VALU,
 * ~39 SALU and ~14 branches per row, and at three waves per SIMD the kernel was bound by instruction issue
 * (710 cycles per row and wave); the block is 13 VALU + 10 SALU and no branch.  It changes EXEC inside and
 * restores it; every mask is a wave-uniform SGPR pair. */
#ifdef CRUMBLE_EMU
#define KC2_ROW(v_, MM_, _M_, om_) do { \
    const unsigned cls_ = KC2_CLS(v_); \
    const bool cov_ = cls_ != 7u; \
    const bool m0_ = cls_ == k0, m1_ = cls_ == k1, m2_ = cls_ == k2, m3_ = cls_ == k3; \
    const unsigned long long B0_ = __ballot(m0_), B123_ = __ballot(m1_ || m2_ || m3_), Bc_ = __ballot(cov_); \
    const unsigned long long mis_ = Bc_ & ~B0_; \
    if (mis_ & ~B123_) {                                  /* a '*' or an N somewhere in the row (rare) */ \
        if (cls_ == 4u) { s04 += (_M_); } \
        if (cls_ == 5u) { s00 += (MM_); s01 += (MM_); s02 += (MM_); s03 += (MM_); s04 += (_M_); } \
    } \
    if (m0_) { s00 += (MM_); d0 += (om_); s04 += (_M_); } \
    if (m0_ || m1_) { s01 += (_M_); } \
    if (m0_ || m2_) { s02 += (_M_); } \
    if (m0_ || m3_) { s03 += (_M_); } \
    if (mis_) { \
        const bool ml_ = cov_ && !m0_; \
        if (ml_) pend = (pend << 16) | (v_); \
        if (__ballot(ml_ && pend > 0xffffu)) KC2_DRAIN(); \
    } } while (0)
#else
#define KC2_ROW(v_, MM_, _M_, om_) do { \
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
        "v_lshl_or_b32 %[pend], %[pend], 16, %[v]\n\t"          /* park this entry behind the lane's older one */ \
        "v_cmp_lt_u32_e64 %[coll], %[ffff], %[pend]\n\t"        /* a lane with two parked entries: drain */ \
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
        if (odd_ != 0ull) {                               /* a '*' or an N somewhere in the row */ \
            if (cls_ == 4u) { KC2_MASKED; s04 += (_M_); } \
            if (cls_ == 5u) { KC2_MASKED; s00 += (MM_); s01 += (MM_); s02 += (MM_); s03 += (MM_); s04 += (_M_); } \
        } \
        if (coll_ != 0ull) KC2_DRAIN(); \
    } } while (0)
#endif

#ifdef KC2_EXP_STAMP      /* diagnostic build: cycles per phase, summed over waves, into the unused stats slots 24..29 */
#define KC2_STAMP(k_) do { const long long t_ = (long long)__builtin_amdgcn_s_memtime(); tacc[k_] += t_ - tlast; tlast = t_; } while (0)
#else
#define KC2_STAMP(k_) do { } while (0)
#endif

/* a staged record (r0, r1 at matrix row `slot` of the pass): the record itself, the difference arrays, the row window
 * bounds, the row's flags and the head / tail counts */
#define KC2_STAGE_HIT() \
                if (ov) { \
                    W.recs[slot][0] = r0; W.recs[slot][1] = r1; \
                    int a = (int)r0.x - c0, b = (int)r0.y - c0; \
                    a = a < 0 ? 0 : a; b = b > 64 ? 64 : b; \
                    atomicMax(&W.rowhi[a], slot + 1);                    /* (op starts are not sorted: a read's later ops) */ \
                    atomicMin(&W.rowlo[b - 1], slot); \
                    lowrec = (int)(r1.w & 255u) <= min_mqual; \
                    { const uint32_t one = lowrec ? 0x10001u : 1u; atomicAdd(&W.cnt[a], one); atomicAdd(&W.cnt[b], 0u - one); } \
                    {   /* the row's flags for the consumer */ \
                        const unsigned info = r1.w, ty = SR_TYPE(info); \
                        /* D/N op, no SEQ, or an indel behind the op -- which only its last column sees (long reads: \
                         * nearly every op ends in one, but rarely inside this tile) */ \
                        const bool rare = ty != SEG_M || r1.x == 0u || (r1.z != 0u && (int)r0.y - 1 <= c1); \
                        const bool full = ty == SEG_M && (int)r1.x > 0 && (int)r0.x <= c0 && (int)r0.y > c1; \
                        W.rowinfo[slot] = (info & 0xffu) | ((info >> 8) & 0xff00u) | (rare ? KC2_RARE : 0u) | \
                                          (full ? KC2_FULL : 0u) | ((MQ && (int)(info & 255u) != hot_mq) ? KC2_NONHOT : 0u); \
                    } \
                    /* is_head / is_tail bookkeeping (snp:1701-1706): two columns per read; a ref-skip entry is not \
                     * counted (snp:1696); head and tail on one column count once */ \
                    const unsigned info = r1.w; \
                    if (SR_TYPE(info) != SEG_N && (info & (SR_FIRST | SR_LAST))) { \
                        const int hc = (int)r0.x - c0, tc = (int)r0.y - 1 - c0; \
                        const bool hd = (info & SR_FIRST) && hc >= 0, tl = (info & SR_LAST) && tc < 64; \
                        if (hd && tl && hc == tc) \
                            atomicAdd(&W.ht[hc], 1u + ((info & (SR_HEADCLIP | SR_TAILCLIP)) ? 0x10000u : 0u)); \
                        else { \
                            if (hd) atomicAdd(&W.ht[hc], 1u + ((info & SR_HEADCLIP) ? 0x10000u : 0u)); \
                            if (tl) atomicAdd(&W.ht[tc], 1u + ((info & SR_TAILCLIP) ? 0x10000u : 0u)); \
                        } \
                    } \
                }

/* The next unit for a workgroup.  KC2_XCD = 1 (round-5 experiment, off): chunks of KC2_XCD_CHUNK consecutive units are dealt
 * round-robin to eight groups of workgroups (blockIdx % 8: the blocks that share an XCD under round-robin placement), each
 * group with its own counter, so that the 32 workgroups of an XCD work on neighbouring units and the reads that straddle a
 * unit border are in THEIR L2.  Measured on one box: chr20 4.531 (one counter) / 4.536 (this) / 4.549 (groups of blockIdx / 32)
 * -- the re-fetch across XCDs is not what the kernel waits for.  Two things learnt on the way: eight contiguous eighths of the
 * contig instead of interleaved chunks, and ANY look at the counter before the add (a volatile global load under the
 * workgroup's lock), each cost 2.7-3.2 ms of 4.5. */
#ifndef KC2_XCD
#define KC2_XCD 0
#endif
#ifndef KC2_XCD_CHUNK
#define KC2_XCD_CHUNK 32
#endif
__device__ __forceinline__ int kc2_next_unit(unsigned *__restrict__ ctr, int n_units)
{
#if KC2_XCD
    const int grp = (int)(blockIdx.x & 7u);
    const int C = KC2_XCD_CHUNK, full = n_units / (8 * C), rem = n_units % (8 * C);
    for (int j = 0; j < 8; j++) {
        const int y = (grp + j) & 7;
        int extra = rem - y * C;
        extra = extra < 0 ? 0 : extra > C ? C : extra;
        const unsigned len = (unsigned)(full * C + extra);                   /* units u with (u / C) % 8 == y */
        if (len == 0u) continue;
        const unsigned k = atomicAdd(&ctr[y], 1u);
        if (k < len) return (int)(((k / (unsigned)C) * 8u + (unsigned)y) * (unsigned)C + k % (unsigned)C);
    }
    return n_units;
#else
    return (int)atomicAdd(ctr, 1u);
#endif
}

/* MQ: the active consensus uses CONS_MQUAL (mode B, snp:1535); ANYPRES: -k/-K/-y pbccs set preserve_qual[];
 * MULTI: long reads (many ops per read: a tile's records come from the batch's hit list); WIN: the batch is a window
 * of a contig (crumble_hip_set_window: columns outside [col_lo, col_hi) read as empty, columns from stat_hi on stay
 * out of the statistics) -- its own instances, the whole-contig ones carry none of it */
template <bool MQ, bool ANYPRES, bool MULTI, bool WIN>
__global__ void __launch_bounds__(KC2_THREADS)
k_columns2(DBatch B, const DParams *__restrict__ P, const DTables *__restrict__ T,
           uint32_t *__restrict__ colrec, int32_t *__restrict__ nplp_out,
           int32_t *__restrict__ score_out, int32_t *__restrict__ trigcnt, const uint32_t *__restrict__ ht,
           u64 *stats, unsigned *__restrict__ tile_ctr)
{
    /* ONE shared object, the table first: its rows then sit at LDS offsets below 64 KB and the consumer's row reads
     * need no address add (the offset is an immediate) */
    __shared__ KC2Lds L;
    double *const sRow = L.row;            /* [index] -> {MM, _M, 1 - q2p, preserve_qual}: 32-byte rows.  0..127: a raw
                                            * quality of a read with the dominant mapq (or of any read without
                                            * CONS_MQUAL); 128 + a: adjusted quality a */
    uint8_t *const sAdjC = L.adjc;         /* [slot][raw quality] -> adjusted quality, other mapqs */
    uint8_t *const sPv = L.pv;             /* pvc[] (call_preserve input of a raw quality) */
    double *const sE = L.e;                /* the reachable halves of the exp tables (cons_finalize) */
    unsigned *const sStat = L.stat;
    KC2Wave *const sWave = L.wave;
    const int hot_mq = MQ ? B.hot_mq : -1;
    if (threadIdx.x < 32) sStat[threadIdx.x] = 0;
    for (int i = threadIdx.x; i < KC2_NROWTAB; i += blockDim.x) {
        const int a = i >= 128 ? i - 128 : (MQ ? (int)T->adjc[hot_mq * 256 + i] : (int)T->capc[i]);
        sRow[i * 4] = T->MMd[a]; sRow[i * 4 + 1] = T->_Md[a]; sRow[i * 4 + 2] = T->omq[a];
        /* call_preserve input of the quality (snp:611-623) as the bits an entry of class 0 adds to st.cp: bit 0 for a
         * preserved quality, bit 8 as well for a really-kept one; the consumer shifts them by the entry's class */
        const unsigned pvl = i < 128 ? (unsigned)T->pvc[i] : 0u;
        long long pvb = (long long)((pvl ? 1u : 0u) | (pvl > 1u ? 0x100u : 0u));
        memcpy(&sRow[i * 4 + 3], &pvb, 8);
    }
    if (MQ)
        for (int i = threadIdx.x; i < B.n_slots * 128; i += blockDim.x)
            sAdjC[i] = T->adjc[(int)B.slot_mq[i >> 7] * 256 + (i & 127)];
    if (threadIdx.x < 128) sPv[threadIdx.x] = T->pvc[threadIdx.x];
    if (threadIdx.x < 17 * 16) {
        const int x = threadIdx.x >> 4, k = threadIdx.x & 15;
        ((uint8_t *)L.mask_lt)[threadIdx.x] = (uint8_t)(0x08 | (k < x ? 0x70 : 0));
        ((uint8_t *)L.mask_ge)[threadIdx.x] = (uint8_t)(k >= x ? 0x70 : 0);
    }
    if (KC2_LDS_EXP)
        for (int i = threadIdx.x; i < CONS_NEXP; i += blockDim.x) sE[i] = i <= 500 ? T->e_tab2[i] : T->e_tab[i - 501];
    __syncthreads();

    const int lane = threadIdx.x & 63;
    KC2Wave &W = sWave[threadIdx.x >> 6];
    const char *rowbase = (const char *)L.row;
    const int min_mqual = P->min_mqual;
    const int rsub = lane >> 2, seg = lane & 3;
    const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;     /* lanes below this one */
#ifdef KC2_EXP_STAMP
    long long tacc[6] = {0, 0, 0, 0, 0, 0}, tlast = (long long)__builtin_amdgcn_s_memtime();
    long long npass = 0, nstep = 0;
#endif

#if KC2_WGDEAL
    if (threadIdx.x == 0) { L.deal = 0ull; L.deal_lock = 0u; L.deal_done = 0u; }
    __syncthreads();
    for (;;) {
    int tile;
    {
        /* next tile of the workgroup's unit (one 64-bit LDS atomic: a consistent next / end pair); an exhausted unit is
         * replaced by whichever wave gets the lock -- it takes the new unit's first tile itself */
        unsigned lo32 = 0, hi32 = 0, st = 0;
        for (;;) {
            if (lane == 0) {
                const unsigned long long d = atomicAdd(&L.deal, 1ull);
                lo32 = (unsigned)d; hi32 = (unsigned)(d >> 32);
                st = lo32 < hi32 ? 1u : 0u;
                if (!st) {
                    if (L.deal_done) st = 2u;
                    else if (atomicCAS(&L.deal_lock, 0u, 1u) == 0u) {
                        const unsigned long long d2 = *(volatile unsigned long long *)&L.deal;     /* refilled meanwhile? */
                        if ((unsigned)d2 < (unsigned)(d2 >> 32)) st = 3u;
                        else {
                            const int g = kc2_next_unit(tile_ctr, B.n_units);
                            if (g >= B.n_units) { L.deal_done = 1u; st = 2u; }
                            else {
                                const unsigned tb = (unsigned)g * (unsigned)B.unit_tiles;
                                const unsigned te = tb + (unsigned)B.unit_tiles < (unsigned)B.n_tiles ? tb + (unsigned)B.unit_tiles : (unsigned)B.n_tiles;
                                atomicExch(&L.deal, ((unsigned long long)te << 32) | (tb + 1u));
                                lo32 = tb; hi32 = te; st = tb < te ? 1u : 3u;
                            }
                        }
                        KC2_FENCE_WG();
                        atomicExch(&L.deal_lock, 0u);
                    } else {
                        while (*(volatile unsigned *)&L.deal_lock) KC2_SLEEP();
                        st = 3u;
                    }
                }
            }
            st = (unsigned)READLANE(st, 0);
            if (st != 3u) break;                                         /* 3: look again */
        }
        if (st == 2u) break;
        tile = READLANE(lo32, 0);
    }
    KC2_STAMP(0);
    {
#else
    for (;;) {
    int t_begin, t_end;
    {
        int g = 0;
        if (lane == 0) g = kc2_next_unit(tile_ctr, B.n_units);
        g = READLANE(g, 0);
        if (g >= B.n_units) break;
        t_begin = g * B.unit_tiles; t_end = t_begin + B.unit_tiles < B.n_tiles ? t_begin + B.unit_tiles : B.n_tiles;
    }
    KC2_STAMP(0);
    for (int tile = t_begin; tile < t_end; tile++) {
#endif
        const int c = tile * 64 + lane;
        const int apos = B.p0 + c;
        const bool inr = c < B.L;
        if (WIN && (tile * 64 + 63 < B.col_lo || tile * 64 >= B.col_hi)) {       /* outside the window: empty columns */
            if (inr) { colrec[c] = 0; nplp_out[c] = 0; score_out[c] = 0; trigcnt[c] = 0; }
            continue;
        }
        /* the tile's records: a candidate range of B.sr (short reads), or its slice of the hit list (long reads) */
        const int lo = UNIFORM(MULTI ? (int)B.hit_off[tile] : B.stile_lo[tile]), hi = UNIFORM(MULTI ? (int)B.hit_off[tile + 1] : B.stile_hi[tile]);
        const int c0 = B.p0 + tile * 64, c1 = c0 + 63;

        ColState st;
        KC2_SUMS_DECL;
        st.depth = st.nN = st.n_plp = st.nref = st.low_mq = st.indel_cnt = 0;
        st.clipped = st.n_overlap = st.n_isindel = st.has_ins = 0;
        st.cp = 0;
        int n_lq0 = 0;
        unsigned htacc = 0;
        unsigned pend = 0, key = 0;
        bool first = true;
        int wnext = lo;

        while (wnext < hi) {
            /* ---- stage: the records that reach the tile, compacted in file order into W.recs[0, nq); the two
             * order-independent per-column counts (n_plp, and the entries of low-mapq reads snp:1663) come from
             * difference arrays over the tile's columns, not from the rows */
            int nq = 0;
            W.cnt[lane] = 0; W.ht[lane] = 0; W.rowhi[lane] = 0; W.rowlo[lane] = KC2_R;
            if (lane < 4) W.cnt[64 + lane] = 0;
            WAVE_SYNC();
            if (MULTI) {
                /* long reads: most ops of the overlapping reads miss the tile; the ones that reach it were listed with
                 * the batch, in file order (k_p0_hitfill): one round trip for up to KC2_R indices, one for the records,
                 * a lane each */
                nq = hi - wnext < KC2_R ? hi - wnext : KC2_R;
                const bool ov = lane < nq;
                const int slot = lane;
                uint4 r0, r1;
                r0.x = r0.y = r0.z = r0.w = 0; r1 = r0;
                if (ov) { const uint4 *rp = (const uint4 *)&B.sr[B.hits[wnext + lane]]; r0 = rp[0]; r1 = rp[1]; }
                wnext += nq;
                bool lowrec = false;
                KC2_STAGE_HIT();
            } else {
                while (wnext < hi) {                                     /* short reads: one window, most candidates are hits */
                    const int idx = wnext + lane;
                    bool ov = false;
                    uint4 r0, r1;
                    r0.x = r0.y = r0.z = r0.w = 0; r1 = r0;
                    if (idx < hi) {
                        const uint4 *rp = (const uint4 *)&B.sr[idx];
                        r0 = rp[0]; r1 = rp[1];
                        ov = (int)r0.y > c0 && (int)r0.x <= c1;
                    }
                    unsigned long long m = __ballot(ov);
                    int h = __popcll(m), adv = 64;
                    if (KC2_R < 64 && h > KC2_R) {
                        /* more hits in the window than the matrix has rows: the first KC2_R of them now, the window
                         * resumes at the first one left out */
                        const unsigned long long over = __ballot(ov && __popcll(m & lt) == KC2_R);
                        adv = __builtin_ctzll(over);
                        ov = ov && lane < adv;
                        m &= (1ull << adv) - 1ull;
                        h = KC2_R;
                    }
                    const int slot = nq + __popcll(m & lt);
                    bool lowrec = false;
                    KC2_STAGE_HIT();
                    nq += h;
                    wnext += adv;
                    break;
                }
            }
            if (nq == 0) continue;
            WAVE_SYNC();
            KC2_STAMP(1);
            {   /* inclusive prefix sums of the difference arrays across the lanes */
                const uint32_t w2 = (uint32_t)kc2_scan_add((int)W.cnt[lane]);
                st.n_plp += (int)(w2 & 0xffffu); st.low_mq += (int)(w2 >> 16);
                htacc += W.ht[lane];
            }
            /* lane c's rows: [rowL, rowL + rowD) = from the first row that ends beyond c to the last row that starts
             * at or before c (a suffix minimum / a prefix maximum over the columns; the minimum runs over the
             * mirrored lanes).  Rows in between that do not reach c hold class 7 there.  For reads of one length the
             * window is exactly the column's depth. */
            int rowL = __shfl(kc2_scan_min(W.rowlo[63 - lane]), 63 - lane), rowD = kc2_scan_max(W.rowhi[lane]);
            rowD = rowD > rowL ? rowD - rowL : 0;
            if (rowD == 0) rowL = 0;
            const int maxd = READLANE(kc2_scan_max(rowD), 63);
            for (int g0 = 0; g0 < nq; g0 += KC2_PR) {
            const int rows = nq - g0 < KC2_PR ? nq - g0 : KC2_PR, rows4 = (rows + 3) & ~3;
            /* ---- produce: 16 records per step, four lanes (16-column segments) each; two steps' loads in flight.
             * Straight-line code over all lanes: a lane without a record or whose record does not reach its segment
             * takes the all-invalid byte mask and writes sixteen class-7 entries (that also pads the matrix to a
             * multiple of four rows); only the loads are under EXEC.  Addresses are 32-bit offsets from the pass's
             * first read (uniform base; the upload refuses batches where a tile's reads span 4 GB of bases). */
            const uint2 sof = *(const uint2 *)&W.recs[g0][0].z;
            const uint32_t so_lo = (uint32_t)UNIFORM(sof.x);
            const uint64_t so_first = ((uint64_t)(uint32_t)UNIFORM(sof.y) << 32) | so_lo;
            const uint8_t *const pbase = B.bases + so_first - BASE_FRONT_PAD, *const qbase = B.quals + so_first - BASE_FRONT_PAD;
            const unsigned phase = (unsigned)(uintptr_t)pbase & 3u;          /* bases and quals share the byte phase */
            for (int s0 = 0; s0 < rows; s0 += 32) {
                const int ne = rows - s0 > 16 ? 2 : 1;
                uint32_t bwr[2][5], qwr[2][5];
                unsigned shb[2], clD[2];
                uint4 mkA[2], mkB[2];
                bool isD[2];
                #pragma unroll
                for (int e = 0; e < 2; e++) {
                    if (e >= ne) continue;
                    const int row = s0 + 16 * e + rsub;
                    const bool rv = row < rows;
                    const uint4 r0 = W.recs[g0 + (rv ? row : 0)][0], r1 = W.recs[g0 + (rv ? row : 0)][1];
                    const int rpos = (int)r0.x, rend = (int)r0.y, lq = (int)r1.x, qoff = (int)r1.y;
                    const unsigned ty = SR_TYPE(r1.w);
                    const int sc0 = c0 + 16 * seg;                           /* first column of the segment */
                    int a = rpos - sc0, b = rend - sc0;                      /* columns [a, b) of it are reached */
                    a = a < 0 ? 0 : a; b = b > 16 ? 16 : b;
                    const bool act = rv && ty != SEG_N && lq > 0 && a < b;   /* snp:589,594 */
                    const uint32_t rel = r0.z - so_lo + (uint32_t)BASE_FRONT_PAD;
                    mkA[e] = L.mask_lt[act ? a : 16]; mkB[e] = L.mask_ge[act ? b : 16];
                    isD[e] = false;
                    #pragma unroll
                    for (int t = 0; t < 5; t++) { KC2_UNDEF(bwr[e][t]); KC2_UNDEF(qwr[e][t]); }
                    KC2_UNDEF(shb[e]); KC2_UNDEF(clD[e]);
                    if (act) {
                        if (ty == SEG_M) {
                            const uint32_t vo = rel + (uint32_t)(qoff + sc0);    /* qpos = qoff + apos */
                            shb[e] = (vo + phase) & 3u;
                            const uint32_t va = vo - shb[e];
                            const KC2U4 vb4 = *(const KC2U4 *)(pbase + va), vq4 = *(const KC2U4 *)(qbase + va);
                            bwr[e][0] = vb4.x; bwr[e][1] = vb4.y; bwr[e][2] = vb4.z; bwr[e][3] = vb4.w;
                            qwr[e][0] = vq4.x; qwr[e][1] = vq4.y; qwr[e][2] = vq4.z; qwr[e][3] = vq4.w;
                            bwr[e][4] = *(const uint32_t *)(pbase + va + 16); qwr[e][4] = *(const uint32_t *)(qbase + va + 16);
                        } else {                                             /* D: '*' with the quality of the next base; fence: read ends in D */
                            isD[e] = true;
                            clD[e] = qoff < lq ? (unsigned)qbase[rel + (uint32_t)qoff] : 0u;
                        }
                    }
                }
                #pragma unroll
                for (int e = 0; e < 2; e++) {
                    if (e >= ne) continue;
                    const int row = s0 + 16 * e + rsub;
                    uint32_t cb[4], qw[4];
                    #pragma unroll
                    for (int t = 0; t < 4; t++) {
                        cb[t] = kc2_alignbyte(bwr[e][t + 1], bwr[e][t], shb[e]) & 0x70707070u;    /* class << 4 */
                        qw[t] = kc2_alignbyte(qwr[e][t + 1], qwr[e][t], shb[e]);
                    }
                    if (isD[e]) {
                        KC2_MASKED;
                        #pragma unroll
                        for (int t = 0; t < 4; t++) { qw[t] = clD[e] * 0x01010101u; cb[t] = 0x40404040u; }
                    }
                    const uint32_t ma[4] = {mkA[e].x, mkA[e].y, mkA[e].z, mkA[e].w}, mb[4] = {mkB[e].x, mkB[e].y, mkB[e].z, mkB[e].w};
                    uint32_t out[8];
                    #pragma unroll
                    for (int t = 0; t < 4; t++) {
                        /* class byte: class << 4 | 0x08 (the mask's), 0x78 where the record does not reach the column;
                         * u16 = quality << 8 | class byte, shifted down 3: quality << 5 | class << 1 | 1 */
                        const uint32_t cm = cb[t] | ma[t] | mb[t];
                        out[2 * t]     = kc2_perm(qw[t], cm, 0x05010400u) >> 3;
                        out[2 * t + 1] = kc2_perm(qw[t], cm, 0x07030602u) >> 3;
                    }
                    uint4 *dst = (uint4 *)&W.plane[row][16 * seg];
                    uint4 o0, o1;
                    o0.x = out[0]; o0.y = out[1]; o0.z = out[2]; o0.w = out[3];
                    o1.x = out[4]; o1.y = out[5]; o1.z = out[6]; o1.w = out[7];
                    if (KC2_PR % 16 == 0 || row < KC2_PR) { dst[0] = o0; dst[1] = o1; }      /* (a step is sixteen rows) */
                }
            }
            WAVE_SYNC();
            KC2_STAMP(2);

            /* ---- consume */
#if !defined(CRUMBLE_EMU) && defined(KC2_PRIO)
            __builtin_amdgcn_s_setprio(KC2_PRIO);                         /* diagnostic: the row walk ahead of the waves that wait on memory */
#endif
            const unsigned ri = lane < rows ? W.rowinfo[lane] : 0u;
            if (first) {
                /* class 0 of each column: Boyer-Moore majority over a few rows, preferably rows that span the tile
                 * (ANY key gives the same sums; a minority key only sends the column's entries through the
                 * pending slot) */
                first = false;
                unsigned long long fm = __ballot((ri & KC2_FULL) != 0u);
                if (!fm) fm = rows >= 64 ? ~0ull : ((1ull << rows) - 1ull);
                int cand = 0, cnt = 0;
                for (int ns = 0; fm && ns < 7; ns++) {
                    const int j = __builtin_ctzll(fm);
                    fm &= fm - 1;
                    const int cl = (int)KC2_CLS(W.plane[j][lane]);
                    if (cl < 4) {
                        if (cnt == 0) { cand = cl; cnt = 1; }
                        else cnt += (cl == cand) ? 1 : -1;
                    }
                }
                key = (unsigned)cand;
            }
            const unsigned k0 = key, k1 = key ^ 1u, k2 = key ^ 2u, k3 = key ^ 3u;
            if (MQ) {
                /* rows of reads with another mapq than the batch's dominant one: the entries' table row index becomes
                 * 128 + adjusted quality (snp:632-641), one row per step with a lane per column (the lanes sit on
                 * different rows in the steps below, so a row-uniform look-up has no place there); the call_preserve
                 * input of the raw quality (the new table row carries none) is taken here */
                unsigned long long nm = __ballot((ri & KC2_NONHOT) != 0u);
                if (nm) {
                    while (nm) {
                        const int j = __builtin_ctzll(nm);
                        nm &= nm - 1;
                        const unsigned rinfo = READLANE(ri, j), slot = (rinfo >> 8) & 255u;
                        const unsigned e = W.plane[j][lane], q = (e >> 5) & 255u;
                        unsigned aq;
                        if (slot < (unsigned)B.n_slots) { aq = sAdjC[slot * 128u + (q & 127u)]; KC2_PIN(aq); }
                        else aq = T->adjc[(rinfo & 255u) * 256u + (q & 255u)];
                        if (KC2_CLS(e) != 7u) {
                            W.plane[j][lane] = (uint16_t)(((128u + aq) << 5) | (e & 0x1fu));
                            if (ANYPRES) {
                                const unsigned pv = (unsigned)sPv[q & 127u], cl = KC2_CLS(e);
                                if (pv) { st.cp |= 1u << cl; if (pv > 1u) st.cp |= (1u << cl) << 8; }
                            }
                        }
                    }
                    WAVE_SYNC();
                }
            }
            {   /* rows with a D/N op, an indel behind the op or no SEQ: the order-independent indel bookkeeping of
                 * the whole row at once (their consensus entries are ordinary matrix entries) */
                unsigned long long rm = __ballot((ri & KC2_RARE) != 0u);
                while (rm) {
                    const int j = __builtin_ctzll(rm);
                    rm &= rm - 1;
                    const uint4 r0 = W.recs[j][0], r1 = W.recs[j][1];
                    const int rpos = (int)r0.x, rend = (int)r0.y, lq = UNIFORM((int)r1.x), qoff = (int)r1.y;
                    const int indel_after = (int)r1.z;
                    const unsigned info = UNIFORM(r1.w), ty = SR_TYPE(info);
                    const bool isM = ty == SEG_M, isN = ty == SEG_N;
                    const bool cov = inr & ((unsigned)(apos - rpos) < (unsigned)(rend - rpos));
                    const int e_indel = (apos == rend - 1) ? indel_after : 0;
                    const bool isind = (e_indel != 0) || !isM;
                    const bool head = (info & SR_FIRST) && apos == rpos, tail = (info & SR_LAST) && apos == rend - 1;
                    st.indel_cnt += (cov && isind) ? 1 : 0;                    /* snp:1664 */
                    st.nref += (cov && isN) ? 1 : 0;                           /* snp:1468 */
                    st.n_isindel += (cov && !isN && isind) ? 1 : 0;
                    st.has_ins |= (cov && !isN && !head && !tail && e_indel > 0) ? 1 : 0;   /* snp:1708 */
                    if (lq == 0) n_lq0 += cov ? 1 : 0;
                    if (!(isN || lq == 0) && ANYPRES && cov && e_indel > 0) {  /* qualities of the inserted bases, snp:616-622 */
                        const int64_t so = (int64_t)(((uint64_t)r0.w << 32) | r0.z);
                        const int e_qpos = isM ? qoff + apos : qoff;
                        for (int ins = 0; ins < e_indel; ins++) {
                            const int qi = e_qpos + 1 + ins;
                            if (qi < lq && T->pvc[B.quals[so + qi]]) st.cp |= 1u << 4;
                        }
                    }
                }
            }
            /* every lane walks its own row window, all lanes in step: step t is row rowL + t of the lane's column */
            const int steps4 = (maxd + 3) & ~3;
            /* all lanes take steps4 rows; a window that would run past the padded matrix starts earlier instead
             * (rows before it ended before the column, rows behind it start after it: class 7 either way) */
            const uint16_t *basep = &W.plane[0][0] + (rowL < rows4 - steps4 ? rowL : rows4 - steps4) * 64 + lane;
#ifdef KC2_EXP_STAMP
            npass++; nstep += steps4;
#endif
            unsigned vn[4];
            #pragma unroll
            for (int u = 0; u < 4; u++) vn[u] = basep[u * 64];
#ifdef KC2_EXP_NOCONSUME
            for (int j0 = 0; j0 < 4 && j0 < steps4; j0 += 4) {
#else
            for (int j0 = 0; j0 < steps4; j0 += 4) {
#endif
                unsigned ve[4];
                double rMM[4], r_M[4], rom[4];
                unsigned rpv[4];
                #pragma unroll
                for (int u = 0; u < 4; u++) {
                    ve[u] = vn[u];
#ifdef KC2_EXP_NOCONFLICT    /* diagnostic build (wrong results): every lane reads ONE table row -- no LDS bank conflicts */
                    const double *rw = (const double *)(rowbase + ((ve[u] & 0u) | (37u << 5)));
#else
                    const double *rw = (const double *)(rowbase + (ve[u] & KC2_ROWMASK));
#endif
                    rMM[u] = rw[0]; r_M[u] = rw[1]; rom[u] = rw[2];
                    if (ANYPRES) memcpy(&rpv[u], &rw[3], 4); else rpv[u] = 0;
                }
                if (j0 + 4 < steps4) {
                    #pragma unroll
                    for (int u = 0; u < 4; u++) vn[u] = basep[(j0 + 4 + u) * 64];
                }
                #pragma unroll
                for (int u = 0; u < 4; u++) {
                    KC2_ROW(ve[u], rMM[u], r_M[u], rom[u]);
                    /* call_preserve bits, snp:611-623: the table row's bits at the entry's class.  Class 7 -- no entry,
                     * its row index may point anywhere, past the table too: hence the mask -- lands on bits 7 and 15,
                     * dropped after the last pass */
                    if (ANYPRES) st.cp |= (rpv[u] & 0x101u) << KC2_CLS(ve[u]);
                }
            }
#if !defined(CRUMBLE_EMU) && defined(KC2_PRIO)
            __builtin_amdgcn_s_setprio(0);
#endif
            WAVE_SYNC();                                                 /* the next group / pass overwrites plane / recs */
            KC2_STAMP(3);
            }
        }
        KC2_DRAIN();
        st.nN = nN;
        if (ANYPRES) st.cp &= 0x7f7fu;
        if (inr) {
            const uint32_t h = htacc;
            st.depth = st.n_plp - st.nref - n_lq0;       /* entries that reached the sums (snp:685) */
            st.clipped = (int)(h >> 16);
            st.n_overlap = st.n_plp - st.nref - (int)(h & 0xffff);
        }
#ifdef KC2_EXP_NODECIDE
        if (inr) { colrec[c] = (s00 + s13 + d2 > 1e300) ? 1u : 0u; nplp_out[c] = st.n_plp; score_out[c] = 0; trigcnt[c] = 0; }
        continue;
#endif
        const bool in_win = !WIN || (c >= B.col_lo && c < B.col_hi);
        const int bed_lo = B.n_bed ? UNIFORM(B.tile_bed[tile]) : 0; /* keep-regions: the tile's cursor (k_p0_tilebed), not a search per column */
#if KC2_FASTFIN
        {   /* the common tile: in every column that reaches the consensus the ten genotypes without the key base are out
             * of reach (cons_fast_ok, kernels.h) -- the result comes from the five key genotypes in place, no un-permute
             * of 20 sums, 5 exp look-ups instead of 15 */
            const bool needs = inr && in_win && st.n_plp > 0 && st.nref != st.n_plp && st.n_plp <= MAX_DEPTH_COL;
            const double dmax = dev_max(dev_max(dev_max(dev_max(s11, s12), dev_max(s13, s14)), dev_max(dev_max(s22, s23), dev_max(s24, s33))), dev_max(s34, s44));
            const bool ok = !needs || cons_fast_ok(T, s00, s01, s02, s03, s04, dmax, st.depth, st.nN);
            if (__ballot(!ok) == 0ull) {
                KC2_EMU_COUNT(0);
                ConsOut co;
                cons_finalize_fast(T, s00, s01, s02, s03, s04, d0, d1, d2, d3, d4, key, co, sE, MQ ? P->min_discrep_B : P->min_discrep_A, needs);
                if (inr) {
                    if (in_win)
                        col_decide<MQ ? 1 : 0, true>(st, B, P, T, c, apos, true, colrec, nplp_out, score_out, trigcnt,
                                                     sStat, sE, WIN ? c < B.stat_hi : true, &co, bed_lo);
                    else { colrec[c] = 0; nplp_out[c] = 0; score_out[c] = 0; trigcnt[c] = 0; }
                }
                KC2_STAMP(4);
                continue;
            }
        }
#endif
        KC2_EMU_COUNT(1);
        st.S[0] = s00; st.S[1] = s01; st.S[2] = s02; st.S[3] = s03; st.S[4] = s04;
        st.S[5] = s11; st.S[6] = s12; st.S[7] = s13; st.S[8] = s14; st.S[9] = s22;
        st.S[10] = s23; st.S[11] = s24; st.S[12] = s33; st.S[13] = s34; st.S[14] = s44;
        st.sC[0] = d0; st.sC[1] = d1; st.sC[2] = d2; st.sC[3] = d3; st.sC[4] = d4;
        if (key) col_unpermute(st.S, st.sC, (int)key);  /* back to the reference's A C G T numbering */
        if (inr) {
            if (in_win)
                col_decide<MQ ? 1 : 0>(st, B, P, T, c, apos, true, colrec, nplp_out, score_out, trigcnt,
                                       sStat, KC2_LDS_EXP ? sE : (const double *)nullptr, WIN ? c < B.stat_hi : true, nullptr, bed_lo);
            else { colrec[c] = 0; nplp_out[c] = 0; score_out[c] = 0; trigcnt[c] = 0; }
        }
        KC2_STAMP(4);
    }
    }
#ifdef KC2_EXP_STAMP
    if (lane == 0) for (int k = 0; k < 6; k++) atomicAdd(&stats[24 + k], (u64)tacc[k]);
    if (lane == 0) { atomicAdd(&stats[30], (u64)npass); atomicAdd(&stats[31], (u64)nstep); }
#endif
    __syncthreads();
    if (threadIdx.x < 32 && sStat[threadIdx.x]) atomicAdd(&stats[threadIdx.x], (u64)sStat[threadIdx.x]);
}


#endif /* CRUMBLE_KCOL2_H */
