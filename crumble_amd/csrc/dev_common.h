/*
 * dev_common.h -- structures shared by the HIP kernels and the host layer.
 */
#ifndef CRUMBLE_DEV_COMMON_H
#define CRUMBLE_DEV_COMMON_H

#include <stdint.h>
#include <limits.h>
#include <string.h>

#ifndef CRUMBLE_EMU
#include <hip/hip_runtime.h>
#define CRUMBLE_LAUNCH(kern, grid, block, stream, ...) \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, stream, __VA_ARGS__)
/* kernels whose lanes never communicate (the CPU test emulator may run their
 * lanes sequentially); on the GPU both forms are the same launch */
#define CRUMBLE_LAUNCH_INDEP CRUMBLE_LAUNCH
#endif

#define HD __host__ __device__ __forceinline__

/* Device copy of crumble_params plus derived values. */
struct DParams {
    int32_t reduce_qual, binary_qual, iSTR_add, sSTR_add;
    double  iSTR_mul, sSTR_mul;
    int32_t qlow, qcutoff, qhigh, qcap, min_mqual;
    int32_t min_qual_A, min_indel_A, min_qual_B, min_indel_B;
    int32_t pblock, softclip, perfect_col;
    double  indel_fract, min_discrep_A, min_discrep_B, low_mqual_perc;
    double  clip_perc, ins_len_perc, over_depth, indel_ov_perc;
    int32_t str_snp;        /* sSTR_add || sSTR_mul          snp:1345 */
    int32_t any_preserve;   /* any preserve_qual[] non-zero            */
    /* the preserved qualities as a short list, for the packed replay (k_replay_fast): n_pv = -1 when there are more
     * than four of them or one is >= 128 (then only the byte-wise kernels apply) */
    int32_t n_pv;
    uint8_t pv_val[4], pv_lvl[4];
    uint8_t preserve_qual[256];
    uint8_t bin2[256];      /* init_bins                     snp:234  */
};

/* Host-built (libm) tables: consensus_init snp:378-489 + statics snp:560-575. */
struct DTables {
    double MMd[101];        /* pMM[q] - p__[q]   snp:645 */
    double _Md[101];        /* p_M[q] - p__[q]   snp:646 */
    double omq[101];        /* 1 - q2p[q]        snp:651 */
    double lprior15[15];
    double e_tab[1001];     /* exp(i),    i = -500..500  (index +500) */
    double e_tab2[1001];    /* exp(i/10), i = -500..500  (index +500) */
    double min_e_exp;       /* DBL_MIN_EXP*log(2)+1  snp:540 */
    uint8_t adj[65536];     /* [mapq][qual] -> adjusted quality, snp:632-638 */
    /* the same lookups with this engine's parameters folded in (k_columns2 stages table ROW INDICES):
     *   capc[q]     = clamp(cap_quality(q), 1, 100)            the row of a quality without CONS_MQUAL (snp:1325, 641)
     *   adjc[m][q]  = clamp(adj[m][cap_quality(q)], 1, 100)    the row with CONS_MQUAL (snp:632-641)
     *   pvc[q]      = preserve_qual[cap_quality(q)]            call_preserve input (snp:611-623) */
    uint8_t capc[256];
    uint8_t pvc[256];
    uint8_t adjc[65536];
};
#define CIX_UNIT 16
#define BASE_FRONT_PAD 64   /* bytes in front of bases[0] / quals[0]: k_columns2 loads whole 16-column segments */

/* One reference-consuming CIGAR op (M,=,X / D / N) of a read, with what
 * htslib's resolve_cigar2 would report on its last column. */
#define SEG_M 0u
#define SEG_D 1u
#define SEG_N 2u
struct Seg {
    int32_t  roff;          /* reference offset from the read's pos          */
    int32_t  qstart;        /* query index of its first base (M) or of the   */
                            /* next base (D/N)                               */
    uint32_t len_type;      /* len | type << 30                              */
    int32_t  indel_after;   /* bam_pileup1_t.indel on the op's last column   */
};
HD int32_t  seg_len(const Seg &s)  { return (int32_t)(s.len_type & 0x3fffffffu); }
HD uint32_t seg_type(const Seg &s) { return s.len_type >> 30; }

/* Compact per-read record for the column kernel (one 32-byte load per read). */
#define RR_SIMPLE 0x100u    /* one M/=/X op covers the whole alignment, no indel flag, l_qseq > 0 */
struct ReadRec {
    int32_t  pos, end;      /* end = pos + reference length; == pos if not in the pileup */
    int64_t  so;            /* seq_off */
    int32_t  lq;            /* l_qseq */
    int32_t  q0;            /* query index of the first aligned base */
    uint32_t info;          /* mapq | RR_SIMPLE */
    uint32_t cix;           /* general CIGARs: first entry of this read in the base->column table, in units of
                             * CIX_UNIT entries (every read's slice is padded to a multiple: 2^32 * 16 entries reachable) */
};

/* Long-read batches, replay: the base array cut into territories, one per reference-consuming op of a read (bases
 * from the op's first query base up to the next op's) plus one in front of every read (leading clip; the whole read
 * when it is outside the pileup).  A match op's territory starts with its aligned bases, so k_replay_fast treats a
 * territory exactly like a single-match read: slot = seg_off[r] + r + (1 + op index). */
struct VRec {
    int64_t  so;            /* first byte of the territory in the flat base array                */
    int32_t  pos, end;      /* reference span of a match op; end == pos: no aligned bases here   */
    int32_t  lq;            /* bytes in the territory                                            */
    int32_t  q0;            /* (0: aligned bases start the territory)                            */
    int32_t  hcol;          /* head column of a low-mapq read (snp:1852), -1: no mark            */
    int32_t  read;          /* the read, for the per-read flags                                  */
};

/* One reference-consuming CIGAR op of one read, self-contained, in file order: the unit
 * the column kernel iterates over (a read with k such ops contributes k consecutive
 * records, so every record is "one op covering [pos,end)" and the kernel has one path). */
#define SR_TYPE(i)     (((i) >> 8) & 3u)     /* SEG_M / SEG_D / SEG_N            */
#define SR_FIRST       (1u << 10)            /* contains the read's first column */
#define SR_LAST        (1u << 11)            /* contains the read's last column  */
#define SR_HEADCLIP    (1u << 12)            /* qpos > 0 on the head column      (snp:1701) */
#define SR_TAILCLIP    (1u << 13)            /* qpos+1 < l_qseq on the tail column (snp:1702) */
struct SegRec {
    int32_t  pos, end;      /* reference span of this op                                   */
    int64_t  so;            /* seq_off of the read                                         */
    int32_t  lq;            /* l_qseq                                                      */
    int32_t  qoff;          /* M: qpos = qoff + apos;  D/N: qpos = qoff (the next base)    */
    int32_t  indel_after;   /* bam_pileup1_t.indel on the op's last column                 */
    uint32_t info;          /* mapq | type << 8 | SR_* flags | mapq slot << 16 (255: none) */
};

/* Device-resident batch. Positions are stored relative to p0. */
struct DBatch {
    int64_t n_reads, n_bases;
    int32_t p0;             /* first pileup position                         */
    int32_t L;              /* number of positions covered: [p0, p0+L)       */
    int32_t n_tiles;        /* ceil(L/64)                                    */
    int32_t n_bed;
    int32_t hot_mq;         /* most frequent mapq of the batch (sampled): a performance hint only */
    int32_t pad0_;
    /* window of a contig (crumble_hip_set_window; the whole batch by default): columns [col_lo, col_hi) are decided
     * (the others read as empty), columns < stat_hi enter the statistics */
    int32_t col_lo, col_hi, stat_hi, pad1_;
    const int32_t *pos, *end;       /* end = pos + reflen; == pos if passthru */
    const uint8_t *mapq, *inplp;
    const int64_t *seq_off, *cig_off, *seg_off;
    const uint32_t *cigar;
    const Seg *segs;
    const ReadRec *rr;
    const int32_t *cix;     /* per base of every non-simple read: column (pos-p0), -1 none, -2 general routine */
    const VRec *vr;         /* long-read batches: territories (see VRec); n_vrec of them, chunk table in chunk_read */
    int64_t n_vrec;
    const SegRec *sr;
    const int32_t *stile_lo, *stile_hi;   /* candidate SegRec range per 64-column tile */
    const int64_t *hit_off;               /* long-read batches: [n_tiles+1] the tile's slice of hits[] (k_p0_hitfill) */
    const int32_t *hits;                  /* SegRec indices of the records that reach a tile, file order */
    int32_t unit_tiles;                   /* tiles per unit of the column kernels' work counter (set per launch) */
    int32_t n_units;                      /* ceil(n_tiles / unit_tiles) */
    int32_t n_slots;                      /* mapq values (other than hot_mq) with a slot in k_columns2's compact adjustment table */
    const uint8_t *slot_mq;               /* [n_slots] slot -> mapq, most frequent first; SegRec.info carries the slot */
    const uint8_t *bases, *quals;
    const int32_t *tile_lo, *tile_hi;
    const int32_t *bed_start, *bed_end, *bed_pmax;
    const int32_t *tile_bed;              /* [n_tiles] first keep-region whose prefix-max end reaches the tile (n_bed > 0) */
};

/* per-position column record */
#define CR_A1(r)        ((r) & 7u)
#define CR_A2(r)        (((r) >> 3) & 7u)
#define CR_PRESERVE(r)  (((r) >> 6) & 3u)
#define CR_PERFECT      (1u << 8)
#define CR_KEEP         (1u << 9)
#define CR_TRIG         (1u << 10)
#define CR_WIN          (1u << 11)
#define CR_COUNTED      (1u << 12)   /* column went through the full engine  */
#define CR_ALLREFSKIP   (1u << 13)
#define CR_TOODEEP      (1u << 14)
#define CR_HASINS       (1u << 15)   /* indel_sz > 0: needs the histogram    */
#define CR_EXISTS       (1u << 16)   /* n_plp > 0                            */
#define CR_SCAN         (1u << 17)   /* indel_cnt >= n_plp*indel_fract       */
/* why keep_qual was forced on (the reference's -b reasons, snp:1676,1768,1802,1810) */
#define CR_R_DEEP       (1u << 18)
#define CR_R_CLIP       (1u << 19)
#define CR_R_INSLEN     (1u << 20)
#define CR_R_INDELOV    (1u << 21)
#define CR_SKIP         (CR_ALLREFSKIP | CR_TOODEEP)

/* one STR/indel trigger (snp:1718-1756) */
struct TrigItem {
    int32_t read;
    int32_t qpos1_kind;     /* (qpos+1) | is_indel << 30 | noscan << 29      */
    int32_t lo, hi;         /* mask_LC_regions result (INT_MAX,0 = none)     */
};

struct ScanVal {            /* per-position quantities that need prefix sums */
    int64_t depth;          /* sum of n_plp over counted columns             */
    int32_t cols;           /* counted columns                               */
    int32_t trigcols;       /* columns with >= 1 trigger                     */
    int32_t items;          /* triggers                                      */
    int32_t ins;            /* columns needing the indel-size histogram      */
    int32_t toodeep;        /* columns deeper than MAX_DEPTH                 */
    int32_t pad;
};

struct Epoch {              /* running-average state of snp:1490-1491,1684   */
    int64_t ord_start;      /* ordinal of the first counted column           */
    int64_t tc_start;       /* total_col before that column                  */
    int64_t td_start;       /* total_depth before that column                */
    int64_t dpre_start;     /* exclusive depth prefix at that column         */
};
#define MAX_EPOCHS 8192

#define MAX_DEPTH_COL 20000     /* MAX_DEPTH snp:92   */
/* tiles handed out per grab of the column kernels' work counter.  k_columns2's workgroups deal a unit's tiles to their
 * waves one by one, so the unit only sets how often the global counter is touched and how close together the tiles in
 * flight on a CU lie; k_columns gives a wave the whole unit.  (Rounds 1-3 cut units of equal estimated cost from a prefix
 * sum of per-tile costs: three more kernels and a read-back per upload for a balance the per-tile dealing gives anyway.) */
#ifndef KCOL2_UTILES
#define KCOL2_UTILES  8
#endif
#ifndef KCOL1_UTILES
#define KCOL1_UTILES  2
#endif
#define MASK_WIN_DEV  250       /* MASK_WIN  snp:1229 */
#define STR_CAP       640       /* live repeats per window   */

#endif
