/*
 * crumble_hip.h -- C ABI of the MI355X-native quality-quantisation engine.
 *
 * This is the drop-in boundary for crumble's column engine: everything that
 * reference crumble does inside the body of transcode()'s pileup loop
 * (/root/reference/snp_score.c:1437-1983) plus the per-read epilogue of
 * flush_bam_list() (snp_score.c:1090-1100: strip 0x80, P-block) is done by
 * crumble_hip_contig() on one GPU for one reference contig at a time.
 * File I/O (htslib sam_read1/sam_write1), aux-tag purging and the CLI stay on
 * the host; see INTEGRATION.md for the binding a crumble maintainer would add.
 *
 * Plain C, plain pointers and sizes; no HIP or torch types cross this ABI.
 * The caller owns every host buffer; the library owns every device buffer.
 * All entry points return 0 on success or a negative CRUMBLE_E* code and never
 * abort (the reference asserts / aborts instead: snp_score.c:1075,1830,1842).
 */
#ifndef CRUMBLE_HIP_H
#define CRUMBLE_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------- ABI version */
/* The structs below are passed by pointer and read in full by the library: a host built against an older header would
 * hand over a shorter crumble_batch / crumble_params.  Check once at start-up:
 *     crumble_hip_abi_version() == CRUMBLE_HIP_ABI  &&  crumble_hip_struct_size(0) == sizeof(crumble_batch)  && ...
 * and zero every struct (memset) before filling it in: fields added by later versions then read as "not given".
 * CRUMBLE_HIP_ABI changes whenever a struct of this header gains, loses or re-orders a field. */
#define CRUMBLE_HIP_ABI 6
int    crumble_hip_abi_version(void);
/* sizeof, as the LIBRARY was built: which = 0 crumble_batch, 1 crumble_params, 2 crumble_coded, 3 crumble_window,
 * 4 crumble_carry, 5 crumble_column; 0 for any other value */
size_t crumble_hip_struct_size(int which);

/* ------------------------------------------------------------------ errors */
#define CRUMBLE_OK            0
#define CRUMBLE_EINVAL       -1   /* bad argument / unsupported parameter     */
#define CRUMBLE_ENOMEM       -2   /* host or device allocation failed         */
#define CRUMBLE_EHIP         -3   /* HIP runtime error (see strerror)         */
#define CRUMBLE_EUNSORTED    -4   /* reads not coordinate sorted (snp:1075)   */
#define CRUMBLE_EOVERFLOW    -5   /* internal work-list capacity exceeded     */
#define CRUMBLE_ENODEVICE    -6   /* no usable GPU: the product never falls   */
                                  /* back to a CPU path                       */
#define CRUMBLE_ERCCL        -7   /* RCCL failure in the stats all-reduce     */

/* --------------------------------------------------------------- parameters */
/*
 * POD restatement of the hot-path fields of cram_lossy_params
 * (snp_score.c:185-226) plus the file-scope preserve_qual[] table
 * (snp_score.c:232, filled by -k/-K snp_score.c:2362-2375 and -y pbccs
 * snp_score.c:2326).  Fields that only drive host plumbing (region, bed_fn,
 * aux white/black lists, BD/BI binning, verbose, noPG, bed_fp) are not here.
 */
typedef struct crumble_params {
    int32_t reduce_qual;      /* -L  snp:103  */
    int32_t binary_qual;      /* -B  snp:104  */
    int32_t iSTR_add;         /* -i  snp:129  */
    int32_t sSTR_add;         /* -s  snp:126  */
    double  iSTR_mul;         /* -i  snp:128  */
    double  sSTR_mul;         /* -s  snp:125  */
    int32_t qlow;             /* -l  snp:94   */
    int32_t qcutoff;          /* -c  snp:95   */
    int32_t qhigh;            /* -u  snp:96   */
    int32_t qcap;             /* -U  snp:97   */
    int32_t min_mqual;        /* -m  snp:100  */
    int32_t min_qual_A;       /* -q  snp:110  */
    int32_t min_indel_A;      /* -d  snp:111  */
    int32_t min_qual_B;       /* -Q  snp:115  */
    int32_t min_indel_B;      /* -D  snp:116  */
    int32_t pblock;           /* -p  snp:147  */
    int32_t softclip;         /* -S           */
    int32_t perfect_col;      /* -N           */
    double  indel_fract;      /* -Y  snp:107  */
    double  min_discrep_A;    /* -x  snp:112  */
    double  min_discrep_B;    /* -X  snp:117  */
    double  low_mqual_perc;   /* -M  snp:133  */
    double  clip_perc;        /* -C  snp:139  */
    double  ins_len_perc;     /* -Z  snp:136  */
    double  over_depth;       /* -P  snp:142  */
    double  indel_ov_perc;    /* -V  snp:145  */
    uint8_t preserve_qual[256]; /* 0 none, 1 = -k, 2 = -K */
} crumble_params;

/* Defaults of main() (snp_score.c:2152-2192) == level 9. */
void crumble_params_default(crumble_params *p);
/* Presets -1 -3 -5 -7 -8 -9 (snp_score.c:2380-2482); others -> CRUMBLE_EINVAL. */
int  crumble_params_preset(crumble_params *p, int level);
/* -y illumina | pbccs (snp_score.c:2316-2328). */
int  crumble_params_tech(crumble_params *p, const char *tech);
/* -k / -K quality list "93" or "2-5,40" (snp_score.c:2362-2375); really=1 for -K. */
int  crumble_params_keep_quals(crumble_params *p, const char *list, int really);
/* init_bins() (snp_score.c:234-247). */
void crumble_init_bins(const crumble_params *p, uint8_t bin2[256]);

/* ------------------------------------------------------------------ batches */
/*
 * One reference contig worth of reads in SoA form, coordinate sorted exactly
 * as the BAM was (file order == (pos, arrival) order, snp_score.c:863-872).
 * This replaces the bam1_t stream that pileup_callback() (snp_score.c:1113)
 * pulls from htslib.  Reads that htslib's pileup never shows crumble
 * (BAM_FUNMAP, or no reference-consuming CIGAR op: snp_score.c:1130-1149) may
 * be present: they are passed through like the reference's b_hist entries
 * (only the P-block epilogue touches them).  Reads carrying
 * SECONDARY/QCFAIL/DUP are handled the same way (extension; the reference
 * asserts on them).  tid == -1 marks the trailing
 * unmapped block (snp_score.c:1998-2012): every read is passed through.
 */
typedef struct crumble_batch {
    int32_t  tid;
    int64_t  n_reads;
    const int32_t  *pos;      /* [n_reads] 0-based leftmost coordinate          */
    const uint8_t  *mapq;     /* [n_reads]                                      */
    const uint16_t *flag;     /* [n_reads] BAM flag                             */
    const int64_t  *seq_off;  /* [n_reads+1] first base of read i in bases/quals*/
    const int64_t  *cig_off;  /* [n_reads+1] first op of read i in cigar        */
    const uint32_t *cigar;    /* BAM encoding: len<<4 | op (MIDNSHP=X)          */
    const uint8_t  *bases;    /* one nt16 code (0..15, "=ACMGRSVTWYHKDBN") per  */
                              /* base; l_qseq(i) = seq_off[i+1]-seq_off[i]      */
    const uint8_t  *quals;    /* phred, one byte per base; must be < 128: a    */
                              /* larger value is found by the kernels that     */
                              /* read the qualities, i.e. reported (EINVAL) by */
                              /* crumble_hip_run / _run_finish / _contig, not  */
                              /* by crumble_hip_upload; such a batch adds      */
                              /* nothing to the accumulated statistics         */
    int32_t  n_bed;           /* keep-regions of THIS contig, after             */
    const int32_t *bed_start; /* crumble_bed_collapse() (bed.c:20-40)           */
    const int32_t *bed_end;
    /* Optional, INSTEAD of bases (then bases == NULL): the bases as BAM stores them (bam_get_seq: two nt16 codes per
     * byte, the earlier base in the high nibble, every read starting on a byte).  Read i occupies the
     * (l_qseq(i) + 1) / 2 bytes from CRUMBLE_BASES4_OFF(seq_off[i], i); the array is CRUMBLE_BASES4_SIZE(n_bases,
     * n_reads) bytes.  Half the host -> device bytes of the one-code-per-byte form and no expansion pass on the host:
     * the device unpacks while it rewrites the base bytes anyway (k_p0_unpack4). */
    const uint8_t  *bases4;
    /* Optional, the compact form of the two offset tables: BAM's own per-read fields, l_qseq[i] = l_seq of read i
     * (int32) and n_cigar[i] = its n_cigar_op (uint16), with the totals n_bases = seq_off[n_reads] and n_cigar_ops =
     * cig_off[n_reads].  When BOTH arrays are given the engine sends these 6 bytes per read to the device instead of the
     * 16 of seq_off / cig_off (which may then be NULL) and derives the offsets there (k_p0_len_*).  A batch that holds a
     * read of more than 65535 CIGAR ops (SAMv1 4.2.2, CG tag) uses cig_off. */
    const int32_t  *l_qseq;
    const uint16_t *n_cigar;
    int64_t  n_bases, n_cigar_ops;
    /* Optional, INSTEAD of quals (then quals may be NULL): the qualities as codes of qual_bits (2 or 4) bits each into
     * qual_lut (quality of base i = qual_lut[code i]; code i sits in byte i * qual_bits / 8 of qualsp at bit
     * (i * qual_bits) % 8, counted over ALL bases of the batch, not per read; CRUMBLE_QUALSP_SIZE bytes).  What binned
     * instruments write (four or eight distinct values) then crosses PCIe as 0.25 / 0.5 bytes per base instead of 1;
     * the device makes the one-byte form while it expands the batch (k_p0_quals).  crumble_pack_quals() makes the codes
     * and says when a batch has too many distinct values for them.  qual_bits == 0: quals is used. */
    const uint8_t  *qualsp;
    int32_t  qual_bits;
    uint8_t  qual_lut[16];
} crumble_batch;
#define CRUMBLE_QUALSP_SIZE(n_bases, bits)    (((int64_t)(n_bases) * (bits) + 7) / 8 + 16)
/* quals (one byte per base) -> the coded form above.  Returns 2 or 4 (the code width; lut filled, out written:
 * CRUMBLE_QUALSP_SIZE(n_bases, 4) bytes are enough for either) or 0 when the batch holds more than sixteen distinct
 * values (out unspecified: hand over quals).  Host helper (a team of host threads). */
int crumble_pack_quals(int64_t n_bases, const uint8_t *quals, uint8_t *out, uint8_t lut[16]);
/* (out == NULL: the look alone -- which values occur -- so that a caller can decide before it allocates: returns the width
 * the batch would take, or 0) */
/* Host threads the library's host helpers (crumble_pack_quals, crumble_pack_bases4, crumble_decode_quals with threads == 0) may
 * use from now on; n <= 0: all the process may use (affinity mask, cgroup quota; at most 32).  A driver with one worker
 * per device hands each its share of the user's -@. */
void crumble_set_host_threads(int n);
#define CRUMBLE_BASES4_OFF(seq_off_i, i)      (((int64_t)(seq_off_i) >> 1) + (int64_t)(i))
#define CRUMBLE_BASES4_SIZE(n_bases, n_reads) (((int64_t)(n_bases) >> 1) + (int64_t)(n_reads) + 1)
/* bases (one code per byte) -> the bases4 layout above; out has CRUMBLE_BASES4_SIZE bytes.  Host helper (a team of host threads). */
void crumble_pack_bases4(int64_t n_reads, const int64_t *seq_off, const uint8_t *bases, uint8_t *out);

/* per-read output flags */
#define CRUMBLE_RF_KEEP     1   /* bam_sorted_item.keep_qual was set (snp:1847) */
#define CRUMBLE_RF_NOPBLOCK 2   /* flag bit 15 was set (snp:1890-1892)          */
#define CRUMBLE_RF_PASSTHRU 4   /* read never entered the pileup                */

/* ------------------------------------------------------------------- stats  */
/* int64 vector: the reference's counters (snp_score.c:1292-1311), read/base
 * totals and a 256-bin histogram of the output qualities (new: this is what
 * the multi-GPU all-reduce sums). */
enum {
    CRUMBLE_ST_COLUMNS = 0,
    CRUMBLE_ST_HET_A, CRUMBLE_ST_HET_QUAL_A, CRUMBLE_ST_HOM_A, CRUMBLE_ST_HOM_QUAL_A,
    CRUMBLE_ST_DISCREP_A,
    CRUMBLE_ST_HET_B, CRUMBLE_ST_HET_QUAL_B, CRUMBLE_ST_HOM_B, CRUMBLE_ST_HOM_QUAL_B,
    CRUMBLE_ST_DISCREP_B,
    CRUMBLE_ST_DIFF, CRUMBLE_ST_INDEL, CRUMBLE_ST_INDEL_QUAL,
    CRUMBLE_ST_LOW_MQUAL_PERC, CRUMBLE_ST_CLIP_PERC, CRUMBLE_ST_INS_LEN_PERC,
    CRUMBLE_ST_INDEL_OV_PERC, CRUMBLE_ST_OVER_DEPTH,
    CRUMBLE_ST_READS, CRUMBLE_ST_BASES, CRUMBLE_ST_BASES_CHANGED,
    CRUMBLE_ST_TRIGGERS, CRUMBLE_ST_TOO_DEEP,
    CRUMBLE_ST_HIST0 = 32,          /* [32..287] output quality histogram */
    CRUMBLE_ST_N = 288
};

/* ----------------------------------------------------------- BED (bed.c)   */
/* Sort + merge exactly like bed_collapse() (bed.c:20-40) minus its phantom
 * trailing entry.  In place; returns the new count. */
int crumble_bed_collapse(int32_t *tid, int32_t *start, int32_t *end, int n);

/* ------------------------------------------------------------ the engine   */
typedef struct crumble_hip crumble_hip;

/* Creates an engine bound to HIP device `device`. Builds the log-probability
 * tables of consensus_init() (snp_score.c:378-489) on the host with libm and
 * uploads them. *err receives the error code when NULL is returned. */
crumble_hip *crumble_hip_create(const crumble_params *p, int device, int *err);
void         crumble_hip_destroy(crumble_hip *h);

/* Implementation choices that never change a result; the parity tests force every variant through this.
 *   "kcol"   = auto | v1 | v1perm | v2      column kernel (k_columns, its per-column-class form, k_columns2)
 *   "pblock" = auto | reads | segments | words   P-block decomposition
 *   "replay" = auto | flat                  per-base replay kernel
 *   "run"    = fast | careful               crumble_hip_run's first attempt: work lists sized by guess and measured on
 *                                           the device (no read-back in the middle; repeated the careful way when a
 *                                           guess was too small), or sized from a read-back as the windows' phases are
 *   "caps"   = <n>                          the fast attempt's list capacities (0: guessed; tests force the repeat)
 *   "stages" = off | on                     HIP events around the stages of a run (crumble_hip_kernel_times)
 *   "dl_chunk" = <bytes>                    size of one staged device -> host copy (256 .. 8 MiB; crumble_hip_download
 *                                           of >= 4 chunks goes through page-locked staging buffers on 8 streams)
 * kcol / pblock / replay take effect at the next crumble_hip_upload(), the others at once.  Unknown names / values ->
 * CRUMBLE_EINVAL. */
int crumble_hip_set_option(crumble_hip *h, const char *name, const char *value);

/* Replaces one contig's worth of transcode() loop iterations + flush:
 *   qual_out[seq_off[n_reads]]  final quality bytes (0x80 stripped, P-blocked)
 *   read_flags_out[n_reads]     CRUMBLE_RF_* (may be NULL)
 *   stats[CRUMBLE_ST_N]         ADDED to (caller zeroes) (may be NULL)
 * = crumble_hip_upload + crumble_hip_run + crumble_hip_download. */
int crumble_hip_contig(crumble_hip *h, const crumble_batch *b,
                       uint8_t *qual_out, uint8_t *read_flags_out, int64_t *stats);

/* crumble_hip_contig() without its download: host -> device, expansion, run (the expansion's tail overlaps the run's
 * first kernels, one wait less than crumble_hip_upload + crumble_hip_run).  Follow with crumble_hip_download(), or
 * crumble_hip_encode() + crumble_hip_download_coded(). */
int crumble_hip_contig_run(crumble_hip *h, const crumble_batch *b);

/* The three stages separately, so callers (and bench.py) can keep a batch
 * resident in HBM and time only the kernels. */
int crumble_hip_upload(crumble_hip *h, const crumble_batch *b);
/* The copy half of crumble_hip_upload() alone: returns when the batch's raw arrays are on the device.  Follow with
 * crumble_hip_expand() (+ crumble_hip_run()): a caller that streams contigs through several handles keeps the link busy
 * with the next contig's copy while this one's expansion waits for its turn on the GPU (bench.py's streamed leg). */
int crumble_hip_upload_raw(crumble_hip *h, const crumble_batch *b);
/* The device half of crumble_hip_upload() once more, on the raw arrays the last upload left on the device: routing and
 * validation, op tables, tile lists, unpacked bases (what pileup_callback + htslib's bam_plp_auto do per read,
 * snp_score.c:1113-1153).  bench.py times this together with crumble_hip_run(): a contig pays both, once. */
int crumble_hip_expand(crumble_hip *h);
int crumble_hip_run(crumble_hip *h);      /* asynchronous launches + one sync */
int crumble_hip_download(crumble_hip *h, uint8_t *qual_out,
                         uint8_t *read_flags_out, int64_t *stats);

/* ---- coded download: a quarter of the bytes over PCIe ---------------------------------------------------------
 * What the host does with the result is flush_bam_list()'s "replace the record's qualities" (snp_score.c:1081-1105),
 * and it still holds the input qualities.  After the pass almost every output byte is one of two values (qhigh / qlow
 * at the numeric presets) or the read's own input quality after cap_quality (snp_score.c:1325-1332).  So instead of one
 * byte per base the device can hand back 2 bits per base,
 *     0 = value[0]   1 = value[1]   2 = cap[input quality]   3 = the next byte of the escape stream,
 * 16 bases to a 32-bit word (base i of the flat base array in bits 2 * (i & 15) of codes[i >> 4]), plus the escape
 * bytes (P-block averages, bin2[] levels) in base order inside every block of CRUMBLE_CODED_BLOCK bases, block b's
 * starting at esc[esc_off[b]].  Expanded on the host it is byte for byte what crumble_hip_download() returns.
 *
 *   crumble_hip_encode(h, &c)            after crumble_hip_run(): encodes on the device, fills the sizes and tables of c
 *   (caller provides c.codes[c.n_words], c.esc_off[c.n_blocks], c.esc[c.n_escapes] -- any host memory)
 *   crumble_hip_download_coded(h, &c, flags, stats)       the three streams, the per-read flags, the statistics
 *   crumble_decode_seek / crumble_decode_run              the host-side expansion (plain C, no device), or
 *   crumble_decode_quals(&c, quals_in, qual_out, threads) the whole batch at once
 * A batch whose escapes are many (c.n_escapes above n_bases / 8, say: long reads with P-block) is better served by
 * crumble_hip_download(); both stay valid after crumble_hip_encode(). */
#define CRUMBLE_CODED_BLOCK 4096
typedef struct crumble_coded {
    int64_t n_bases;
    int64_t n_words;            /* (n_bases + 15) / 16 code words                                   */
    int64_t n_blocks;           /* (n_bases + CRUMBLE_CODED_BLOCK - 1) / CRUMBLE_CODED_BLOCK        */
    int64_t n_escapes;          /* bytes of the escape stream                                        */
    uint8_t value[2];           /* the bytes codes 0 and 1 stand for                                 */
    uint8_t pad_[6];
    uint8_t cap[256];           /* code 2: cap[input quality] (cap_quality with this engine's -U / -k / -K) */
    uint32_t *codes;            /* caller's buffers, filled by crumble_hip_download_coded            */
    int64_t  *esc_off;
    uint8_t  *esc;
} crumble_coded;
int crumble_hip_encode(crumble_hip *h, crumble_coded *c);
int crumble_hip_download_coded(crumble_hip *h, const crumble_coded *c, uint8_t *read_flags_out, int64_t *stats);
/* Host-side expansion.  A cursor walks the flat base array: seek to any base (counts the escapes of the block in
 * front of it: at most CRUMBLE_CODED_BLOCK / 16 words), then run over consecutive bases.  quals_in / qual_out point at
 * the FIRST base of the run (they may be the same memory: a record's quality field is rewritten in place);
 * qual_out == NULL skips the run (a record without qualities). */
typedef struct crumble_decode_cursor { int64_t base, esc_pos; } crumble_decode_cursor;
void crumble_decode_seek(const crumble_coded *c, int64_t base, crumble_decode_cursor *cur);
void crumble_decode_run(const crumble_coded *c, crumble_decode_cursor *cur, int64_t n, const uint8_t *quals_in, uint8_t *qual_out);
/* the whole batch: qual_out[n_bases] from quals_in[n_bases] on `threads` host threads (0: all the process may use) */
void crumble_decode_quals(const crumble_coded *c, const uint8_t *quals_in, uint8_t *qual_out, int threads);

/* ---- one contig over several devices: windows ------------------------------------------------
 * transcode() carries three pieces of state from column to column: the running depth average total_col / total_depth
 * with its halvings (snp_score.c:1490-1491, 1684-1687), the STR window min_pos max_pos min_pos2 max_pos2
 * (snp_score.c:1508-1511, 1741-1755) and the BED cursor (a search here).  A contig is cut at reference positions
 * w_0 < w_1 < ...; window k OWNS the reads that start in [w_k, w_k+1) and is uploaded as a batch of the contig's reads
 * that overlap [w_k, E_k), E_k = max(the largest end of an owned read, w_k+1) (reads are in file order, so that is one
 * contiguous slice: owned reads, the reads before them that reach w_k, the reads after them that start before E_k;
 * E_k >= w_k+1 because a read of an EARLIER window may still cover columns behind the owned reads' end).  The window
 * decides the columns [w_k, E_k) -- every read that covers them is in the batch -- counts the columns [w_k, w_k+1)
 * and the owned reads in the statistics, and starts from the state window k-1 reached at w_k.  Everything else is
 * what a whole contig does; the owned reads' bytes, flags and the summed statistics equal the unsharded run's.
 *
 * crumble_hip_run() is five phases; a caller with one engine per device runs crumble_hip_run_columns, _triggers and
 * _finish of all windows at once and only _depth and _fold (a few small kernels each) in window order, handing the
 * crumble_carry on. */
typedef struct crumble_window {
    int32_t col_start;      /* w_k: first reference position whose column this window decides               */
    int32_t col_end;        /* E_k: one past the last decided column (>= every owned read's end, >= stat_end) */
    int32_t stat_end;       /* w_k+1: columns before it enter the statistics; the carry is reported there    */
    int32_t pad_;
    int64_t read_first;     /* owned reads [read_first, read_end) of the uploaded batch                      */
    int64_t read_end;
} crumble_window;
typedef struct crumble_carry {
    int64_t total_col, total_depth;                 /* <- / -> crumble_hip_run_depth */
    int32_t min_pos, max_pos, min_pos2, max_pos2;   /* <- / -> crumble_hip_run_fold  */
} crumble_carry;
/* After crumble_hip_upload(), before the run.  NULL: the whole batch again.  crumble_hip_download() still copies
 * every read of the batch; only the owned ones are final. */
int crumble_hip_set_window(crumble_hip *h, const crumble_window *w);
void crumble_carry_init(crumble_carry *c);          /* the state at the start of a contig (snp_score.c:1478-1488) */
int crumble_hip_run_columns(crumble_hip *h);
int crumble_hip_run_depth(crumble_hip *h, const crumble_carry *in, crumble_carry *out);   /* in/out may be NULL */
int crumble_hip_run_triggers(crumble_hip *h);
int crumble_hip_run_fold(crumble_hip *h, const crumble_carry *in, crumble_carry *out);
int crumble_hip_run_finish(crumble_hip *h);

/* HIP-event timings of the last crumble_hip_run() (option "stages" = on; 0 stages otherwise): names[i] / ms[i] for up to
 * `max` kernels (returns the number filled).  bytes[i] (may be NULL) receives
 * the algorithmic bytes that launch is charged with. */
int crumble_hip_kernel_times(crumble_hip *h, const char **names, float *ms,
                             double *bytes, int max);

/* Stream time of the last crumble_hip_upload(): the host->device copies of the raw arrays, and the device-side
 * expansion (P0: routing, validation, op tables, tile ranges) including its two small size read-backs. */
int crumble_hip_upload_times(crumble_hip *h, float *h2d_ms, float *p0_ms);

/* Bytes that crossed PCIe: what the last crumble_hip_upload / _contig copied from the host, and what the last
 * crumble_hip_download / _download_coded copied back. */
int crumble_hip_transfer_bytes(crumble_hip *h, int64_t *up, int64_t *down);

/* Accumulated device-side stats vector (int64[CRUMBLE_ST_N]) of this engine:
 * a raw device pointer so torch.distributed / RCCL can all-reduce it in place. */
void *crumble_hip_stats_devptr(crumble_hip *h);
int   crumble_hip_stats_reset(crumble_hip *h);
int   crumble_hip_stats_read(crumble_hip *h, int64_t *stats);
/* Adds a host vector (int64[CRUMBLE_ST_N]) to the accumulated one: a host that runs several engines on one device folds
 * their vectors into one engine before that device's single crumble_hip_stats_allreduce(). */
int   crumble_hip_stats_add(crumble_hip *h, const int64_t *stats);
/* Sums the stats vector over all ranks of an RCCL communicator (ncclComm_t
 * passed as void*). The only collective on the path. */
int   crumble_hip_stats_allreduce(crumble_hip *h, void *nccl_comm);
/* For a single-process multi-GPU host (crumble-hip -j N): one communicator per device (ncclCommInitAll);
 * comms[i] goes with devices[i].  The all-reduce is then called once per engine, concurrently. */
int   crumble_hip_comm_init_all(int n, const int *devices, void **comms);
void  crumble_hip_comm_destroy(void *comm);

/* Operator-level seams (used by the parity tests; same maths as the engine):
 *  calculate_consensus_pileup (snp:533) over every column of a batch ->
 *  per-position records; find_STR/mask_LC_regions (str_finder.c:135, snp:1230)
 *  over explicit (read, qpos+1, is_indel) queries; pblock (snp:803) per read. */
/* why a column forced keep_qual: the reasons reference crumble writes to its -b BED file
 * (snp_score.c:1676 DEEP, 1768 CLIP, 1802 INDEL_LEN, 1810 INDEL_COVERAGE; 1496 VDEEP is skipped == 2).
 * One line "chrom  max(pos-50,0)  pos+50  REASON" per set bit, in this order, per column. */
#define CRUMBLE_WHY_VDEEP           0x08   /* only from crumble_hip_suspect(): n_plp > 20000 */
#define CRUMBLE_WHY_DEEP            0x10
#define CRUMBLE_WHY_CLIP            0x20
#define CRUMBLE_WHY_INDEL_LEN       0x40
#define CRUMBLE_WHY_INDEL_COVERAGE  0x80

typedef struct crumble_column {        /* one pileup column, after step 18 of  */
    int32_t pos;                       /* SURVEY.md App. A.2                   */
    int32_t n_plp;
    int32_t score;                     /* sB (or sA)  snp:1575-1591            */
    uint8_t call1, call2;              /* allele index 0..5 (A C G T * N)      */
    uint8_t preserve;                  /* 0..3 (bit1 = BED)                    */
    uint8_t perfect;
    uint8_t keep_qual;
    uint8_t trigger;                   /* `indel` != 0  snp:1870               */
    uint8_t window;                    /* min_pos != INT_MAX  snp:1880         */
    uint8_t skipped;                   /* 1 all-refskip, 2 too deep (= the -b  */
                                       /* reason VDEEP); else CRUMBLE_WHY_* bits */
    int32_t min_pos2;                  /* valid when trigger                   */
} crumble_column;
/* Fills cols[0..max) with the columns of the uploaded batch in position order
 * after crumble_hip_run(); returns the number of columns (may exceed max). */
int64_t crumble_hip_columns(crumble_hip *h, crumble_column *cols, int64_t max);

/* The suspicious columns of the last crumble_hip_run() in position order: pos[k], why[k] =
 * CRUMBLE_WHY_* bits; returns their number (may exceed max).  This is what crumble -b prints. */
int64_t crumble_hip_suspect(crumble_hip *h, int32_t *pos, uint8_t *why, int64_t max);

int crumble_hip_mask_lc(crumble_hip *h, int64_t n, const int64_t *read_idx,
                        const int32_t *qpos1, const uint8_t *is_indel,
                        int32_t *lo, int32_t *hi);
int crumble_hip_pblock(crumble_hip *h, int64_t n_reads, const int64_t *seq_off,
                       uint8_t *quals /* in/out, host */);

/* indel_only.c (the reference's stand-alone toy, indel_only.c:83-122): keep the qualities
 * within +-dist query bases of every pileup entry with indel != 0 or is_del (reference:
 * D = 20), set all other qualities of reads that reach a pileup column to q (reference:
 * Q = 37).  Uses the batch of the last crumble_hip_upload(). */
int crumble_hip_indel_only(crumble_hip *h, int dist, int q, uint8_t *qual_out);

/* ---- BGZF members inflated / deflated on the device ----------------------------------------------
 * What htslib's bgzf layer does on host threads under sam_read1 (snp_score.c:1116-1117) and sam_write1
 * (snp_score.c:1101): every 64 KiB member is an independent DEFLATE stream (RFC 1951/1952, SAMv1 4.1).  A crumble_gz is a
 * codec context on one device (its own stream and buffers, independent of any engine; calls on one handle are serial).
 *
 *   crumble_bgzf_scan(comp, n, blk, max, &consumed, &raw_total)   host only: walks the gzip headers of the whole members in
 *       comp[0, n), fills blk[] (payload position, ISIZE, CRC-32, uoff = running sum of ISIZE); returns their number (may
 *       stop at max), < 0 for a malformed header; consumed = bytes of whole members walked
 *   crumble_gz_inflate(g, comp, n, blk, n_blocks, raw, raw_len, check_crc)   raw[blk[i].uoff ...] = member i inflated;
 *       a corrupt member (bad code lengths, distance too far back, wrong size, CRC-32) -> CRUMBLE_EINVAL, text in
 *       crumble_hip_last_error()
 *   crumble_gz_deflate(g, raw, n, level, out, out_cap, &out_len)   raw cut into members of 0xff00 bytes (htslib's block
 *       size), each a complete gzip member with the BC extra field; out_cap >= crumble_gz_deflate_bound(n); no EOF member;
 *       level 0 = stored, anything else = the device's one level (hash + run + repeat-distance candidates, greedy parse,
 *       dynamic Huffman codes).  Any inflater reads the result; the bytes differ from zlib's (as zlib's do from libdeflate's). */
typedef struct crumble_gz crumble_gz;
typedef struct crumble_gz_block {
    int64_t  coff;              /* first byte of the member's DEFLATE payload in the compressed buffer */
    int64_t  uoff;              /* where its bytes go in the uncompressed buffer                     */
    uint32_t clen, ulen, crc;   /* payload bytes; ISIZE; CRC-32 of the uncompressed bytes            */
    uint32_t pad_;
} crumble_gz_block;
crumble_gz *crumble_gz_create(int device, int *err);
void    crumble_gz_destroy(crumble_gz *g);
int64_t crumble_bgzf_scan(const uint8_t *comp, int64_t comp_len, crumble_gz_block *blk, int64_t max_blocks,
                          int64_t *consumed, int64_t *raw_total);
int     crumble_gz_inflate(crumble_gz *g, const uint8_t *comp, int64_t comp_len, const crumble_gz_block *blk, int64_t n_blocks,
                           uint8_t *raw, int64_t raw_len, int check_crc);
int64_t crumble_gz_deflate_bound(int64_t raw_len);
int     crumble_gz_deflate(crumble_gz *g, const uint8_t *raw, int64_t raw_len, int level, uint8_t *out, int64_t out_cap, int64_t *out_len);
/* implementation choices that never change a result: "tokens" = auto | wave | lane (k_gz_tokens: a wave per member, the
 * right thing for a few thousand members; k_gz_tokens_lanes: a lane per member, for tens of thousands and more) */
int     crumble_gz_set_option(crumble_gz *g, const char *name, const char *value);
/* stream time of the last inflate / deflate call on g: host -> device copies, kernels, device -> host copies (ms) */
int     crumble_gz_times(crumble_gz *g, float *h2d_ms, float *kernels_ms, float *d2h_ms);

/* ---- a BAM file's records on the device from inflate to deflate (round 6) --------------------------
 * What sam_read1 / pileup_callback (snp_score.c:1113-1153) and flush_bam_list / sam_write1 (snp_score.c:1081-1105) do on
 * the host around transcode()'s loop, with the records never leaving the device: the host reads compressed bytes, walks the
 * BGZF member headers (crumble_bgzf_scan) and writes compressed bytes; inflate, finding the records, the engine's input, its
 * output back into the records and deflate all happen in HBM.
 *
 *   s = crumble_stream_create(device, n_ref, header_bytes, &err)   header_bytes = size of the BAM header in the inflated
 *       stream (magic, text, references: the host parses it from the first members and writes the output's own)
 *   crumble_stream_feed(s, comp, n, blk, n_blocks, eof)            whole members, in file order
 *   n = crumble_stream_groups(s, g, max)                           the contig runs in the buffer (refID, reads, bytes); all but
 *       the last are complete, the last one too once eof was fed
 *   crumble_stream_process(s, engine, i, n_bed, bed_start, bed_end)   run i through an engine of the same device (up to four
 *       engines, a host thread each, may process different runs at once; feed / drain / groups: one thread, nothing beside them)
 *   crumble_stream_drain(s, k, level, flush, out, cap, &len)       the first k runs are done: they join what waits at the front of
 *       the buffer (crumble_stream_pending), and as many whole BGZF members of 0xff00 bytes as cap holds leave; with flush the
 *       last, shorter one too when everything fits -- a contig of any size through a fixed output buffer
 * Files with a CIGAR of more than 65535 operations (CG tag) are refused by process (CRUMBLE_EINVAL): the caller takes the
 * host path.  A record stream that cannot be delimited (the chain of block_size fields is checked) is refused by feed. */
typedef struct crumble_stream crumble_stream;
typedef struct crumble_stream_group {
    int32_t tid;            /* refID of the run (-1: the unmapped tail)                           */
    int32_t complete;       /* 1: every record of the run is in the buffer                         */
    int64_t first_read;     /* index of its first record among the buffer's records                */
    int64_t n_reads;
    int64_t n_bases;        /* filled by crumble_stream_process                                    */
    int64_t byte_start, byte_end;   /* its records' bytes in the inflated stream held by the buffer */
} crumble_stream_group;
crumble_stream *crumble_stream_create(int device, int32_t n_ref, int64_t header_bytes, int *err);
void crumble_stream_destroy(crumble_stream *s);
int  crumble_stream_feed(crumble_stream *s, const uint8_t *comp, int64_t comp_len, const crumble_gz_block *blk, int64_t n_blocks, int eof);
int  crumble_stream_groups(crumble_stream *s, crumble_stream_group *out, int max);
int  crumble_stream_process(crumble_stream *s, crumble_hip *h, int group, int32_t n_bed, const int32_t *bed_start, const int32_t *bed_end);
int  crumble_stream_drain(crumble_stream *s, int k, int level, int flush, uint8_t *out, int64_t out_cap, int64_t *out_len);
int64_t crumble_stream_pending(crumble_stream *s);   /* bytes processed and not drained yet */
/* the drain in two halves: _begin marks the runs, enqueues the deflate of what a buffer of out_cap bytes will take (*taken
 * record bytes; 0: nothing to do and no _end needed; out_cap at most a codec round, 256 MB of records) and returns; _end waits
 * and copies the members out.  In between further runs may be processed (other streams, other bytes) -- no feed, no drain. */
int  crumble_stream_drain_begin(crumble_stream *s, int k, int level, int flush, int64_t out_cap, int64_t *taken);
int  crumble_stream_drain_end(crumble_stream *s, uint8_t *out, int64_t out_cap, int64_t *out_len);
/* optional set-up before the first feed: device memory for feeds of up to comp_bytes that inflate to about raw_bytes (record
 * buffers, token and slot arrays), so that no feed / drain stops to allocate; larger feeds still work (they grow the buffers) */
int  crumble_stream_reserve(crumble_stream *s, int64_t comp_bytes, int64_t raw_bytes);
/* tests: bytes [from, from + n) of the buffer, its length, its record count and record offsets */
int  crumble_stream_peek(crumble_stream *s, int64_t from, int64_t n, uint8_t *out, int64_t *len, int64_t *n_rec, int64_t *rec_off, int64_t max_rec);
/* host-clock milliseconds so far -- ms[0..4] = inflate (H2D + kernels), record delimiting, engine (fields, expansion, run,
 * scatter), deflate (kernels + D2H), moving what is left of the buffer -- and the compressed bytes in / out */
int  crumble_stream_times(crumble_stream *s, float ms[5], int64_t *bytes_in, int64_t *bytes_out);

/* Page-locked host memory (hipHostMalloc) for a caller's batch arrays and output buffers: copies to and from it cross
 * PCIe at the link's rate without the runtime's staging through its own buffers (pageable memory: about half).  Any host
 * memory works with every entry point; this is an option for callers that keep their buffers.  NULL on failure. */
void *crumble_hip_host_alloc(size_t bytes);
/* diagnostic: device allocations (hipFree + hipMalloc of a growing buffer) of this process so far and the host time they took */
void  crumble_hip_alloc_stats(int64_t *calls, double *ms);
void  crumble_hip_host_free(void *p);

const char *crumble_hip_strerror(int err);
/* Text of the last HIP/RCCL runtime error seen by this thread. */
const char *crumble_hip_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* CRUMBLE_HIP_H */
