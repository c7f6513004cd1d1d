/*
 * crumble_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT (see crumble_oracle.c).
 * CPU restatement of reference crumble's column engine; the checker that the
 * HIP path is compared against, and bench.py's timed cpu_baseline ("port").
 */
#ifndef CRUMBLE_ORACLE_H
#define CRUMBLE_ORACLE_H
#include <stdint.h>
#include "../include/crumble_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* transcode() (snp_score.c:1336-2029) for one contig.  Same contract as
 * crumble_hip_contig(); `trace` (optional) receives one record per pileup
 * column in position order. */
int crumble_oracle_contig(const crumble_params *p, const crumble_batch *b,
                          uint8_t *qual_out, uint8_t *read_flags_out,
                          int64_t *stats, crumble_column *trace, int64_t trace_max,
                          int64_t *n_trace);

/* calculate_consensus_pileup (snp_score.c:533) on one synthetic column. */
int crumble_oracle_consensus(const crumble_params *p, int flags, int np,
                             const uint8_t *base16, const uint8_t *qual,
                             const uint8_t *mapq, const uint8_t *is_del,
                             const uint8_t *is_refskip,
                             int *call, int *het_call, int *het_phred,
                             int *phred, float *discrep, int *call_preserve);
/* pblock (snp_score.c:803). */
void crumble_oracle_pblock(uint8_t *qual, int len, int level, int qcap,
                           const uint8_t *preserve_qual);
/* find_STR (str_finder.c:135): returns the repeat count, fills up to max. */
int crumble_oracle_find_str(const char *cons, int len, int *starts, int *ends, int max);
/* ref2query_pos (snp:1156), bam_qpos2rpos (snp:1205), mask_LC_regions (snp:1230). */
int crumble_oracle_ref2query_pos(const crumble_batch *b, int64_t r, int pos);
int crumble_oracle_qpos2rpos(const crumble_batch *b, int64_t r, int qpos);
int crumble_oracle_mask_lc(const crumble_params *p, const crumble_batch *b,
                           int64_t r, int is_indel, int rpos, int *min_pos, int *max_pos);
/* The (qual, mapq) -> adjusted quality table of CONS_MQUAL (snp:632-638). */
void crumble_oracle_adjust_lut(uint8_t *lut);
/* pileup_callback routing (snp:1130-1149). */
int crumble_oracle_in_pileup(const crumble_batch *b, int64_t r);
/* bam_plp_auto columns (htslib), flattened. */
int64_t crumble_oracle_pileup(const crumble_batch *b, int64_t max,
                              int32_t *col_pos, int64_t *read, int32_t *qpos,
                              int32_t *indel, uint8_t *bits);
/* indel_only.c:83-122 */
int crumble_oracle_indel_only(const crumble_batch *b, int D, int Q, uint8_t *qual_out);
#ifdef __cplusplus
}
#endif
#endif
