/*
 * crumble_synth.h -- seeded synthetic alignment generator.  TEST / BENCH INFRASTRUCTURE: built into its own
 * tests/synth/libcrumble_synth.so, never part of the product libraries.
 *
 * There is no NA12878 BAM, no htslib and no network in the build or GPU images,
 * so the benchmark configurations of BASELINE.json are generated on the fly:
 * a random diploid genome with short tandem repeats, SNP and indel variants,
 * Poisson read starts, Illumina-like binned qualities (or HiFi-like), sequencing
 * substitution/indel errors, soft clips, mapq-0 repeat blocks and a few
 * unmapped mates.  Output is a crumble_batch whose arrays are
 * owned by the generator.  Deterministic for a given cfg (independent of the
 * number of threads).
 */
#ifndef CRUMBLE_SYNTH_H
#define CRUMBLE_SYNTH_H
#include <stdint.h>
#include "../../include/crumble_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct crumble_synth_cfg {
    uint64_t seed;
    int32_t  tid;
    int32_t  contig_len;       /* reads start in [0, contig_len - read_len)        */
    int32_t  read_len;         /* mean read length                                 */
    int32_t  read_len_sigma;   /* 0: fixed length; else log-normal-ish spread (%)  */
    double   depth;            /* mean coverage                                    */
    double   depth_sigma;      /* per-10kb-block coverage modulation (fraction)    */
    int32_t  qual_model;       /* 0 Illumina 4-bin, 1 uniform 20..41, 2 HiFi       */
    double   het_snp, hom_snp; /* per-position variant rates                       */
    double   indel_var;        /* per-position indel variant rate                  */
    double   err_indel;        /* per-base sequencing indel error rate             */
    int32_t  str_every;        /* one STR per this many bp (0: none)               */
    double   mapq0_frac;       /* fraction of genome in 50 kb mapq-0 blocks        */
    double   mapq_mid_frac;    /* reads with uniform mapq 1..59                    */
    double   softclip_frac;    /* reads with a 5..20 bp soft clip                  */
    double   unmapped_frac;    /* reads flagged BAM_FUNMAP (placed at mate pos)    */
    double   refskip_frac;     /* reads with an N (ref-skip) op                    */
    int32_t  target_len;       /* >0: panel mode, coverage only on targets         */
    int32_t  target_every;     /*      one target per this many bp                 */
    int32_t  spike_pos;        /* >=0: position of an ultra-deep pile              */
    int32_t  spike_depth;      /*      its depth                                   */
} crumble_synth_cfg;

typedef struct crumble_synth crumble_synth;

/* Named configurations of BASELINE.json ("ce", "chr20", "panel", "hifi");
 * scale < 1 shrinks contig_len (CI-sized variants). */
int crumble_synth_config(crumble_synth_cfg *cfg, const char *name, double scale);

crumble_synth *crumble_synth_generate(const crumble_synth_cfg *cfg, int n_threads);
const crumble_batch *crumble_synth_batch(const crumble_synth *s);
int64_t crumble_synth_n_bases(const crumble_synth *s);
void crumble_synth_free(crumble_synth *s);

#ifdef __cplusplus
}
#endif
#endif
