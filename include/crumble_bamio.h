/*
 * crumble_bamio.h -- host ingest / egress either side of the column engine.
 *
 * Reference crumble reads and writes through htslib: sam_read1() inside pileup_callback()
 * (snp_score.c:1116-1117) and sam_write1() inside flush_bam_list() (snp_score.c:1101).  htslib is
 * not part of /root/reference, so this is a from-the-spec BGZF + BAM (SAMv1 section 4) reader and
 * writer that produces exactly the SoA `crumble_batch` the engine consumes and writes the records
 * back with the engine's quality bytes.  BGZF blocks are inflated / deflated by a team of host threads (csrc/team.h)
 * (zlib per block), record fields are unpacked per read in parallel.
 *
 * Plain C ABI, no HIP.  Lives in libcrumble_bamio.so so that hosts which already have htslib need
 * not link it.  CIGARs of more than 65535 ops (CG:B,I tag, SAMv1 4.2.2) are read from the tag; the
 * record is written back as it came.  Not supported: CRAM/SAM text, BAM index output.
 */
#ifndef CRUMBLE_BAMIO_H
#define CRUMBLE_BAMIO_H

#include <stdint.h>
#include <stddef.h>
#include "crumble_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define CRUMBLE_BAM_EIO      -20   /* open/read/write failed                      */
#define CRUMBLE_BAM_EFORMAT  -21   /* not BGZF / not BAM / truncated / corrupt    */
#define CRUMBLE_BAM_ENOMEM   -22
#define CRUMBLE_BAM_EUNSUPP  -23   /* valid BAM feature this reader does not take */
#define CRUMBLE_BAM_ENOHTS   -24   /* SAM / CRAM / -I / -O need a system libhts (>= 1.10); none was found */

typedef struct crumble_bam_reader crumble_bam_reader;
typedef struct crumble_bam_writer crumble_bam_writer;
typedef struct crumble_bam_group  crumble_bam_group;

/* Opens a BAM file and parses its header.  threads <= 0: the CPUs the process may use (affinity, cgroup quota), at most 64. */
crumble_bam_reader *crumble_bam_open(const char *path, int threads, int *err);
void        crumble_bam_close(crumble_bam_reader *r);
int32_t     crumble_bam_n_ref(const crumble_bam_reader *r);
const char *crumble_bam_ref_name(const crumble_bam_reader *r, int32_t tid);
int64_t     crumble_bam_ref_len(const crumble_bam_reader *r, int32_t tid);
const char *crumble_bam_header_text(const crumble_bam_reader *r, int64_t *len);

/* BGZF members inflated / deflated by the caller instead of the library's host threads: crumble-hip hands the device codec
 * in (crumble_gz_inflate / crumble_gz_deflate of crumble_hip.h; this library itself stays free of HIP).  crumble_bam_block is
 * crumble_gz_block field for field.  inflate: blk[i].coff / clen locate member i's DEFLATE payload in comp[0, comp_len),
 * its ulen bytes go to raw[blk[i].uoff ...], crc is the member's CRC-32 (to be checked); deflate: raw[0, raw_len) as
 * members of 0xff00 input bytes side by side in out (out_cap >= deflate_bound(raw_len)), their total size in *out_len.
 * Both return 0 or a negative code.  Set before the first read / write; NULL = back to the host threads.  With a codec the
 * reader reads 256 MiB of compressed bytes per round. */
typedef struct crumble_bam_block { int64_t coff, uoff; uint32_t clen, ulen, crc, pad_; } crumble_bam_block;
typedef struct crumble_bam_coder {
    void *ctx;
    int (*inflate)(void *ctx, const uint8_t *comp, int64_t comp_len, const crumble_bam_block *blk, int64_t n_blocks,
                   uint8_t *raw, int64_t raw_len);
    int (*deflate)(void *ctx, const uint8_t *raw, int64_t raw_len, int level, uint8_t *out, int64_t out_cap, int64_t *out_len);
    int64_t (*deflate_bound)(int64_t raw_len);
} crumble_bam_coder;
void crumble_bam_reader_set_codec(crumble_bam_reader *r, const crumble_bam_coder *c);
void crumble_bam_writer_set_codec(crumble_bam_writer *w, const crumble_bam_coder *c);

/* Reads the next run of records sharing one refID (a contig, or the trailing tid -1 block: the
 * unit transcode() resets its state on, snp_score.c:1478-1488) and fills *b with SoA views owned
 * by the reader, valid until the next call.  b->n_bed is 0 (the caller attaches keep regions).
 * Missing qualities (0xff, SAMv1 4.2.3) are presented as 0 and written back untouched.
 * Returns 1 when a group was produced, 0 at end of file, < 0 on error. */
int crumble_bam_next_contig(crumble_bam_reader *r, crumble_batch *b);
/* on != 0: batches carry the bases as BAM packs them (crumble_batch.bases4; bases == NULL) -- no nibble expansion on
 * the host and half the host -> device bytes (sam_read1's bam_get_seq layout handed on as it is).  From the next
 * crumble_bam_next_contig() on. */
void crumble_bam_set_packed_bases(crumble_bam_reader *r, int on);

/* Takes the group that crumble_bam_next_contig() just returned out of the reader: its record bytes and the arrays
 * behind the crumble_batch now belong to the returned object (the batch's pointers stay valid until
 * crumble_bam_group_free) and the reader is free to read the next contig -- several contigs can be in flight
 * (one per GPU, one being written).  crumble_bam_write_detached() = crumble_bam_write_group() for such a group;
 * groups must be written in the order they were read. */
crumble_bam_group *crumble_bam_detach_group(crumble_bam_reader *r);
int  crumble_bam_write_detached(crumble_bam_writer *w, crumble_bam_group *g, const uint8_t *qual_out);
void crumble_bam_group_free(crumble_bam_group *g);

/* Creates an output BAM with the reader's header; pg_line (may be NULL) is a complete "@PG\t..."
 * line without newline that is appended to the header text (the reference adds its @PG through
 * sam_hdr_add_pg, snp_score.c:2588-2609).  level: zlib 0..9 (-1 default). */
crumble_bam_writer *crumble_bam_create(const char *path, const crumble_bam_reader *hdr_from,
                                       const char *pg_line, int level, int threads, int *err);
/* purge_tags() of the reference (snp_score.c:989-1054), applied to every record written:
 * whitelist (-t) or blacklist (-T) of two-character tags in the reference's "XX,YY,ZZ"
 * list syntax (a white list wins), and BD / BI quantisation (-e -f -g / -E -F -G): bd[] / bi[] =
 * {low, mid, high} as character values (the option value + 33), all 0 = off.  NULL = none. */
int crumble_bam_set_aux_filter(crumble_bam_writer *w, const char *whitelist, const char *blacklist,
                               const int bd[3], const int bi[3]);
/* Writes the reader's current group with qual_out (same layout as crumble_hip_contig's output;
 * NULL = qualities unchanged) replacing the quality bytes: flush_bam_list(), snp_score.c:1081-1105. */
int crumble_bam_write_group(crumble_bam_writer *w, crumble_bam_reader *r, const uint8_t *qual_out);
/* The same with the engine's result in coded form (crumble_hip_encode / crumble_hip_download_coded, include/crumble_hip.h):
 * 2 bits per base are expanded straight into the records' quality fields, which still hold the input qualities the
 * codes refer to -- no intermediate quality array on the host, a quarter of the bytes over PCIe. */
int crumble_bam_write_group_coded(crumble_bam_writer *w, crumble_bam_reader *r, const crumble_coded *c);
int crumble_bam_write_detached_coded(crumble_bam_writer *w, crumble_bam_group *g, const crumble_coded *c);
/* bytes of the BAM header (magic, text, reference list) in the inflated stream: where the first record starts */
int64_t crumble_bam_header_bytes(const crumble_bam_reader *r);
/* appends bytes that are whole BGZF members already (crumble_stream_drain's output) */
int crumble_bam_write_members(crumble_bam_writer *w, const uint8_t *members, size_t n);
/* Writes the BGZF EOF block and closes. */
int crumble_bam_writer_close(crumble_bam_writer *w);

/* Test / bench helper: writes SoA batches as a BAM (read names "r<n>", no aux tags, mate fields
 * empty).  Batches must be in header order. */
int crumble_bam_write_batches(const char *path, int32_t n_ref, const char *const *names,
                              const int64_t *lens, int32_t n_batches, const crumble_batch *batches,
                              int level, int threads);

/* "chrom<TAB>start<TAB>end" keep regions (-R, bed.c:42-105 bed_load) resolved against the
 * reader's reference names and collapsed per contig with crumble_bed_collapse().  Returns the
 * number of regions (arrays malloc'ed, sorted by tid then start; caller frees) or < 0. */
int crumble_bam_load_bed(const crumble_bam_reader *r, const char *path,
                         int32_t **tid, int32_t **start, int32_t **end);

/* htslib backend: SAM / BAM / CRAM in and out through a system libhts.so.3 resolved with dlopen at run time -- what
 * reference crumble's main() does with sam_open_format / sam_hdr_read / sam_read1 / sam_write1 / sam_hdr_add_PG
 * (snp_score.c:2201-2207, 2561-2626).  Same group contract as the native reader; without a usable libhts every
 * entry point fails with CRUMBLE_BAM_ENOHTS (crumble_hts_available() returns 0) and BAM keeps working through the
 * native reader / writer.  in_fmt / out_fmt are the reference's -I / -O strings (hts_parse_format), NULL = by
 * file content / name.  pg_name NULL = no @PG line (-z). */
typedef struct crumble_hts crumble_hts;
int  crumble_hts_available(char *version, size_t len);
crumble_hts *crumble_hts_open(const char *in_fn, const char *in_fmt, const char *out_fn, const char *out_fmt,
                              const char *pg_name, const char *pg_version, const char *pg_cmdline, int threads, int *err);
int32_t     crumble_hts_n_ref(const crumble_hts *h);
const char *crumble_hts_ref_name(const crumble_hts *h, int32_t tid);
int  crumble_hts_load_bed(const crumble_hts *h, const char *path, int32_t **tid, int32_t **start, int32_t **end);
int  crumble_hts_set_aux_filter(crumble_hts *h, const char *whitelist, const char *blacklist, const int bd[3], const int bi[3]);
int  crumble_hts_next_contig(crumble_hts *h, crumble_batch *b);
int  crumble_hts_write_group(crumble_hts *h, const uint8_t *qual_out);
int  crumble_hts_close(crumble_hts *h);

/* Huge-page backed host buffers for multi-GB arrays (e.g. the caller's qual_out): mmap +
 * MADV_HUGEPAGE; release with the same size. */
void *crumble_bam_big_alloc(size_t n);
void  crumble_bam_big_free(void *p, size_t n);

/* Where the host time went so far (process-wide phase timers), as one line of text. */
int crumble_bam_timing(char *buf, size_t len);
/* "libdeflate" (found on the host at run time) or "zlib": what inflates / deflates BGZF blocks in this process;
 * CRUMBLE_BAM_ZLIB=1 in the environment keeps zlib */
const char *crumble_bam_codec(void);
/* the team size `threads <= 0` stands for: the CPUs this process may use (affinity mask, cgroup quota), at most 64 */
int crumble_bam_default_threads(void);

const char *crumble_bam_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
