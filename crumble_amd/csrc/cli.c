/*
 * cli.c -- `crumble-hip`: file -> file driver around the HIP column engine.
 *
 * The same shape as reference crumble's main() + transcode() (snp_score.c:2144-2680, 1336-2030)
 * with the loop body replaced by crumble_hip_contig(): options are parsed with the reference's own
 * getopt string (snp_score.c:2199) in argv order (presets overwrite, later options refine), records
 * are gathered per contig (the unit the reference resets its state on, snp_score.c:1478-1488), the
 * engine rewrites the qualities, records go out in input order.  BAM only (no htslib here): options
 * that belong to htslib / host plumbing that was not rebuilt (-I -O -r) are refused
 * rather than silently ignored.  There is no CPU path: without a GPU the program stops with
 * CRUMBLE_ENODEVICE.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <unistd.h>
#include <time.h>
#include "../../include/crumble_hip.h"
#include "../../include/crumble_bamio.h"

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

static void usage(FILE *fp)
{
    fprintf(fp,
        "crumble-hip: crumble's quality quantisation on an AMD MI355X (BAM in, BAM out)\n"
        "Usage: crumble-hip [options] in.bam out.bam\n"
        "  -1 -3 -5 -7 -8 -9       compression presets (as crumble)\n"
        "  -y illumina|pbccs       technology profile\n"
        "  -q -d -x  /  -Q -D -X   min SNP qual, indel qual, discrepancy (consensus A / B)\n"
        "  -m N  -l N  -u N  -c N  -U N   min mapq, qlow, qhigh, qcutoff, qcap\n"
        "  -i mul[,add]  -s mul[,add]     indel / SNP STR window scaling\n"
        "  -L 0|1  -B              reduce qualities; binary qualities\n"
        "  -C -M -Z -P -Y -V f     clip / low-mapq / ins-len / over-depth / indel fraction / indel overlap\n"
        "  -p N   -S   -N          P-block, soft-clip quantisation, perfect columns\n"
        "  -k list  -K list        qualities to preserve (-K: always)\n"
        "  -R keep.bed             regions whose qualities are kept\n"
        "  -b out.bed              write the suspicious regions (DEEP, CLIP, INDEL_LEN, ...) as BED\n"
        "  -t list  -T list        auxiliary tags to keep (white list) / to drop (black list)\n"
        "  -e -f -g / -E -F -G n   BD / BI tag low, cutoff, high values\n"
        "  -z                      no @PG header line\n"
        "  -v                      statistics and timings on stderr\n"
        "  -@ N                    host threads for BGZF/BAM work\n"
        "  (device: environment CRUMBLE_HIP_DEVICE, default 0; output level: CRUMBLE_BAM_LEVEL)\n");
}

static const char *const ST_NAME[] = {
    "columns", "het_A", "het_qual_A", "hom_A", "hom_qual_A", "discrep_A", "het_B", "het_qual_B", "hom_B",
    "hom_qual_B", "discrep_B", "diff", "indel", "indel_qual", "low_mqual_perc", "clip_perc", "ins_len_perc",
    "indel_ov_perc", "over_depth", "reads", "bases", "bases_changed", "triggers", "too_deep" };

int main(int argc, char **argv)
{
    crumble_params P;
    crumble_params_default(&P);
    int opt, verbose = 0, threads = 0, no_pg = 0;
    const char *bed_fn = NULL, *wl = NULL, *bl = NULL;
    int bd[3] = {0, 0, 0}, bi[3] = {0, 0, 0};
    FILE *bed_fp = NULL;
    while ((opt = getopt(argc, argv, "I:O:q:d:x:Q:D:X:m:l:u:U:c:i:L:Bs:t:T:hr:b:vC:M:Z:P:V:p:e:f:g:E:F:G:S135789zR:Y:y:k:K:N@:")) != -1) {
        switch (opt) {
        case 'q': P.min_qual_A = atoi(optarg); break;
        case 'd': P.min_indel_A = atoi(optarg); break;
        case 'x': P.min_discrep_A = atof(optarg); break;
        case 'Q': P.min_qual_B = atoi(optarg); break;
        case 'D': P.min_indel_B = atoi(optarg); break;
        case 'X': P.min_discrep_B = atof(optarg); break;
        case 'm': P.min_mqual = atoi(optarg); break;
        case 'l': P.qlow = atoi(optarg); break;
        case 'u': P.qhigh = atoi(optarg); break;
        case 'c': P.qcutoff = atoi(optarg); break;
        case 'U': P.qcap = atoi(optarg); break;
        case 'i': P.iSTR_mul = atof(optarg); if (strchr(optarg, ',')) P.iSTR_add = atoi(strchr(optarg, ',') + 1); break;
        case 's': P.sSTR_mul = atof(optarg); if (strchr(optarg, ',')) P.sSTR_add = atoi(strchr(optarg, ',') + 1); break;
        case 'L': P.reduce_qual = atoi(optarg); break;
        case 'B': P.binary_qual = 1; break;
        case 'R': bed_fn = optarg; break;
        case 'C': P.clip_perc = atof(optarg); break;
        case 'M': P.low_mqual_perc = atof(optarg); break;
        case 'Z': P.ins_len_perc = atof(optarg); break;
        case 'P': P.over_depth = atof(optarg); break;
        case 'Y': P.indel_fract = atof(optarg); break;
        case 'V': P.indel_ov_perc = atof(optarg); break;
        case 'p': P.pblock = atoi(optarg); break;
        case 'y': crumble_params_tech(&P, optarg); break;
        case 'k': crumble_params_keep_quals(&P, optarg, 0); break;
        case 'K': crumble_params_keep_quals(&P, optarg, 1); break;
        case 'N': P.perfect_col = 1; break;
        case 'S': P.softclip = 1; break;
        case '1': case '3': case '5': case '7': case '8': case '9': crumble_params_preset(&P, opt - '0'); break;
        case 'z': no_pg = 1; break;
        case 'v': verbose++; break;
        case '@': threads = atoi(optarg); break;
        case 'h': usage(stdout); return 0;
        case 't': wl = optarg; break;
        case 'T': bl = optarg; break;
        case 'e': bd[0] = atoi(optarg) + 33; break;
        case 'f': bd[1] = atoi(optarg) + 33; break;
        case 'g': bd[2] = atoi(optarg) + 33; break;
        case 'E': bi[0] = atoi(optarg) + 33; break;
        case 'F': bi[1] = atoi(optarg) + 33; break;
        case 'G': bi[2] = atoi(optarg) + 33; break;
        case 'b': if (!(bed_fp = fopen(optarg, "w"))) { perror(optarg); return 1; } break;
        case 'I': case 'O': case 'r':
            fprintf(stderr, "crumble-hip: option -%c (htslib formats / regions) is host "
                            "plumbing that this build does not provide\n", opt);
            return 1;
        default: usage(stderr); return 1;
        }
    }
    if (argc - optind != 2) { usage(stderr); return 1; }
    const char *in_fn = argv[optind], *out_fn = argv[optind + 1];
    const int device = getenv("CRUMBLE_HIP_DEVICE") ? atoi(getenv("CRUMBLE_HIP_DEVICE")) : 0;
    const int level = getenv("CRUMBLE_BAM_LEVEL") ? atoi(getenv("CRUMBLE_BAM_LEVEL")) : -1;

    int err = 0;
    crumble_hip *eng = crumble_hip_create(&P, device, &err);
    if (!eng) { fprintf(stderr, "crumble-hip: %s (%s)\n", crumble_hip_strerror(err), crumble_hip_last_error()); return 2; }
    crumble_bam_reader *rd = crumble_bam_open(in_fn, threads, &err);
    if (!rd) { fprintf(stderr, "crumble-hip: %s\n", crumble_bam_last_error()); return 1; }

    int32_t *bt = NULL, *bs = NULL, *be = NULL; int nbed = 0;
    if (bed_fn && (nbed = crumble_bam_load_bed(rd, bed_fn, &bt, &bs, &be)) < 0) { fprintf(stderr, "crumble-hip: %s\n", crumble_bam_last_error()); return 1; }

    char *pg = NULL;
    if (!no_pg) {
        size_t len = 64;
        for (int i = 0; i < argc; i++) len += strlen(argv[i]) + 1;
        pg = malloc(len);
        char *q = pg + sprintf(pg, "@PG\tID:crumble-hip\tPN:crumble-hip\tVN:0.1\tCL:");
        for (int i = 0; i < argc; i++) q += sprintf(q, "%s%s", i ? " " : "", argv[i]);
    }
    crumble_bam_writer *wr = crumble_bam_create(out_fn, rd, pg, level, threads, &err);
    if (!wr) { fprintf(stderr, "crumble-hip: %s\n", crumble_bam_last_error()); return 1; }
    if (crumble_bam_set_aux_filter(wr, wl, bl, bd, bi)) { fprintf(stderr, "crumble-hip: %s\n", crumble_bam_last_error()); return 1; }

    int64_t stats[CRUMBLE_ST_N] = {0};
    uint8_t *qout = NULL; size_t qcap = 0;
    double t_read = 0, t_gpu = 0, t_write = 0;
    const double t_all = now();
    int bed_i = 0;
    for (;;) {
        crumble_batch B;
        double t0 = now();
        const int rc = crumble_bam_next_contig(rd, &B);
        t_read += now() - t0;
        if (rc < 0) { fprintf(stderr, "crumble-hip: %s\n", crumble_bam_last_error()); return 1; }
        if (rc == 0) break;
        while (bed_i < nbed && bt[bed_i] < B.tid) bed_i++;      /* groups arrive in header order, tid -1 last */
        int j = bed_i;
        while (j < nbed && bt[j] == B.tid) j++;
        if (B.tid >= 0 && j > bed_i) { B.n_bed = j - bed_i; B.bed_start = bs + bed_i; B.bed_end = be + bed_i; }
        const size_t nb = (size_t)B.seq_off[B.n_reads];
        if (qcap < nb + 16) {
            crumble_bam_big_free(qout, qcap);
            qcap = nb + nb / 8 + 64; qout = crumble_bam_big_alloc(qcap);
            if (!qout) { fprintf(stderr, "crumble-hip: out of memory\n"); return 1; }
        }
        t0 = now();
        const int e = crumble_hip_contig(eng, &B, qout, NULL, stats);
        t_gpu += now() - t0;
        if (e) { fprintf(stderr, "crumble-hip: contig %s: %s (%s)\n", crumble_bam_ref_name(rd, B.tid), crumble_hip_strerror(e), crumble_hip_last_error()); return 2; }
        if (bed_fp && B.tid >= 0) {                             /* -b: snp_score.c:1496,1676,1768,1802,1810 */
            const int64_t ns = crumble_hip_suspect(eng, NULL, NULL, 0);
            if (ns < 0) { fprintf(stderr, "crumble-hip: %s\n", crumble_hip_strerror((int)ns)); return 2; }
            int32_t *sp = malloc((size_t)(ns + 1) * 4); uint8_t *sw = malloc((size_t)ns + 1);
            if (!sp || !sw || crumble_hip_suspect(eng, sp, sw, ns) != ns) { fprintf(stderr, "crumble-hip: -b failed\n"); return 2; }
            static const struct { int bit; const char *name; } why[] = {
                { CRUMBLE_WHY_VDEEP, "VDEEP" }, { CRUMBLE_WHY_DEEP, "DEEP" }, { CRUMBLE_WHY_CLIP, "CLIP" },
                { CRUMBLE_WHY_INDEL_LEN, "INDEL_LEN" }, { CRUMBLE_WHY_INDEL_COVERAGE, "INDEL_COVERAGE" } };
            for (int64_t k = 0; k < ns; k++)
                for (int y = 0; y < 5; y++)
                    if (sw[k] & why[y].bit)
                        fprintf(bed_fp, "%s\t%d\t%d\t%s\n", crumble_bam_ref_name(rd, B.tid), sp[k] > 50 ? sp[k] - 50 : 0, sp[k] + 50, why[y].name);
            free(sp); free(sw);
        }
        t0 = now();
        if (crumble_bam_write_group(wr, rd, qout)) { fprintf(stderr, "crumble-hip: %s\n", crumble_bam_last_error()); return 1; }
        t_write += now() - t0;
        if (verbose > 1) fprintf(stderr, "  %s: %lld reads, %zu bases\n", crumble_bam_ref_name(rd, B.tid), (long long)B.n_reads, nb);
    }
    if (bed_fp && fclose(bed_fp)) { perror("-b"); return 1; }
    if (crumble_bam_writer_close(wr)) { fprintf(stderr, "crumble-hip: %s\n", crumble_bam_last_error()); return 1; }
    const double wall = now() - t_all;
    if (verbose) {
        for (int i = 0; i < (int)(sizeof ST_NAME / sizeof *ST_NAME); i++) fprintf(stderr, "%-16s %lld\n", ST_NAME[i], (long long)stats[i]);
        const double gb = stats[CRUMBLE_ST_BASES] / 1e9;
        fprintf(stderr, "end-to-end %.3f s (read+parse %.3f, upload+kernels+download %.3f, write %.3f): %.4f Gbases/s file to file\n",
                wall, t_read, t_gpu, t_write, wall > 0 ? gb / wall : 0.0);
        char tb[512];
        crumble_bam_timing(tb, sizeof tb);
        fprintf(stderr, "%s\n", tb);
    }
    crumble_bam_close(rd);
    crumble_hip_destroy(eng);
    crumble_bam_big_free(qout, qcap); free(bt); free(bs); free(be); free(pg);
    return 0;
}
