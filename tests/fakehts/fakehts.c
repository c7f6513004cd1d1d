/*
 * fakehts.c -- TEST INFRASTRUCTURE ONLY: a stand-in for the handful of libhts.so.3 entry points that
 * crumble_amd/csrc/bamio.c resolves with dlopen (its "htslib backend": snp_score.c:2201-2207, 2561-2626,
 * 1116-1117, 1101 in the reference).  It reads and writes SAM TEXT only and lays bam1_t out as htslib >= 1.10
 * does, so that hts_load(), hts_layout_check(), the record conversion and `crumble-hip -I sam -O sam` EXECUTE in
 * an image that has no htslib.  It pins no parity with the real library: nothing here was checked against htslib
 * (there is none to check against), and no product code links it -- the tests put this directory on
 * LD_LIBRARY_PATH of a child process.  Built with -DFAKEHTS_SHIFT it lays bam1_core_t out differently (as a
 * library with another ABI would): the backend's layout self-check must then refuse it.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <stdarg.h>
#include <ctype.h>

typedef struct {
#ifdef FAKEHTS_SHIFT
    int64_t some_new_field;             /* a different ABI: everything behind it moves */
#endif
    int64_t pos; int32_t tid; uint16_t bin; uint8_t qual, l_extranul; uint16_t flag, l_qname;
    uint32_t n_cigar; int32_t l_qseq, mtid; int64_t mpos, isize;
} bam1_core_t;
typedef struct { bam1_core_t core; uint64_t id; uint8_t *data; int l_data; uint32_t m_data; uint32_t mempolicy; } bam1_t;
typedef struct { size_t l, m; char *s; } kstring_t;

typedef struct { char *text; size_t l_text; int nref; char **names; int64_t *lens; } hdr_t;
typedef struct { FILE *fp; int wr; char *line; size_t cap; } file_t;

const char *hts_version(void) { return "1.19"; }

int hts_parse_format(void *fmt, const char *str)
{
    memset(fmt, 0, 32);
    return (str && (!strncmp(str, "sam", 3) && (str[3] == 0 || str[3] == ','))) ? 0 : -1;     /* SAM text only */
}
int sam_open_mode(char *mode, const char *fn, const char *format) { (void)fn; (void)format; mode[0] = 0; return 0; }
void *hts_open_format(const char *fn, const char *mode, const void *fmt)
{
    (void)fmt;
    file_t *f = calloc(1, sizeof *f);
    if (!f) return NULL;
    f->wr = mode[0] == 'w';
    f->fp = fopen(fn, f->wr ? "w" : "r");
    if (!f->fp) { free(f); return NULL; }
    if (!f->wr) {                                       /* SAM text only: anything that starts like a BGZF / CRAM file is refused */
        const int c = fgetc(f->fp);
        if (c == 0x1f || c == 'C') { fclose(f->fp); free(f); return NULL; }
        if (c != EOF) ungetc(c, f->fp);
    }
    return f;
}
int hts_close(void *fp) { file_t *f = fp; int rc = fclose(f->fp); free(f->line); free(f); return rc; }
int hts_set_threads(void *fp, int n) { (void)fp; (void)n; return 0; }

static void hdr_index(hdr_t *h)
{
    for (char *p = h->text; p && *p; ) {
        char *e = strchr(p, '\n');
        const size_t len = e ? (size_t)(e - p) : strlen(p);
        if (len > 3 && !strncmp(p, "@SQ", 3)) {
            char *line = strndup(p, len), *save = NULL, *nm = NULL;
            int64_t ln = 0;
            for (char *t = strtok_r(line, "\t", &save); t; t = strtok_r(NULL, "\t", &save)) {
                if (!strncmp(t, "SN:", 3)) nm = t + 3;
                else if (!strncmp(t, "LN:", 3)) ln = atoll(t + 3);
            }
            h->names = realloc(h->names, (size_t)(h->nref + 1) * sizeof *h->names);
            h->lens = realloc(h->lens, (size_t)(h->nref + 1) * sizeof *h->lens);
            h->names[h->nref] = strdup(nm ? nm : "?"); h->lens[h->nref] = ln; h->nref++;
            free(line);
        }
        p = e ? e + 1 : NULL;
    }
}
void *sam_hdr_parse(size_t l_text, const char *text)
{
    hdr_t *h = calloc(1, sizeof *h);
    h->text = malloc(l_text + 1); memcpy(h->text, text, l_text); h->text[l_text] = 0; h->l_text = l_text;
    hdr_index(h);
    return h;
}
void sam_hdr_destroy(void *hv)
{
    hdr_t *h = hv;
    if (!h) return;
    for (int i = 0; i < h->nref; i++) free(h->names[i]);
    free(h->names); free(h->lens); free(h->text); free(h);
}
void *sam_hdr_read(void *fp)
{
    file_t *f = fp;
    char *text = NULL; size_t l = 0;
    for (;;) {
        const int c = fgetc(f->fp);
        if (c != '@') { if (c != EOF) ungetc(c, f->fp); break; }
        ungetc(c, f->fp);
        const ssize_t n = getline(&f->line, &f->cap, f->fp);
        if (n <= 0) break;
        text = realloc(text, l + (size_t)n + 1); memcpy(text + l, f->line, (size_t)n); l += (size_t)n;
    }
    void *h = sam_hdr_parse(l, text ? text : "");
    free(text);
    return h;
}
int sam_hdr_write(void *fp, const void *hv) { const hdr_t *h = hv; file_t *f = fp; return fwrite(h->text, 1, h->l_text, f->fp) == h->l_text ? 0 : -1; }
int sam_hdr_nref(const void *hv) { return ((const hdr_t *)hv)->nref; }
const char *sam_hdr_tid2name(const void *hv, int tid) { const hdr_t *h = hv; return tid >= 0 && tid < h->nref ? h->names[tid] : NULL; }
int64_t sam_hdr_tid2len(const void *hv, int tid) { const hdr_t *h = hv; return tid >= 0 && tid < h->nref ? h->lens[tid] : 0; }
int sam_hdr_add_pg(void *hv, const char *name, ...)
{
    hdr_t *h = hv;
    char buf[4096];
    int n = snprintf(buf, sizeof buf, "@PG\tID:%s\tPN:%s", name, name);
    va_list ap;
    va_start(ap, name);
    for (;;) {
        const char *k = va_arg(ap, const char *);
        if (!k) break;
        const char *v = va_arg(ap, const char *);
        n += snprintf(buf + n, sizeof buf - (size_t)n, "\t%s:%s", k, v ? v : "");
    }
    va_end(ap);
    n += snprintf(buf + n, sizeof buf - (size_t)n, "\n");
    h->text = realloc(h->text, h->l_text + (size_t)n + 1);
    memcpy(h->text + h->l_text, buf, (size_t)n + 1);
    h->l_text += (size_t)n;
    return 0;
}

bam1_t *bam_init1(void) { return calloc(1, sizeof(bam1_t)); }
void bam_destroy1(bam1_t *b) { if (b) { free(b->data); free(b); } }

static int tid_of(const hdr_t *h, const char *nm)
{
    if (!strcmp(nm, "*")) return -1;
    for (int i = 0; i < h->nref; i++) if (!strcmp(h->names[i], nm)) return i;
    return -2;
}
static void put(bam1_t *b, const void *p, size_t n)
{
    if ((size_t)b->l_data + n > b->m_data) { b->m_data = (uint32_t)(((size_t)b->l_data + n) * 2 + 64); b->data = realloc(b->data, b->m_data); }
    memcpy(b->data + b->l_data, p, n); b->l_data += (int)n;
}
static const char NT16[] = "=ACMGRSVTWYHKDBN";
int sam_parse1(kstring_t *s, void *hv, bam1_t *b)
{
    const hdr_t *h = hv;
    char *line = strndup(s->s, s->l), *f[12], *save = NULL;
    int nf = 0;
    char *rest = NULL;
    for (char *t = strtok_r(line, "\t", &save); t; t = strtok_r(NULL, "\t", &save)) { f[nf++] = t; if (nf == 11) break; }
    rest = save;
    if (nf < 11) { free(line); return -2; }
    bam1_core_t *c = &b->core;
    memset(c, 0, sizeof *c);
    b->l_data = 0;
    c->flag = (uint16_t)atoi(f[1]); c->tid = tid_of(h, f[2]); c->pos = atoll(f[3]) - 1; c->qual = (uint8_t)atoi(f[4]);
    c->mtid = !strcmp(f[6], "=") ? c->tid : tid_of(h, f[6]); c->mpos = atoll(f[7]) - 1; c->isize = atoll(f[8]);
    if (c->tid == -2 || c->mtid == -2) { free(line); return -2; }
    {   /* qname, NUL-padded to a multiple of four */
        const size_t l = strlen(f[0]) + 1, pad = (4 - (l & 3)) & 3;
        put(b, f[0], l);
        put(b, "\0\0\0", pad);
        c->l_qname = (uint16_t)(l + pad); c->l_extranul = (uint8_t)pad;
    }
    if (strcmp(f[5], "*")) {
        for (const char *p = f[5]; *p; ) {
            char *e; const unsigned long n = strtoul(p, &e, 10);
            const char *o = strchr("MIDNSHP=X", *e);
            if (!o || e == p) { free(line); return -2; }
            const uint32_t w = (uint32_t)(n << 4) | (uint32_t)(o - "MIDNSHP=X");
            put(b, &w, 4); c->n_cigar++;
            p = e + 1;
        }
    }
    const int lseq = strcmp(f[9], "*") ? (int)strlen(f[9]) : 0;
    c->l_qseq = lseq;
    for (int i = 0; i < lseq; i += 2) {
        const char *a = strchr(NT16, toupper((unsigned char)f[9][i])), *d = i + 1 < lseq ? strchr(NT16, toupper((unsigned char)f[9][i + 1])) : NT16;
        const uint8_t v = (uint8_t)(((a ? a - NT16 : 15) << 4) | (d ? d - NT16 : 15));
        put(b, &v, 1);
    }
    for (int i = 0; i < lseq; i++) { const uint8_t q = strcmp(f[10], "*") ? (uint8_t)(f[10][i] - 33) : 0xff; put(b, &q, 1); }
    for (char *t = strtok_r(rest, "\t", &save); t; t = strtok_r(NULL, "\t", &save)) {     /* TG:T:value */
        if (strlen(t) < 5) continue;
        put(b, t, 2);
        const char ty = t[3], *v = t + 5;
        if (ty == 'i') {
            const long long x = atoll(v);
            if (x >= 0 && x < 256) { const uint8_t y = (uint8_t)x; put(b, "C", 1); put(b, &y, 1); }
            else if (x >= -128 && x < 0) { const int8_t y = (int8_t)x; put(b, "c", 1); put(b, &y, 1); }
            else if (x >= 0 && x < 65536) { const uint16_t y = (uint16_t)x; put(b, "S", 1); put(b, &y, 2); }
            else if (x >= -32768 && x < 0) { const int16_t y = (int16_t)x; put(b, "s", 1); put(b, &y, 2); }
            else if (x >= 0) { const uint32_t y = (uint32_t)x; put(b, "I", 1); put(b, &y, 4); }
            else { const int32_t y = (int32_t)x; put(b, "i", 1); put(b, &y, 4); }
        } else if (ty == 'A') { put(b, "A", 1); put(b, v, 1); }
        else if (ty == 'f') { const float y = (float)atof(v); put(b, "f", 1); put(b, &y, 4); }
        else { put(b, ty == 'H' ? "H" : "Z", 1); put(b, v, strlen(v) + 1); }       /* (B arrays: kept as text) */
    }
    free(line);
    return 0;
}
int sam_read1(void *fp, void *h, bam1_t *b)
{
    file_t *f = fp;
    ssize_t n = getline(&f->line, &f->cap, f->fp);
    if (n <= 0) return -1;
    while (n > 0 && (f->line[n - 1] == '\n' || f->line[n - 1] == '\r')) n--;
    kstring_t ks = { (size_t)n, f->cap, f->line };
    return sam_parse1(&ks, h, b) < 0 ? -2 : 0;
}
int sam_write1(void *fp, const void *hv, const bam1_t *b)
{
    const hdr_t *h = hv; file_t *f = fp;
    const bam1_core_t *c = &b->core;
    const uint8_t *d = b->data + c->l_qname;
    fprintf(f->fp, "%s\t%u\t%s\t%lld\t%u\t", (const char *)b->data, c->flag, c->tid >= 0 ? h->names[c->tid] : "*", (long long)c->pos + 1, c->qual);
    if (!c->n_cigar) fputc('*', f->fp);
    for (uint32_t i = 0; i < c->n_cigar; i++) { uint32_t w; memcpy(&w, d + 4 * i, 4); fprintf(f->fp, "%u%c", w >> 4, "MIDNSHP=X"[w & 15]); }
    d += 4 * (size_t)c->n_cigar;
    fprintf(f->fp, "\t%s\t%lld\t%lld\t", c->mtid < 0 ? "*" : c->mtid == c->tid ? "=" : h->names[c->mtid], (long long)c->mpos + 1, (long long)c->isize);
    if (!c->l_qseq) fputc('*', f->fp);
    for (int i = 0; i < c->l_qseq; i++) fputc(NT16[(d[i >> 1] >> ((~i & 1) << 2)) & 15], f->fp);
    d += ((size_t)c->l_qseq + 1) / 2;
    fputc('\t', f->fp);
    if (!c->l_qseq || d[0] == 0xff) fputc('*', f->fp);
    else for (int i = 0; i < c->l_qseq; i++) fputc(d[i] + 33, f->fp);
    d += c->l_qseq;
    const uint8_t *lim = b->data + b->l_data;
    while (d + 3 <= lim) {
        fprintf(f->fp, "\t%c%c:", d[0], d[1]);
        const char ty = (char)d[2];
        d += 3;
        if (ty == 'C') { fprintf(f->fp, "i:%u", d[0]); d += 1; }
        else if (ty == 'c') { fprintf(f->fp, "i:%d", (int8_t)d[0]); d += 1; }
        else if (ty == 'S') { uint16_t y; memcpy(&y, d, 2); fprintf(f->fp, "i:%u", y); d += 2; }
        else if (ty == 's') { int16_t y; memcpy(&y, d, 2); fprintf(f->fp, "i:%d", y); d += 2; }
        else if (ty == 'I') { uint32_t y; memcpy(&y, d, 4); fprintf(f->fp, "i:%u", y); d += 4; }
        else if (ty == 'i') { int32_t y; memcpy(&y, d, 4); fprintf(f->fp, "i:%d", y); d += 4; }
        else if (ty == 'A') { fprintf(f->fp, "A:%c", d[0]); d += 1; }
        else if (ty == 'f') { float y; memcpy(&y, d, 4); fprintf(f->fp, "f:%g", y); d += 4; }
        else { fprintf(f->fp, "%c:%s", ty, (const char *)d); d += strlen((const char *)d) + 1; }
    }
    fputc('\n', f->fp);
    return ferror(f->fp) ? -1 : 0;
}
