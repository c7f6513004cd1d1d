/*
 * team.h -- a persistent team of host threads for the BGZF / BAM work (bamio.c) and the host helpers of params.c.
 *
 * Replaces OpenMP there.  crumble-hip runs a reader team and a writer team side by side with the GPU worker threads
 * (read / GPU / write of consecutive contigs overlap): idle threads of one team must not spin on the cores the other
 * team needs.  libgomp decides between spinning and sleeping when it is LOADED (OMP_WAIT_POLICY), which round 4 could
 * only reach by re-executing the process -- an exec that must never happen in a process that may already have touched
 * the GPU (a profiler's preload does).  Here idle workers sleep on a condition variable, always.
 *
 * team_run(t, total, chunk, fn, ctx): fn(ctx, lo, hi, tid) over [0, total) in pieces of `chunk` items handed out by
 * an atomic counter (chunk <= 0: about eight pieces per thread); the caller works as tid 0 and returns when every piece
 * is done.  One team_run per team at a time (each reader / writer owns its team).  Plain C11 + pthreads.
 */
#ifndef CRUMBLE_TEAM_H
#define CRUMBLE_TEAM_H
#include <pthread.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdlib.h>

typedef void (*team_fn)(void *ctx, int64_t lo, int64_t hi, int tid);
typedef struct team {
    int n;                              /* threads including the caller */
    pthread_t *th;
    int started;                        /* workers that exist */
    pthread_mutex_t mu;
    pthread_cond_t cv_go, cv_done;
    uint64_t gen;                       /* bumped per team_run */
    int stop, running;                  /* running: workers still inside the current job */
    team_fn fn; void *ctx;
    int64_t total, chunk;
    atomic_llong next;
    void **slot;                        /* [n] per-thread state of the owner (a codec object), freed by slot_free */
    void (*slot_free)(void *);
} team;
typedef struct { team *t; int tid; } team_arg_;

static void team_work_(team *t, int tid)
{
    const int64_t total = t->total, chunk = t->chunk;
    for (;;) {
        const int64_t lo = (int64_t)atomic_fetch_add(&t->next, (long long)chunk);
        if (lo >= total) break;
        t->fn(t->ctx, lo, lo + chunk < total ? lo + chunk : total, tid);
    }
}

static void *team_main_(void *a_)
{
    team_arg_ *a = (team_arg_ *)a_;
    team *t = a->t; const int tid = a->tid;
    free(a);
    uint64_t seen = 0;
    pthread_mutex_lock(&t->mu);
    for (;;) {
        while (!t->stop && t->gen == seen) pthread_cond_wait(&t->cv_go, &t->mu);
        if (t->stop) break;
        seen = t->gen;
        pthread_mutex_unlock(&t->mu);
        team_work_(t, tid);
        pthread_mutex_lock(&t->mu);
        if (--t->running == 0) pthread_cond_signal(&t->cv_done);
    }
    pthread_mutex_unlock(&t->mu);
    return NULL;
}

static team *team_create(int n)
{
    if (n < 1) n = 1;
    team *t = (team *)calloc(1, sizeof *t);
    if (!t) return NULL;
    t->n = n;
    t->th = (pthread_t *)calloc((size_t)n, sizeof *t->th);
    t->slot = (void **)calloc((size_t)n, sizeof *t->slot);
    if (!t->th || !t->slot) { free(t->th); free(t->slot); free(t); return NULL; }
    pthread_mutex_init(&t->mu, NULL);
    pthread_cond_init(&t->cv_go, NULL); pthread_cond_init(&t->cv_done, NULL);
    atomic_init(&t->next, 0);
    for (int i = 1; i < n; i++) {
        team_arg_ *a = (team_arg_ *)malloc(sizeof *a);
        if (!a) break;
        a->t = t; a->tid = i;
        if (pthread_create(&t->th[i], NULL, team_main_, a) != 0) { free(a); break; }
        t->started++;
    }
    return t;                           /* (fewer workers than asked for still work: the pieces are handed out dynamically) */
}

static void team_destroy(team *t)
{
    if (!t) return;
    pthread_mutex_lock(&t->mu);
    t->stop = 1;
    pthread_cond_broadcast(&t->cv_go);
    pthread_mutex_unlock(&t->mu);
    for (int i = 1; i <= t->started; i++) pthread_join(t->th[i], NULL);
    if (t->slot_free) for (int i = 0; i < t->n; i++) if (t->slot[i]) t->slot_free(t->slot[i]);
    pthread_mutex_destroy(&t->mu); pthread_cond_destroy(&t->cv_go); pthread_cond_destroy(&t->cv_done);
    free(t->th); free(t->slot); free(t);
}

static void team_run(team *t, int64_t total, int64_t chunk, team_fn fn, void *ctx)
{
    if (total <= 0) return;
    if (!t || t->started == 0) { fn(ctx, 0, total, 0); return; }
    if (chunk <= 0) { chunk = total / ((int64_t)(t->started + 1) * 8); if (chunk < 1) chunk = 1; }
    pthread_mutex_lock(&t->mu);
    t->fn = fn; t->ctx = ctx; t->total = total; t->chunk = chunk;
    atomic_store(&t->next, 0);
    t->running = t->started;
    t->gen++;
    pthread_cond_broadcast(&t->cv_go);
    pthread_mutex_unlock(&t->mu);
    team_work_(t, 0);
    pthread_mutex_lock(&t->mu);
    while (t->running) pthread_cond_wait(&t->cv_done, &t->mu);
    pthread_mutex_unlock(&t->mu);
}

/* a one-off fan-out for callers without a team of their own (crumble_pack_bases4, crumble_decode_quals): n threads for
 * one loop; fewer than min_total items are not worth the threads */
__attribute__((unused)) static void team_run_once(int n, int64_t total, int64_t min_total, int64_t chunk, team_fn fn, void *ctx)
{
    if (total <= 0) return;
    if (n <= 1 || total < min_total) { fn(ctx, 0, total, 0); return; }
    team *t = team_create(n);
    team_run(t, total, chunk, fn, ctx);
    team_destroy(t);
}
#endif
