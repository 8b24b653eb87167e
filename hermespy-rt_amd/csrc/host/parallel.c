/* parallel.c -- the parallel-for of the host writers (dense scatter, Q9/Q10 replays, path list):
 * [0, n) cut into one range per thread.  No HIP here: tests/asan builds this file under
 * ThreadSanitizer and AddressSanitizer (tests/test_sanitizers.py). */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "hrt_internal.h"

#include <pthread.h>
#include <sys/types.h>
#include <unistd.h>

/* The dense scatter is host-memory bound and every record owns its slots, so it splits into
 * independent ranges.
 * The helper threads are KEPT, parked on a condition variable, per calling thread (the main thread,
 * or a device worker of HRT_DEVICES: their scatters must not queue behind one another): a warm C3
 * call runs ~25 of these loops, and creating 15 threads for each cost more than some of the loops
 * (the caller creates them one after the other and only then starts its own slice).  A calling
 * thread's helpers end with it (thread-specific destructor); the main thread's stay parked. */
typedef hrt_range_fn range_fn;
typedef struct range_pool range_pool;
typedef struct { range_pool *pool; int idx; uint64_t seen0; } pool_arg;
struct range_pool {
    pthread_mutex_t mu;
    pthread_cond_t cv_start, cv_done;
    int n_workers;                       /* helpers alive (the caller itself is one more) */
    pthread_t th[HRT_MAX_SCATTER_THREADS];
    pool_arg arg[HRT_MAX_SCATTER_THREADS];
    uint64_t gen;                        /* job number: a helper runs each job once */
    int pending, stop;
    range_fn fn; void *ctx; uint64_t n; int parts;
    pid_t owner;                         /* the process that created the helpers: a fork()ed child inherits the
                                          * struct but none of the threads, and must not wait for them */
};
static pthread_key_t g_pool_key;
static pthread_once_t g_pool_once = PTHREAD_ONCE_INIT;

static void *pool_worker(void *a)
{
    pool_arg *pa = (pool_arg *)a;
    range_pool *p = pa->pool;
    uint64_t seen = pa->seen0;   /* the jobs up to the one current at its start are not this helper's */
    pthread_mutex_lock(&p->mu);
    for (;;) {
        while (!p->stop && p->gen == seen) pthread_cond_wait(&p->cv_start, &p->mu);
        if (p->stop) break;
        seen = p->gen;
        if (pa->idx >= p->parts - 1) continue;   /* this job has fewer parts than there are helpers */
        const range_fn fn = p->fn;
        void *ctx = p->ctx;
        const uint64_t n = p->n, parts = (uint64_t)p->parts, t = (uint64_t)pa->idx;
        pthread_mutex_unlock(&p->mu);
        fn(ctx, n * t / parts, n * (t + 1) / parts, (int)t);
        pthread_mutex_lock(&p->mu);
        if (--p->pending == 0) pthread_cond_signal(&p->cv_done);
    }
    pthread_mutex_unlock(&p->mu);
    return NULL;
}

static void pool_destroy(void *vp)
{
    range_pool *p = (range_pool *)vp;
    if (!p) return;
    pthread_mutex_lock(&p->mu);
    p->stop = 1;
    pthread_cond_broadcast(&p->cv_start);
    pthread_mutex_unlock(&p->mu);
    for (int t = 0; t < p->n_workers; ++t) pthread_join(p->th[t], NULL);
    pthread_mutex_destroy(&p->mu);
    pthread_cond_destroy(&p->cv_start);
    pthread_cond_destroy(&p->cv_done);
    free(p);
}
static void pool_key_make(void) { (void)pthread_key_create(&g_pool_key, pool_destroy); }

/* the calling thread's pool with at least `helpers` helpers, or as many as could be started */
static range_pool *pool_get(int helpers)
{
    pthread_once(&g_pool_once, pool_key_make);
    range_pool *p = (range_pool *)pthread_getspecific(g_pool_key);
    if (p && p->owner != getpid()) {
        /* after fork(): the parent's helpers do not exist here.  The inherited pool is dropped without
         * joining (its mutex / condition variables may be in any state: leaked, a few hundred bytes) and
         * a fresh one is made */
        pthread_setspecific(g_pool_key, NULL);
        p = NULL;
    }
    if (!p) {
        p = (range_pool *)calloc(1, sizeof *p);
        if (!p) return NULL;
        p->owner = getpid();
        if (pthread_mutex_init(&p->mu, NULL) != 0) { free(p); return NULL; }
        if (pthread_cond_init(&p->cv_start, NULL) != 0) { pthread_mutex_destroy(&p->mu); free(p); return NULL; }
        if (pthread_cond_init(&p->cv_done, NULL) != 0) {
            pthread_cond_destroy(&p->cv_start); pthread_mutex_destroy(&p->mu); free(p); return NULL;
        }
        if (pthread_setspecific(g_pool_key, p) != 0) { pool_destroy(p); return NULL; }
    }
    while (p->n_workers < helpers && p->n_workers < HRT_MAX_SCATTER_THREADS - 1) {
        const int t = p->n_workers;
        p->arg[t] = (pool_arg){p, t, p->gen};   /* (only the calling thread submits jobs: gen is stable here) */
        if (pthread_create(&p->th[t], NULL, pool_worker, &p->arg[t]) != 0) break;
        ++p->n_workers;
    }
    return p;
}

void hrt_parallel_ranges(hrt_range_fn fn, void *ctx, uint64_t n, int threads)
{
    if (threads > HRT_MAX_SCATTER_THREADS) threads = HRT_MAX_SCATTER_THREADS;
    if ((uint64_t)threads > n / 65536 + 1) threads = (int)(n / 65536 + 1);
    if (threads <= 1) { fn(ctx, 0, n, 0); return; }
    range_pool *p = pool_get(threads - 1);
    const int parts = p ? p->n_workers + 1 < threads ? p->n_workers + 1 : threads : 1;
    if (parts <= 1) { fn(ctx, 0, n, 0); return; }
    pthread_mutex_lock(&p->mu);
    p->fn = fn; p->ctx = ctx; p->n = n; p->parts = parts;
    p->pending = parts - 1;
    ++p->gen;
    pthread_cond_broadcast(&p->cv_start);
    pthread_mutex_unlock(&p->mu);
    /* the caller takes the last part */
    fn(ctx, n * (uint64_t)(parts - 1) / (uint64_t)parts, n, parts - 1);
    pthread_mutex_lock(&p->mu);
    while (p->pending != 0) pthread_cond_wait(&p->cv_done, &p->mu);
    pthread_mutex_unlock(&p->mu);
}

int hrt_host_threads(void)
{
    const char *v = getenv("HRT_HOST_THREADS");
    int t = (v && *v) ? atoi(v) : 0;
    if (t <= 0) {
        long nc = sysconf(_SC_NPROCESSORS_ONLN);
        t = nc > 16 ? 16 : (nc > 0 ? (int)nc : 1);
    }
    /* every per-thread array of the host writers has HRT_MAX_SCATTER_THREADS entries */
    if (t > HRT_MAX_SCATTER_THREADS) t = HRT_MAX_SCATTER_THREADS;
    return t;
}

/* ends the calling thread's parked helpers (hrt_cache_clear) */
void hrt_parallel_release(void)
{
    pthread_once(&g_pool_once, pool_key_make);
    range_pool *p = (range_pool *)pthread_getspecific(g_pool_key);
    if (p) {
        pthread_setspecific(g_pool_key, NULL);
        if (p->owner == getpid()) pool_destroy(p);   /* (a fork()ed child has no helpers to join) */
    }
}
