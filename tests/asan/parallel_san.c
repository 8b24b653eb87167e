/* Sanitizer harness for csrc/host/parallel.c (the parallel-for of the host writers): built once under
 * ThreadSanitizer and once under AddressSanitizer + UBSan by tests/asan/Makefile, run by
 * tests/test_sanitizers.py.  Exercises what the product does: many loops of changing size and thread
 * count from one thread (helpers are started lazily, parked, reused), release + restart, several
 * calling threads at once (the device workers of HRT_DEVICES: one pool each, ended with the thread),
 * and ranges that must partition [0, n) exactly once. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#include "hrt_internal.h"

/* what hrt_internal.h expects from the rest of the library (not used here) */
int hrt_fail(int code, const char *fmt, ...) { (void)fmt; return code; }

typedef struct { unsigned char *hits; unsigned long long sum[HRT_MAX_SCATTER_THREADS]; } ctx_t;
static void mark(void *v, uint64_t i0, uint64_t i1, int tid)
{
    ctx_t *c = (ctx_t *)v;
    unsigned long long s = 0;
    for (uint64_t i = i0; i < i1; ++i) { c->hits[i]++; s += i; }
    c->sum[tid] += s;   /* one writer per tid */
}

static int run_loops(unsigned seed, int loops)
{
    int bad = 0;
    for (int k = 0; k < loops; ++k) {
        seed = seed * 1664525u + 1013904223u;
        const uint64_t n = (seed >> 8) % 1500000u + 1u;
        seed = seed * 1664525u + 1013904223u;
        const int threads = (int)((seed >> 16) % 40u) + 1;   /* beyond HRT_MAX_SCATTER_THREADS too */
        ctx_t c;
        memset(&c, 0, sizeof c);
        c.hits = (unsigned char *)calloc(n, 1);
        if (!c.hits) return 1;
        hrt_parallel_ranges(mark, &c, n, threads);
        unsigned long long s = 0;
        for (int t = 0; t < HRT_MAX_SCATTER_THREADS; ++t) s += c.sum[t];
        for (uint64_t i = 0; i < n; ++i) bad += c.hits[i] != 1;
        bad += s != (unsigned long long)n * (n - 1) / 2;
        free(c.hits);
        if (k % 17 == 16) hrt_parallel_release();
    }
    return bad;
}

static void *worker(void *a)
{
    return (void *)(size_t)run_loops((unsigned)(size_t)a, 25);   /* its pool ends with the thread */
}

int main(void)
{
    int bad = run_loops(1u, 60);
    pthread_t th[4];
    for (int t = 0; t < 4; ++t) pthread_create(&th[t], NULL, worker, (void *)(size_t)(100 + t));
    bad += run_loops(7u, 25);   /* the main thread's pool works beside theirs */
    for (int t = 0; t < 4; ++t) { void *r; pthread_join(th[t], &r); bad += (int)(size_t)r; }
    hrt_parallel_release();
    printf("PARALLEL_%s bad=%d\n", bad ? "FAIL" : "OK", bad);
    return bad != 0;
}
