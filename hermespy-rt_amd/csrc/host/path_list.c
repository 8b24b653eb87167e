/* path_list.c -- hrt_compute_paths_list(): compute_paths() with the result as ONE list of path
 * records instead of the reference's dense [rx][tx][bounce][path] arrays.
 *
 * Same inputs and the same tracing as the drop-in compute_paths (csrc/host/compute_paths.c; the
 * reference's src/compute_paths.c:419-757), same values bit for bit; what differs is the output
 * side: the dense form is > 95 % unwritten slots (C3: 2.3 GB of caller arrays for 23 M non-zero
 * records, most of a warm call's time goes into page-faulting them in), the list holds exactly the
 * records, each with the indices of the dense slot it would occupy.  Built on the public
 * device-resident API (include/hrt_device.h) only.
 */
#define _GNU_SOURCE
#include <malloc.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>

#include "hrt_internal.h"

/* (the slim path forms tau and direction_rx here: nothing in this file may be contracted into an FMA) */
#if defined(__clang__)
#pragma STDC FP_CONTRACT OFF
#elif defined(__GNUC__)
#pragma GCC optimize("fp-contract=off")
#endif
#if defined(__FAST_MATH__)
#error "path_list.c must not be built with -ffast-math"
#endif

static uint64_t pl_env_u64(const char *name, uint64_t dflt)
{
    const char *v = getenv(name);
    return (v && *v) ? (uint64_t)strtoull(v, NULL, 10) : dflt;
}

/* The blocks of the last freed list are kept for the next one (one set, at most HRT_LIST_KEEP_MAX bytes;
 * hrt_cache_clear() releases them, HRT_NO_CACHE=1 disables): a list is written once, front to back, and on fresh
 * memory most of that time is page faults -- 1.5 GB for C3's 23 M records. */
#define HRT_LIST_BLOCKS 14
#define HRT_LIST_KEEP_MAX (4ull << 30)
static pthread_mutex_t g_list_lock = PTHREAD_MUTEX_INITIALIZER;
static void *g_list_blk[HRT_LIST_BLOCKS];      /* in the order of list_blocks() */
static size_t g_list_bytes[HRT_LIST_BLOCKS];

static void list_blocks(hrt_path_list *pl, void ***b)
{
    b[0] = (void **)&pl->rx; b[1] = (void **)&pl->tx; b[2] = (void **)&pl->bounce; b[3] = (void **)&pl->path;
    b[4] = (void **)&pl->a_te_re; b[5] = (void **)&pl->a_te_im; b[6] = (void **)&pl->a_tm_re; b[7] = (void **)&pl->a_tm_im;
    b[8] = (void **)&pl->tau; b[9] = (void **)&pl->direction_rx; b[10] = (void **)&pl->freq_shift;
    b[11] = (void **)&pl->unblocked; b[12] = (void **)&pl->mesh; b[13] = (void **)&pl->face;
}

void hrt_list_cache_clear(void)
{
    pthread_mutex_lock(&g_list_lock);
    for (int k = 0; k < HRT_LIST_BLOCKS; ++k) { free(g_list_blk[k]); g_list_blk[k] = NULL; g_list_bytes[k] = 0; }
    pthread_mutex_unlock(&g_list_lock);
}

/* a kept block of field k with at least `bytes`, or NULL */
static void *list_cache_take(int k, size_t bytes, size_t *got)
{
    void *p = NULL;
    pthread_mutex_lock(&g_list_lock);
    if (g_list_blk[k] && g_list_bytes[k] >= bytes) {
        p = g_list_blk[k]; *got = g_list_bytes[k];
        g_list_blk[k] = NULL; g_list_bytes[k] = 0;
    }
    pthread_mutex_unlock(&g_list_lock);
    return p;
}

void hrt_path_list_free(hrt_path_list *pl)
{
    if (!pl) return;
    void **b[HRT_LIST_BLOCKS];
    list_blocks(pl, b);
    const char *nc = getenv("HRT_NO_CACHE");
    int keep = !(nc && *nc && *nc != '0') && pl->rx != NULL;
    size_t bytes[HRT_LIST_BLOCKS], total = 0;
    for (int k = 0; k < HRT_LIST_BLOCKS && keep; ++k) {
        bytes[k] = *b[k] ? malloc_usable_size(*b[k]) : 0;
        total += bytes[k];
        if (!*b[k]) keep = 0;
    }
    if (keep && total <= HRT_LIST_KEEP_MAX) {
        pthread_mutex_lock(&g_list_lock);
        for (int k = 0; k < HRT_LIST_BLOCKS; ++k) {
            if (g_list_bytes[k] >= bytes[k]) { free(*b[k]); continue; }   /* the kept one is at least as big */
            free(g_list_blk[k]);
            g_list_blk[k] = *b[k]; g_list_bytes[k] = bytes[k];
        }
        pthread_mutex_unlock(&g_list_lock);
    } else {
        for (int k = 0; k < HRT_LIST_BLOCKS; ++k) free(*b[k]);
    }
    free(pl->los);
    memset(pl, 0, sizeof *pl);
}

static int pl_reserve(hrt_path_list *pl, uint64_t *cap, uint64_t need)
{
    if (need <= *cap) return HRT_OK;
    uint64_t nc = *cap ? *cap : 1024;
    while (nc < need) nc += nc / 2 + 1024;
    /* 2 MiB-aligned blocks advised to use huge pages: the list is written once, front to back, by
     * several threads, and with 4 KiB pages most of that time is page faults (free() releases
     * posix_memalign memory; the advice is only a hint) */
#define GROW(field, type)                                                        \
    do {                                                                         \
        void *q_ = NULL;                                                         \
        size_t got_ = 0;                                                         \
        const size_t bytes_ = (((size_t)nc * sizeof(type)) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1); \
        q_ = list_cache_take(blk_++, bytes_, &got_);   /* (a block kept from the last list: already paged in) */ \
        if (!q_) {                                                               \
            if (posix_memalign(&q_, (size_t)2 << 20, bytes_) != 0 || !q_)       \
                return hrt_fail(HRT_E_NOMEM, "out of host memory");              \
            (void)madvise(q_, bytes_, MADV_HUGEPAGE);                            \
        }                                                                        \
        if (pl->field) {                                                         \
            memcpy(q_, pl->field, (size_t)pl->num * sizeof(type));               \
            free(pl->field);                                                     \
        }                                                                        \
        pl->field = (type *)q_;                                                  \
    } while (0)
    int blk_ = 0;   /* (the order of list_blocks()) */
    GROW(rx, uint32_t); GROW(tx, uint32_t); GROW(bounce, uint32_t); GROW(path, uint64_t);
    GROW(a_te_re, float); GROW(a_te_im, float); GROW(a_tm_re, float); GROW(a_tm_im, float);
    GROW(tau, float); GROW(direction_rx, Vec3); GROW(freq_shift, float); GROW(unblocked, uint8_t);
    GROW(mesh, uint32_t); GROW(face, uint32_t);
#undef GROW
    *cap = nc;
    return HRT_OK;
}

/* one (bounce, rx) block of H records -> entries [base + offset of the range ...) of the list */
typedef struct {
    hrt_path_list *out;
    const hrt_shard *s;
    const hrt_problem *prob;
    const uint32_t *ray, *tri;
    const float *fs0;
    float *const *field;
    const uint64_t *mask;
    uint64_t n_loc, base;
    uint32_t rx, bounce;
    int include_blocked;
    /* slim records (default, as in compute_paths.c: HRT_FULL_RECORDS=1 copies all nine fields): tau and
     * direction_rx are not copied (16 of a record's 36 bytes) but formed here from the hit's origin and delay
     * (16 bytes per HIT) with the reference's float sequence (src/compute_paths.c:676-678, :707-709) */
    const float *hs[4];   /* o.x o.y o.z tau of the hits; NULL: the records carry the fields */
    float rxp[3];
    uint64_t start[HRT_MAX_SCATTER_THREADS + 1];   /* first output entry of every range */
    uint64_t i0[HRT_MAX_SCATTER_THREADS], i1[HRT_MAX_SCATTER_THREADS];
} fill_ctx;

static uint64_t count_bits(const uint64_t *mask, uint64_t i0, uint64_t i1)
{
    uint64_t n = 0;
    for (uint64_t i = i0; i < i1;) {
        if ((i & 63) == 0 && i + 64 <= i1) { n += (uint64_t)__builtin_popcountll(mask[i >> 6]); i += 64; }
        else { n += (mask[i >> 6] >> (i & 63)) & 1u; ++i; }
    }
    return n;
}

static void fill_one(fill_ctx *c, int tid)
{
    hrt_path_list *out = c->out;
    uint64_t n = c->base + c->start[tid];
    for (uint64_t i = c->i0[tid]; i < c->i1[tid]; ++i) {
        const int ub = (int)((c->mask[i >> 6] >> (i & 63)) & 1u);
        if (!ub && !c->include_blocked) continue;
        const uint32_t ql = c->ray[i];
        const uint32_t tx = (uint32_t)(ql / c->n_loc);
        out->rx[n] = c->rx;
        out->tx[n] = tx;
        out->bounce[n] = c->bounce;
        out->path[n] = hrt_shard_global_path(c->s, ql - (uint64_t)tx * c->n_loc);
        out->a_te_re[n] = c->field[HRT_REC_A_TE_RE][i];
        out->a_te_im[n] = c->field[HRT_REC_A_TE_IM][i];
        out->a_tm_re[n] = c->field[HRT_REC_A_TM_RE][i];
        out->a_tm_im[n] = c->field[HRT_REC_A_TM_IM][i];
        if (c->hs[0]) {
            if (ub) {
                const float wx = c->rxp[0] - c->hs[0][i], wy = c->rxp[1] - c->hs[1][i], wz = c->rxp[2] - c->hs[2][i];
                const float d2rx = sqrtf((wx * wx + wy * wy) + wz * wz);
                const float ux = wx / d2rx, uy = wy / d2rx, uz = wz / d2rx;
                out->tau[n] = c->hs[3][i] + d2rx / HRT_C_F;
                out->direction_rx[n] = (Vec3){-ux, -uy, -uz};
            } else {
                out->tau[n] = 0.f;   /* :688 */
                out->direction_rx[n] = (Vec3){0.f, 0.f, 0.f};   /* (not written by the reference: Q2) */
            }
        } else {
            out->tau[n] = c->field[HRT_REC_TAU][i];
            out->direction_rx[n] = (Vec3){c->field[HRT_REC_DIRX][i], c->field[HRT_REC_DIRY][i],
                                          c->field[HRT_REC_DIRZ][i]};
        }
        out->freq_shift[n] = c->fs0[i] - c->field[HRT_REC_DFS][i];
        out->unblocked[n] = (uint8_t)ub;
        out->mesh[n] = c->prob->h_tri_mesh[c->tri[i]];
        out->face[n] = c->prob->h_tri_face[c->tri[i]];
        ++n;
    }
}

/* hrt_parallel_ranges splits [0, nt * 65536): slice k of it stands for prepared range k (a thread
 * that could not be started leaves several slices to one caller) */
static void fill_range(void *vctx, uint64_t i0, uint64_t i1, int tid)
{
    (void)tid;
    for (uint64_t k = i0 / 65536; k < i1 / 65536; ++k) fill_one((fill_ctx *)vctx, (int)k);
}

int hrt_compute_paths_list(Scene *scene, const Vec3 *rx_pos, const Vec3 *tx_pos, const Vec3 *rx_vel,
                           const Vec3 *tx_vel, float f_ghz, size_t nrx, size_t ntx, size_t np,
                           size_t nb, int include_blocked, hrt_path_list *out, hrt_stats *stats)
{
    const double t_begin = hrt_now_s();
    if (!scene || !out) return hrt_fail(HRT_E_INVALID, "hrt_compute_paths_list: NULL argument");
    if (np == 0 || nb == 0) return hrt_fail(HRT_E_INVALID, "num_rays and num_bounces must be > 0");
    if (nb > 65535) return hrt_fail(HRT_E_INVALID, "num_bounces > 65535 is not supported");
    memset(out, 0, sizeof *out);

    const int device = (int)pl_env_u64("HRT_DEVICE", 0);
    hrt_stats st;
    memset(&st, 0, sizeof st);
    st.device = device;
    hrt_problem *prob = NULL;
    int rc = hrt_problem_create_for(scene, rx_pos, tx_pos, rx_vel, tx_vel, f_ghz, nrx, ntx, device,
                                    (uint64_t)ntx * np, &prob);
    if (rc) return rc;
    st.t_setup_s = hrt_now_s() - t_begin;

    dev_ctx wc;                      /* the drop-in's worker buffers: workspace + staging, pooled between calls */
    memset(&wc, 0, sizeof wc);
    int pool_taken = 0, have_buffers = 0;
    void *d_ws = NULL, *d_dirs = NULL, *d_order = NULL;
    float *h_dirs = NULL, **h_field = NULL, **h_field2 = NULL, *h_fs0 = NULL, *h_fs02 = NULL;
    uint64_t *h_mask = NULL, *h_mask2 = NULL;
    void *copy_stream = NULL, *copy_stream2 = NULL;
    const int host_launch = (int)pl_env_u64("HRT_HOST_LAUNCH", 0);
    uint32_t *h_order = NULL, *h_ray = NULL, *h_tri = NULL, *h_ray2 = NULL, *h_tri2 = NULL, *h_counts = NULL;
    float *h_hs[4] = {NULL, NULL, NULL, NULL}, *h_hs2[4] = {NULL, NULL, NULL, NULL};
    int slim = 0;
    uint64_t cap_out = 0;
    const int threads = hrt_host_threads();
    double t_dev = 0.0, t_rb = 0.0, t_dirs = 0.0;

    /* batches of round-robin shards so that one workspace fits the budget (as the drop-in does) */
    uint64_t free_b = 0, total_b = 0;
    if ((rc = hrt_device_mem_info(device, &free_b, &total_b))) goto done;
    uint64_t budget = pl_env_u64("HRT_WORKSPACE_BYTES", 0);
    if (!budget) {
        budget = free_b / 2;
        if (budget > (16ull << 30)) budget = 16ull << 30;
    }
    uint32_t G = 1, G_budget = 0;
    hrt_layout L;
    for (;;) {
        hrt_shard s = {np, 0, G, 0, (uint32_t)nb};
        rc = hrt_layout_query(prob, &s, &L);
        const int fits = rc == HRT_OK && L.total_bytes + hrt_shard_num_local(&s) * 16 <= budget;
        if (fits && !G_budget) G_budget = G;
        /* (the pool rule for calls of one or two budget-sized batches: compute_paths.c) */
        if (fits && (G_budget > 2u || hrt_batch_fits_pool(L.total_bytes, hrt_shard_num_local(&s) + 64, L.cap))) break;
        if (rc != HRT_OK && rc != HRT_E_CAPACITY) goto done;
        if ((uint64_t)G * 4096 >= np) {
            if (rc == HRT_OK) break;
            goto done;
        }
        G *= 2;
    }
    {
        wc.prob = prob; wc.nrx = nrx; wc.ntx = ntx; wc.np = np; wc.nb = nb; wc.G = G; wc.index = 0; wc.count = 1;
        wc.device = device;
        pool_taken = hrt_pool_begin();
        wc.use_pool = pool_taken;
        if ((rc = hrt_worker_alloc(&wc))) { wc.rc = rc; goto done; }
        have_buffers = 1;
        d_ws = wc.w.d_ws; d_dirs = wc.w.d_dirs; d_order = wc.w.d_order;
        h_order = wc.w.h_order; h_counts = wc.w.h_counts;
        h_ray = wc.w.ray; h_tri = wc.w.tri; h_fs0 = wc.w.fs0;
        h_ray2 = wc.w.ray2; h_tri2 = wc.w.tri2; h_fs02 = wc.w.fs02;
        h_field = wc.w.rec; h_field2 = wc.w.rec2; h_mask = wc.w.mask; h_mask2 = wc.w.mask2;
        slim = wc.w.hs[0] != NULL && wc.w.hs2[0] != NULL;   /* (allocated unless HRT_FULL_RECORDS=1) */
        for (int k = 0; k < 4; ++k) { h_hs[k] = wc.w.hs[k]; h_hs2[k] = wc.w.hs2[k]; }
        copy_stream = wc.w.copy_stream;
        copy_stream2 = wc.w.copy_stream2;
        if (host_launch) {
            h_dirs = (float *)malloc(hrt_shard_num_local(&(hrt_shard){np, 0, G, 0, (uint32_t)nb}) * 12);
            if (!h_dirs) { rc = hrt_fail(HRT_E_NOMEM, "out of host memory"); goto done; }
        }
        out->los = (float *)malloc(nrx * ntx * HRT_LOS_FLOATS * sizeof(float));
        if (!out->los) { rc = hrt_fail(HRT_E_NOMEM, "out of host memory"); goto done; }
    }
    out->num_rx = (uint32_t)nrx;
    out->num_tx = (uint32_t)ntx;

#define DLP(dst, off, bytes)                                                                    \
    do {                                                                                        \
        if ((rc = hrt_device_download(device, (dst), (const uint8_t *)d_ws + (off), (bytes)))) goto done; \
    } while (0)

    for (uint32_t g = 0; g < G; ++g) {
        hrt_shard s = {np, g, G, 0, (uint32_t)nb};
        const uint64_t n_loc = hrt_shard_num_local(&s);
        if (n_loc == 0) continue;
        if ((rc = hrt_layout_query(prob, &s, &L))) goto done;
        double t0 = hrt_now_s();
        if (host_launch) {
            if (!(G == 1 && hrt_launch_cache_enabled(np) && hrt_launch_cache_get(np, h_dirs, h_order))) {
                if ((rc = hrt_launch_dirs_host(&s, h_dirs, 0))) goto done;
                if ((rc = hrt_launch_order_host(&s, h_dirs, h_order))) goto done;
                if (G == 1) hrt_launch_cache_put(np, h_dirs, h_order);
            }
            t_dirs += hrt_now_s() - t0;
            t0 = hrt_now_s();
            if ((rc = hrt_device_upload(device, d_dirs, h_dirs, n_loc * 12))) goto done;
            if ((rc = hrt_device_upload(device, d_order, h_order, n_loc * 4))) goto done;
        } else {   /* launch tables on the device (the default, as in compute_paths) */
            if ((rc = hrt_launch_dirs_device(&s, (float *)d_dirs, device, NULL, NULL))) goto done;
            if ((rc = hrt_launch_order_device(&s, (uint32_t *)d_order, device, NULL))) goto done;
            t_dirs += hrt_now_s() - t0;
            t0 = hrt_now_s();
        }
        for (int attempt = 0;; ++attempt) {
            if ((rc = hrt_trace(prob, &s, (const float *)d_dirs, (const uint32_t *)d_order, d_ws,
                                L.total_bytes, NULL, NULL))) goto done;
            if ((rc = hrt_device_sync(device, NULL))) goto done;
            DLP(h_counts, L.off_counts, (nb + 2) * 4);
            /* (a fused launch timed out on a shared GPU: the step is void, once more unfused -- compute_paths.c) */
            if (!(h_counts[nb + 1] & HRT_ERR_VOID) || attempt >= 2 || !hrt_void_step_retry(h_counts[nb + 1])) break;
        }
        t_dev += hrt_now_s() - t0;

        t0 = hrt_now_s();
        if (h_counts[nb + 1] != 0) {
            rc = hrt_fail(HRT_E_HIP, "device reported internal error flags %u", h_counts[nb + 1]);
            goto done;
        }
        {
            hrt_stats bs;
            hrt_work_from_counts(prob, &s, h_counts, &bs);
            for (size_t b = 0; b <= nb && b < 34; ++b) st.live[b] += bs.live[b];
            st.records += bs.records;
            st.tests += bs.tests - (g ? (uint64_t)nrx * ntx * prob->num_tri : 0);
        }
        if (g == 0) DLP(out->los, L.off_los, nrx * ntx * HRT_LOS_FLOATS * sizeof(float));
        {   /* room for every record of this batch (exact with include_blocked, <= 1.2x otherwise).  With several
             * batches the first one sizes the whole list: the batches are round-robin shards of one launch set, so
             * each holds 1 / G of the records within a fraction of a percent -- growing the list batch by batch
             * copied it G times (C5, 16 batches, 45 GB of list: 12 s of the call) */
            uint64_t recs = 0;
            for (size_t b = 0; b < nb; ++b) recs += (uint64_t)nrx * h_counts[b + 1];
            uint64_t want = out->num + recs;
            if (g == 0 && G > 1) want = recs * G + recs * G / 32 + 65536;
            if (want < out->num + recs) want = out->num + recs;
            if ((rc = pl_reserve(out, &cap_out, want))) goto done;
        }
        int pre = 0, flip = 0;   /* staging set of a block: (rx + flip) & 1 */
        const int can_pre = !prob->tune.no_bounce_prefetch;
        for (size_t b = 0; b < nb; ++b) {
            const uint64_t H = h_counts[b + 1];
            if (!H) continue;
            /* Per-hit data (ray, triangle, launch Doppler term) and record blocks go out on the copy
             * stream, a block ahead of the one being written into the list; the NEXT bounce's per-hit data
             * and first block are requested during the last block of this one (`pre`), into the second
             * pair of arrays / the other staging set. */
#define FETCH_HITS(B, HN, RAY, TRI, FS0, HS)                                                      \
    do {                                                                                          \
        static const int hs_field_[4] = {HRT_HIT_OX, HRT_HIT_OY, HRT_HIT_OZ, HRT_HIT_TAU};        \
        const uint64_t hb_ = L.off_hits + (uint64_t)(B) * L.hit_block_bytes;                      \
        int e_ = hrt_hip_d2h_async((RAY), (const uint8_t *)d_ws + hb_ + (uint64_t)HRT_HIT_RAY * L.cap * 4, (HN) * 4, copy_stream); \
        if (!e_) e_ = hrt_hip_d2h_async((TRI), (const uint8_t *)d_ws + hb_ + (uint64_t)HRT_HIT_TRI * L.cap * 4, (HN) * 4, copy_stream2); \
        if (!e_) e_ = hrt_hip_d2h_async((FS0), (const uint8_t *)d_ws + hb_ + (uint64_t)HRT_HIT_FS0 * L.cap * 4, (HN) * 4, copy_stream); \
        for (int q_ = 0; q_ < 4 && slim && !e_; ++q_)                                             \
            e_ = hrt_hip_d2h_async((HS)[q_], (const uint8_t *)d_ws + hb_ + (uint64_t)hs_field_[q_] * L.cap * 4, (HN) * 4, \
                                   (q_ & 1) ? copy_stream : copy_stream2);                        \
        if (e_) { rc = hrt_fail(HRT_E_HIP, "hipMemcpyAsync D2H failed (%d)", e_); goto done; }    \
    } while (0)
#define FETCH_PL(B, HN, RX, SET, MASK)                                                            \
    do {                                                                                          \
        const uint64_t rb_ = L.off_recs + (uint64_t)(B) * L.rec_block_bytes + (uint64_t)(RX) * HRT_REC_FIELDS * L.cap * 4; \
        int e_ = 0;                                                                               \
        for (int k = 0; k < HRT_REC_FIELDS && !e_; ++k)                                           \
            if (!(slim && k >= HRT_REC_TAU && k <= HRT_REC_DIRZ))   /* formed on the host from the hits */ \
            e_ = hrt_hip_d2h_async((SET)[k], (const uint8_t *)d_ws + rb_ + (uint64_t)k * L.cap * 4, (HN) * 4,         \
                                   (k & 1) ? copy_stream2 : copy_stream);                         \
        if (!e_) e_ = hrt_hip_d2h_async((MASK), (const uint8_t *)d_ws + L.off_masks + ((uint64_t)(B) * nrx + (RX)) * (L.cap / 64) * 8, \
                                        ((HN) + 63) / 64 * 8, copy_stream);                       \
        if (e_) { rc = hrt_fail(HRT_E_HIP, "hipMemcpyAsync D2H failed (%d)", e_); goto done; }    \
    } while (0)
            if (pre) {   /* requested during the previous bounce */
                uint32_t *t_ = h_ray; h_ray = h_ray2; h_ray2 = t_;
                t_ = h_tri; h_tri = h_tri2; h_tri2 = t_;
                float *f_ = h_fs0; h_fs0 = h_fs02; h_fs02 = f_;
                for (int k = 0; k < 4; ++k) { f_ = h_hs[k]; h_hs[k] = h_hs2[k]; h_hs2[k] = f_; }
                pre = 0;
            } else {
                FETCH_HITS(b, H, h_ray, h_tri, h_fs0, h_hs);
                if (flip & 1) FETCH_PL(b, H, 0, h_field2, h_mask2);
                else FETCH_PL(b, H, 0, h_field, h_mask);
            }
            for (size_t rx = 0; rx < nrx; ++rx) {
                const size_t slot = rx + (size_t)flip;
                float *const *cur_field = (slot & 1) ? h_field2 : h_field;
                const uint64_t *cur_mask = (slot & 1) ? h_mask2 : h_mask;
                {
                    int e = hrt_hip_stream_sync(copy_stream);   /* block rx has landed */
                    if (!e) e = hrt_hip_stream_sync(copy_stream2);
                    if (e) { rc = hrt_fail(HRT_E_HIP, "hipStreamSynchronize failed (%d)", e); goto done; }
                }
                if (rx + 1 < nrx) {   /* the copy of the next block runs while this one is written out */
                    if (slot & 1) FETCH_PL(b, H, rx + 1, h_field, h_mask);
                    else FETCH_PL(b, H, rx + 1, h_field2, h_mask2);
                } else if (can_pre && b + 1 < nb && h_counts[b + 2] != 0) {   /* ... or the start of the next bounce */
                    const uint64_t Hn = h_counts[b + 2];
                    FETCH_HITS(b + 1, Hn, h_ray2, h_tri2, h_fs02, h_hs2);
                    if (slot & 1) FETCH_PL(b + 1, Hn, 0, h_field, h_mask);
                    else FETCH_PL(b + 1, Hn, 0, h_field2, h_mask2);
                    pre = 1;
                    flip = (int)((slot + 1) & 1);
                }
                {
                    fill_ctx fc;
                    memset(&fc, 0, sizeof fc);
                    fc.out = out; fc.s = &s; fc.prob = prob; fc.ray = h_ray; fc.tri = h_tri; fc.fs0 = h_fs0;
                    fc.field = cur_field; fc.mask = cur_mask; fc.n_loc = n_loc; fc.base = out->num;
                    fc.rx = (uint32_t)rx; fc.bounce = (uint32_t)b; fc.include_blocked = include_blocked;
                    if (slim) {
                        for (int k = 0; k < 4; ++k) fc.hs[k] = h_hs[k];
                        fc.rxp[0] = rx_pos[rx].x; fc.rxp[1] = rx_pos[rx].y; fc.rxp[2] = rx_pos[rx].z;
                    }
                    int nt = threads;
                    if ((uint64_t)nt > H / 65536 + 1) nt = (int)(H / 65536 + 1);
                    uint64_t total = 0, unb = 0;
                    for (int t = 0; t < nt; ++t) {
                        /* ranges on 64-entry boundaries, so that no mask word is shared */
                        fc.i0[t] = (H * (uint64_t)t / (uint64_t)nt) & ~63ull;
                        fc.i1[t] = (t + 1 == nt) ? H : ((H * (uint64_t)(t + 1) / (uint64_t)nt) & ~63ull);
                        const uint64_t u = count_bits(cur_mask, fc.i0[t], fc.i1[t]);
                        fc.start[t] = total;
                        total += include_blocked ? fc.i1[t] - fc.i0[t] : u;
                        unb += u;
                    }
                    st.records_unblocked += unb;
                    /* one range per thread: hrt_parallel_ranges(n = nt, threads = nt) */
                    hrt_parallel_ranges(fill_range, &fc, (uint64_t)nt * 65536, nt);
                    out->num += total;
                }
            }
        }
        t_rb += hrt_now_s() - t0;
    }
#undef FETCH_PL
#undef FETCH_HITS
#undef DLP
    st.num_batches = G;
    st.t_launch_dirs_s = t_dirs;
    st.t_device_s = t_dev;
    st.t_readback_s = t_rb;
    st.t_total_s = hrt_now_s() - t_begin;
    if (stats) *stats = st;
    rc = HRT_OK;

done:
    if (have_buffers || wc.w.d_ws || wc.w.ray) {
        wc.rc = rc;
        if (copy_stream) hrt_hip_stream_sync(copy_stream);
        if (copy_stream2) hrt_hip_stream_sync(copy_stream2);
        hrt_worker_release(&wc);
    }
    hrt_pool_end(pool_taken);
    free(h_dirs);
    hrt_problem_destroy(prob);
    if (rc != HRT_OK) hrt_path_list_free(out);
    return rc;
}
