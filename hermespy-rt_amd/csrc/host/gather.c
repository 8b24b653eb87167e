/* gather.c -- the exchange step of the sharded path: every rank packs its compact
 * result into one run of 32-bit words in HBM, and the runs are gathered to one rank.
 *
 * Two layers (include/hrt_device.h, "packed export and gather"):
 *   * hrt_gather_prepare / _set_meta / _pack and the hrt_export_* layout functions are transport
 *     agnostic: a caller with its own transport (hermespy-rt_amd/sharding.py: torch.distributed, gloo
 *     or RCCL) exchanges the small meta block and the packed runs itself;
 *   * hrt_gather_rccl does it all over RCCL for a C / C++ consumer with one process per GPU: an
 *     ncclAllGather of the meta blocks, then ONE group of ncclSend / ncclRecv -- every peer -> root transfer
 *     has its own xGMI link, so the gather is point-to-point, not a ring.  librccl is bound at run time
 *     (dlopen: the copy already in the process -- torch's -- if there is one), so the library has no link
 *     dependency on it.
 * The reference has no counterpart (it is single-process). */
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include "hrt_internal.h"

#define N_HIT_ROWS 4u   /* ray, tri, theta, fs0: the first four fields of a hit block */

struct hrt_gather {
    const hrt_problem *p;
    hrt_shard s;
    hrt_layout L;
    int root;
    uint32_t flags, nb, nrx, meta_words;
    /* device */
    uint32_t *d_meta;       /* [meta_words]: counts[nb + 2] | unblocked[nb * nrx] */
    uint32_t *d_meta_all;   /* [world][meta_words] (hrt_gather_rccl) */
    uint32_t *d_prefix;     /* [nb * nrx][cap / 64] */
    void *d_segs;           /* segment table of the pack */
    uint64_t *d_dst;        /* [nb * nrx] export offsets of the compacted records */
    uint32_t *d_pack;       /* this rank's export */
    uint64_t pack_cap;      /* words */
    uint32_t **d_recv;      /* root: [world] receive buffers */
    uint64_t *recv_cap;
    /* host */
    uint32_t *h_meta_all;   /* [world][meta_words] */
    uint8_t *have;          /* [world] meta of rank r is known */
    void *h_segs;
    uint32_t segs_cap;
};

/* ---- layout ---- */
uint32_t hrt_export_meta_words(uint32_t num_bounces, uint32_t num_rx)
{
    return num_bounces + 2u + num_bounces * num_rx;
}

static uint64_t pair_words(uint64_t H, uint64_t U, uint32_t flags)
{
    if (flags & HRT_EXPORT_UNBLOCKED) return U * (1u + HRT_REC_FIELDS);
    return HRT_REC_FIELDS * H + 2u * ((H + 63u) / 64u);
}

uint64_t hrt_export_words(const uint32_t *meta, uint32_t num_bounces, uint32_t num_rx, uint32_t flags)
{
    uint64_t n = 0;
    for (uint32_t b = 0; b < num_bounces; ++b) {
        const uint64_t H = meta[b + 1];
        n += N_HIT_ROWS * H;
        for (uint32_t rx = 0; rx < num_rx; ++rx) n += pair_words(H, meta[num_bounces + 2u + b * num_rx + rx], flags);
    }
    return n;
}

int hrt_export_locate(const uint32_t *meta, uint32_t num_bounces, uint32_t num_rx, uint32_t flags, uint32_t bounce,
                      uint32_t rx, hrt_export_part *out)
{
    if (!meta || !out || bounce >= num_bounces || rx >= num_rx) return hrt_fail(HRT_E_INVALID, "hrt_export_locate: bad argument");
    uint64_t n = 0;
    for (uint32_t b = 0; b < bounce; ++b) {
        const uint64_t H = meta[b + 1];
        n += N_HIT_ROWS * H;
        for (uint32_t r = 0; r < num_rx; ++r) n += pair_words(H, meta[num_bounces + 2u + b * num_rx + r], flags);
    }
    const uint64_t H = meta[bounce + 1];
    memset(out, 0, sizeof *out);
    out->hits = H;
    out->off_hit = n;
    n += N_HIT_ROWS * H;
    for (uint32_t r = 0; r < rx; ++r) n += pair_words(H, meta[num_bounces + 2u + bounce * num_rx + r], flags);
    const uint64_t U = meta[num_bounces + 2u + bounce * num_rx + rx];
    out->unblocked = U;
    if (flags & HRT_EXPORT_UNBLOCKED) {
        out->records = U;
        out->off_index = n;
        out->off_rec = n + U;
        out->off_mask = 0;
    } else {
        out->records = H;
        out->off_index = 0;
        out->off_rec = n;
        out->off_mask = n + HRT_REC_FIELDS * H;
    }
    return HRT_OK;
}

/* ---- object ---- */
void hrt_gather_destroy(hrt_gather *g)
{
    if (!g) return;
    hrt_hip_set_device(g->p->device);
    if (g->d_meta) hrt_hip_free(g->d_meta);
    if (g->d_meta_all) hrt_hip_free(g->d_meta_all);
    if (g->d_prefix) hrt_hip_free(g->d_prefix);
    if (g->d_segs) hrt_hip_free(g->d_segs);
    if (g->d_dst) hrt_hip_free(g->d_dst);
    if (g->d_pack) hrt_hip_free(g->d_pack);
    if (g->d_recv)
        for (uint32_t r = 0; r < g->s.count; ++r) if (g->d_recv[r]) hrt_hip_free(g->d_recv[r]);
    free(g->d_recv); free(g->recv_cap); free(g->h_meta_all); free(g->have); free(g->h_segs);
    free(g);
}

int hrt_gather_create(const hrt_problem *p, const hrt_shard *s, int root, uint32_t flags, hrt_gather **out)
{
    if (!p || !s || !out || s->count == 0 || s->rank >= s->count || root < 0 || (uint32_t)root >= s->count)
        return hrt_fail(HRT_E_INVALID, "hrt_gather_create: bad argument");
    hrt_gather *g = (hrt_gather *)calloc(1, sizeof *g);
    if (!g) return hrt_fail(HRT_E_NOMEM, "out of host memory");
    g->p = p; g->s = *s; g->root = root; g->flags = flags;
    int rc = hrt_layout_query(p, s, &g->L);
    if (rc) { free(g); return rc; }
    g->nb = s->num_bounces; g->nrx = p->num_rx;
    g->meta_words = hrt_export_meta_words(g->nb, g->nrx);
    const uint32_t world = s->count;
    g->h_meta_all = (uint32_t *)calloc((size_t)world * g->meta_words, 4);
    g->have = (uint8_t *)calloc(world, 1);
    g->d_recv = (uint32_t **)calloc(world, sizeof(uint32_t *));
    g->recv_cap = (uint64_t *)calloc(world, sizeof(uint64_t));
    g->segs_cap = g->nb * (N_HIT_ROWS + g->nrx * (HRT_REC_FIELDS + 1u)) + 1u;
    g->h_segs = calloc(g->segs_cap, 24);
    int e = hrt_hip_set_device(p->device);
    if (!e) e = hrt_hip_malloc((void **)&g->d_meta, (uint64_t)g->meta_words * 4);
    if (!e) e = hrt_hip_malloc((void **)&g->d_meta_all, (uint64_t)world * g->meta_words * 4);
    if (!e) e = hrt_hip_malloc((void **)&g->d_prefix, (uint64_t)g->nb * g->nrx * (g->L.cap / 64) * 4);
    if (!e) e = hrt_hip_malloc(&g->d_segs, (uint64_t)g->segs_cap * 24);
    if (!e) e = hrt_hip_malloc((void **)&g->d_dst, (uint64_t)g->nb * g->nrx * 8);
    if (e || !g->h_meta_all || !g->have || !g->d_recv || !g->recv_cap || !g->h_segs) {
        hrt_gather_destroy(g);
        return e ? hrt_fail_hip(e, "hipMalloc(gather)") : hrt_fail(HRT_E_NOMEM, "out of host memory");
    }
    *out = g;
    return HRT_OK;
}

uint32_t hrt_gather_meta_words(const hrt_gather *g) { return g ? g->meta_words : 0u; }
const uint32_t *hrt_gather_meta_device(const hrt_gather *g) { return g ? g->d_meta : NULL; }
const uint32_t *hrt_gather_meta(const hrt_gather *g, uint32_t r)
{
    return (g && r < g->s.count && g->have[r]) ? g->h_meta_all + (size_t)r * g->meta_words : NULL;
}

/* the meta block of a finished trace, on the device: its counts and -- per (bounce, rx) -- its unblocked
 * records (and, for the compacting pack, their prefix sums) */
int hrt_gather_prepare(hrt_gather *g, const void *d_ws, void *stream)
{
    if (!g || !d_ws) return hrt_fail(HRT_E_INVALID, "hrt_gather_prepare: NULL argument");
    HRT_HIP(hrt_hip_set_device(g->p->device), "hipSetDevice");
    HRT_HIP(hrt_hip_d2d_async(g->d_meta, (const uint8_t *)d_ws + g->L.off_counts, (uint64_t)(g->nb + 2u) * 4, stream), "hipMemcpyAsync");
    HRT_HIP(hrt_hip_export_prefix(d_ws, g->L.off_counts, g->L.off_masks, g->L.cap, g->nb, g->nrx, g->d_prefix,
                                  g->d_meta + g->nb + 2u, stream), "hrt_export_prefix_kernel");
    memset(g->have, 0, g->s.count);
    return HRT_OK;
}

int hrt_gather_set_meta(hrt_gather *g, uint32_t r, const uint32_t *h_meta)
{
    if (!g || !h_meta || r >= g->s.count) return hrt_fail(HRT_E_INVALID, "hrt_gather_set_meta: bad argument");
    memcpy(g->h_meta_all + (size_t)r * g->meta_words, h_meta, (size_t)g->meta_words * 4);
    g->have[r] = 1;
    return HRT_OK;
}

static int grow(uint32_t **buf, uint64_t *cap, uint64_t words)
{
    if (*buf && *cap >= words) return 0;
    if (*buf) hrt_hip_free(*buf);
    *buf = NULL;
    const uint64_t n = words + words / 8 + 1024;   /* a little headroom: the counts move from step to step */
    const int e = hrt_hip_malloc((void **)buf, n * 4);
    *cap = e ? 0 : n;
    return e;
}

/* this rank's export, packed (its meta must be known: hrt_gather_set_meta(rank) or hrt_gather_rccl) */
int hrt_gather_pack(hrt_gather *g, const void *d_ws, void *stream, const void **d_buf, uint64_t *words)
{
    if (!g || !d_ws) return hrt_fail(HRT_E_INVALID, "hrt_gather_pack: NULL argument");
    const uint32_t me = g->s.rank;
    if (!g->have[me]) return hrt_fail(HRT_E_INVALID, "hrt_gather_pack: this rank's meta block is not set");
    const uint32_t *meta = g->h_meta_all + (size_t)me * g->meta_words;
    const uint64_t n = hrt_export_words(meta, g->nb, g->nrx, g->flags);
    HRT_HIP(hrt_hip_set_device(g->p->device), "hipSetDevice");
    if (grow(&g->d_pack, &g->pack_cap, n)) return hrt_fail(HRT_E_NOMEM, "hipMalloc(export) failed");
    struct seg { uint64_t src, dst, n; } *sg = (struct seg *)g->h_segs;
    uint64_t *dst_pair = (uint64_t *)calloc((size_t)g->nb * g->nrx, 8);
    if (!dst_pair) return hrt_fail(HRT_E_NOMEM, "out of host memory");
    uint32_t ns = 0;
    uint64_t max_words = 0, max_hits = 0;
    const hrt_layout *L = &g->L;
    for (uint32_t b = 0; b < g->nb; ++b) {
        const uint64_t H = meta[b + 1];
        if (H > max_hits) max_hits = H;
        hrt_export_part part;
        for (uint32_t rx = 0; rx < g->nrx; ++rx) {
            hrt_export_locate(meta, g->nb, g->nrx, g->flags, b, rx, &part);
            if (rx == 0 && H)
                for (uint32_t f = 0; f < N_HIT_ROWS; ++f) {
                    sg[ns].src = L->off_hits + b * L->hit_block_bytes + (uint64_t)f * L->cap * 4;
                    sg[ns].dst = part.off_hit + (uint64_t)f * H;
                    sg[ns].n = H;
                    ++ns;
                }
            if (g->flags & HRT_EXPORT_UNBLOCKED) { dst_pair[b * g->nrx + rx] = part.off_index; continue; }
            if (!H) continue;
            for (uint32_t f = 0; f < HRT_REC_FIELDS; ++f) {
                sg[ns].src = L->off_recs + b * L->rec_block_bytes + ((uint64_t)rx * HRT_REC_FIELDS + f) * L->cap * 4;
                sg[ns].dst = part.off_rec + (uint64_t)f * H;
                sg[ns].n = H;
                ++ns;
            }
            sg[ns].src = L->off_masks + ((uint64_t)b * g->nrx + rx) * (L->cap / 64) * 8;
            sg[ns].dst = part.off_mask;
            sg[ns].n = 2u * ((H + 63u) / 64u);
            ++ns;
        }
    }
    for (uint32_t k = 0; k < ns; ++k) if (sg[k].n > max_words) max_words = sg[k].n;
    int e = 0;
    if (ns) e = hrt_hip_h2d_async(g->d_segs, sg, (uint64_t)ns * 24, stream);
    if (!e && ns) e = hrt_hip_export_copy(d_ws, g->d_segs, ns, max_words, g->d_pack, stream);
    if (!e && (g->flags & HRT_EXPORT_UNBLOCKED) && max_hits) {
        e = hrt_hip_h2d_async(g->d_dst, dst_pair, (uint64_t)g->nb * g->nrx * 8, stream);
        if (!e) e = hrt_hip_export_compact(d_ws, L->off_counts, L->off_masks, L->off_recs, L->rec_block_bytes, L->cap, g->nb, g->nrx,
                                           max_hits, g->d_prefix, g->d_meta + g->nb + 2u, g->d_dst, g->d_pack, stream);
    }
    /* (the two small tables were copied from pageable memory: those copies have completed on return) */
    if (!e) e = hrt_hip_stream_sync(stream);
    free(dst_pair);
    if (e) return hrt_fail_hip(e, "export pack");
    if (d_buf) *d_buf = g->d_pack;
    if (words) *words = n;
    return HRT_OK;
}

int hrt_gather_export(const hrt_gather *g, uint32_t r, const void **d_buf, uint64_t *words)
{
    if (!g || r >= g->s.count || !g->have[r]) return hrt_fail(HRT_E_INVALID, "hrt_gather_export: rank %u has no export here", r);
    const uint64_t n = hrt_export_words(g->h_meta_all + (size_t)r * g->meta_words, g->nb, g->nrx, g->flags);
    const void *b = (r == g->s.rank) ? (const void *)g->d_pack : (const void *)g->d_recv[r];
    if (n && !b) return hrt_fail(HRT_E_INVALID, "hrt_gather_export: rank %u has no export here", r);
    if (d_buf) *d_buf = b;
    if (words) *words = n;
    return HRT_OK;
}

/* the receive buffer of peer r on the root (for callers with their own transport) */
int hrt_gather_recv_buffer(hrt_gather *g, uint32_t r, void **d_buf, uint64_t *words)
{
    if (!g || r >= g->s.count || !g->have[r]) return hrt_fail(HRT_E_INVALID, "hrt_gather_recv_buffer: meta of rank %u is not set", r);
    const uint64_t n = hrt_export_words(g->h_meta_all + (size_t)r * g->meta_words, g->nb, g->nrx, g->flags);
    HRT_HIP(hrt_hip_set_device(g->p->device), "hipSetDevice");
    if (grow(&g->d_recv[r], &g->recv_cap[r], n)) return hrt_fail(HRT_E_NOMEM, "hipMalloc(receive buffer) failed");
    if (d_buf) *d_buf = g->d_recv[r];
    if (words) *words = n;
    return HRT_OK;
}

/* ---- RCCL, bound at run time ---- */
typedef int (*fn_group)(void);
typedef int (*fn_p2p)(void *, size_t, int, int, void *, void *);
typedef int (*fn_allgather)(const void *, void *, size_t, int, void *, void *);
typedef int (*fn_uid)(void *);
typedef int (*fn_init)(void **, int, hrt_rccl_id, int);
typedef int (*fn_destroy)(void *);
typedef const char *(*fn_errstr)(int);
static struct {
    void *h;
    fn_group group_start, group_end;
    fn_p2p send, recv;
    fn_allgather all_gather;
    fn_uid unique_id;
    fn_init comm_init;
    fn_destroy comm_destroy;
    fn_errstr err;
} R;
enum { NCCL_INT32 = 2, NCCL_UINT32 = 3 };   /* ncclDataType_t (rccl.h) */

static int rccl_bind(void)
{
    if (R.h) return HRT_OK;
    static const char *names[] = {"librccl.so", "librccl.so.1"};
    void *h = NULL;
    for (int k = 0; k < 2 && !h; ++k) h = dlopen(names[k], RTLD_NOW | RTLD_NOLOAD);   /* the copy already loaded (torch's) */
    for (int k = 1; k >= 0 && !h; --k) h = dlopen(names[k], RTLD_NOW | RTLD_GLOBAL);
    if (!h) return hrt_fail(HRT_E_HIP, "librccl not found: %s", dlerror());
    R.group_start = (fn_group)dlsym(h, "ncclGroupStart");
    R.group_end = (fn_group)dlsym(h, "ncclGroupEnd");
    R.send = (fn_p2p)dlsym(h, "ncclSend");
    R.recv = (fn_p2p)dlsym(h, "ncclRecv");
    R.all_gather = (fn_allgather)dlsym(h, "ncclAllGather");
    R.unique_id = (fn_uid)dlsym(h, "ncclGetUniqueId");
    R.comm_init = (fn_init)dlsym(h, "ncclCommInitRank");
    R.comm_destroy = (fn_destroy)dlsym(h, "ncclCommDestroy");
    R.err = (fn_errstr)dlsym(h, "ncclGetErrorString");
    if (!R.group_start || !R.group_end || !R.send || !R.recv || !R.all_gather || !R.unique_id || !R.comm_init || !R.comm_destroy)
        return hrt_fail(HRT_E_HIP, "librccl lacks an expected symbol");
    R.h = h;
    return HRT_OK;
}
static int rccl_fail(int e, const char *what)
{
    return hrt_fail(HRT_E_HIP, "%s failed: %s (%d)", what, R.err ? R.err(e) : "?", e);
}

int hrt_rccl_unique_id(hrt_rccl_id *out)
{
    int rc = rccl_bind();
    if (rc) return rc;
    if (!out) return hrt_fail(HRT_E_INVALID, "hrt_rccl_unique_id: NULL argument");
    const int e = R.unique_id(out);
    return e ? rccl_fail(e, "ncclGetUniqueId") : HRT_OK;
}
int hrt_rccl_comm_create(const hrt_rccl_id *id, int world, int rank, int device, void **comm)
{
    int rc = rccl_bind();
    if (rc) return rc;
    if (!id || !comm) return hrt_fail(HRT_E_INVALID, "hrt_rccl_comm_create: NULL argument");
    HRT_HIP(hrt_hip_set_device(device), "hipSetDevice");
    const int e = R.comm_init(comm, world, *id, rank);
    return e ? rccl_fail(e, "ncclCommInitRank") : HRT_OK;
}
int hrt_rccl_comm_destroy(void *comm)
{
    if (!comm || !R.h) return HRT_OK;
    const int e = R.comm_destroy(comm);
    return e ? rccl_fail(e, "ncclCommDestroy") : HRT_OK;
}

/* the whole exchange over RCCL: meta blocks to everybody, packed exports to the root.  Blocks until the
 * root holds every export (hrt_gather_export).  `self_loop` (tests on one GPU: world 1): the root also
 * sends its own export to itself through ncclSend / ncclRecv. */
int hrt_gather_rccl(hrt_gather *g, void *comm, const void *d_ws, void *stream, int self_loop)
{
    int rc = rccl_bind();
    if (rc) return rc;
    if (!g || !comm || !d_ws) return hrt_fail(HRT_E_INVALID, "hrt_gather_rccl: NULL argument");
    const uint32_t world = g->s.count, me = g->s.rank;
    if ((rc = hrt_gather_prepare(g, d_ws, stream))) return rc;
    int e = R.all_gather(g->d_meta, g->d_meta_all, g->meta_words, NCCL_UINT32, comm, stream);
    if (e) return rccl_fail(e, "ncclAllGather");
    HRT_HIP(hrt_hip_d2h_async(g->h_meta_all, g->d_meta_all, (uint64_t)world * g->meta_words * 4, stream), "hipMemcpyAsync");
    HRT_HIP(hrt_hip_stream_sync(stream), "hipStreamSynchronize");
    memset(g->have, 1, world);
    const void *mine = NULL;
    uint64_t n_mine = 0;
    if ((rc = hrt_gather_pack(g, d_ws, stream, &mine, &n_mine))) return rc;
    const int is_root = (int)me == g->root;
    if (is_root)
        for (uint32_t r = 0; r < world; ++r)
            if ((r != me || self_loop) && (rc = hrt_gather_recv_buffer(g, r, NULL, NULL))) return rc;
    if ((e = R.group_start())) return rccl_fail(e, "ncclGroupStart");
    if (is_root) {
        for (uint32_t r = 0; r < world && !e; ++r) {
            if (r == me && !self_loop) continue;
            const uint64_t n = hrt_export_words(g->h_meta_all + (size_t)r * g->meta_words, g->nb, g->nrx, g->flags);
            if (n) e = R.recv(g->d_recv[r], n, NCCL_INT32, (int)r, comm, stream);
        }
    }
    if (!e && (!is_root || self_loop) && n_mine) e = R.send((void *)mine, n_mine, NCCL_INT32, g->root, comm, stream);
    const int e2 = R.group_end();
    if (e) return rccl_fail(e, "ncclSend / ncclRecv");
    if (e2) return rccl_fail(e2, "ncclGroupEnd");
    HRT_HIP(hrt_hip_stream_sync(stream), "hipStreamSynchronize");
    return HRT_OK;
}

/* (tests) the buffer rank r's export was RECEIVED into on the root */
const void *hrt_gather_received(const hrt_gather *g, uint32_t r) { return (g && r < g->s.count) ? g->d_recv[r] : NULL; }

/* ---- struct sizes of this build (a binding checks them against its own mirror: hrt_stats and hrt_layout
 * are written in full by the library) ---- */
uint64_t hrt_stats_size(void) { return sizeof(hrt_stats); }
uint64_t hrt_layout_size(void) { return sizeof(hrt_layout); }
