/* hrt_internal.h -- shared declarations of the host C part of libhermespy_rt_amd. */
#ifndef HRT_INTERNAL_H
#define HRT_INTERNAL_H

#include <stddef.h>
#include <stdint.h>

#include "hermespy_rt.h"
#include "hrt_device.h"
#include "../hrt_kparams.h"

/* src/compute_paths.c:18-19 of the reference: both are FLOAT constants there */
#define HRT_PI_F 3.14159265358979323846f
#define HRT_C_F 299792458.0f

/* ITU-R P.2040-3 material parameters (materials.c) */
typedef struct {
    float a, b, c, d;   /* eta' = a f^b, sigma = c f^d (f in GHz) */
    float s;            /* scattering coefficient */
    uint8_t s1_alpha;   /* directive lobe width */
} hrt_material;
extern const hrt_material hrt_materials[HRT_NUM_MATERIALS];

/* src/compute_paths.c:125-132 field order */
typedef struct {
    float eta_re, eta_sqrt_re, eta_inv_re, eta_inv_sqrt_re;
    float eta_im, eta_sqrt_im, eta_inv_im, eta_inv_sqrt_im;
    float eta_abs, eta_abs_pow2, eta_abs_inv_sqrt;
    float r;
} hrt_eta;
void hrt_material_eta(uint32_t material_index, float f_ghz, hrt_eta *out);

/* developer / test switches (tune.c: HRT_TUNE="key=value,..."), read once per problem */
typedef struct {
    hrt_ktune k;                 /* the part the launch shims see (hrt_kparams.tune) */
    double wide_cos;             /* 0: by table size */
    int64_t wide_cap;            /* -1: auto */
    int sort_rays, sort_fine, sort_dir_res;   /* sort_rays -1: by table size */
    uint64_t rxt_min_rays;       /* UINT64_MAX: by table size */
    uint32_t rxt_max_tri;
    int no_rxt, no_txt, no_reorder, no_patch;
    double patch_size, accel_sparse;
    uint64_t accel_big, accel_fine_min;   /* accel_fine_min UINT64_MAX: HRT_FINE_MIN_TRI */
    int accel_fine;              /* -1: auto, 0: never */
    int no_bounce_prefetch, no_scatter;
    int no_chain;                  /* launches 1 .. nb one by one instead of hrt_chain_kernel */
    int chain_from;                /* first launch of the chain kernel (default 1) */
} hrt_tune;
void hrt_tune_defaults(hrt_tune *t);
int hrt_tune_load(hrt_tune *t);

/* acceleration structure over the triangle table (accel.c; device side in hrt_kernels.hip) */
typedef struct hrt_accel {
    uint32_t num_tri, num_leaf;
    int big;                     /* inner levels + plane tree present */
    uint32_t *orig;              /* [T] table row -> index in the reference's (mesh, face) loop order */
    uint32_t *newidx;            /* [T] inverse */
    float *leaf;                 /* [num_leaf][HRT_NODE_FLOATS]: c.xyz, R, Lambda */
    float *tg;                   /* [T][2]: qs, longest edge */
    uint32_t num_levels;         /* inner levels above the leaves */
    uint32_t node_count[HRT_ACCEL_MAX_LEVELS];
    float *node[HRT_ACCEL_MAX_LEVELS];
    uint32_t pl_num_leaf, pl_levels;
    uint32_t pl_count[HRT_ACCEL_MAX_LEVELS];
    float *pl_node[HRT_ACCEL_MAX_LEVELS];   /* [0] = cones of the 64-entry leaves */
    uint32_t *pl_index;          /* [pl_num_leaf * 64] table rows, HRT_NO_HIT padded */
    float *pl_rec;               /* [pl_num_leaf * 64][HRT_NODE_FLOATS]: p1.xyz, l, n.xyz, qs */
    int planes;                  /* plane tree present (big, or the fine leaves) */
    uint32_t num_fine;
    float *fine;                 /* [num_fine][HRT_NODE_FLOATS]: sphere + Lambda of HRT_FINE_ROWS rows, or NULL */
} hrt_accel;
int hrt_accel_order(hrt_accel *a, const float *rows_ref_order, uint32_t T, int reorder);
int hrt_accel_build(hrt_accel *a, const float *rows_table_order, const hrt_tune *tune);
void hrt_accel_free(hrt_accel *a);

struct hrt_problem {
    int device;
    hrt_tune tune;
    uint32_t num_tri, num_mesh, num_rx, num_tx;
    float f_ghz, fsl_mult, dop_mult;
    /* host copies */
    float *h_tri;       /* [num_tri][HRT_TRI_FLOATS] */
    float *h_mesh;      /* [num_mesh][HRT_MESH_FLOATS] */
    float *h_mat;       /* [17][HRT_MAT_FLOATS] */
    uint32_t *h_tri_mesh, *h_tri_face;   /* per table row */
    hrt_accel accel;
    hrt_eta eta[HRT_NUM_MATERIALS];
    /* one device allocation holding everything */
    void *d_blob;
    const float *d_tri, *d_mesh, *d_mat, *d_rx_pos, *d_tx_pos, *d_rx_vel, *d_tx_vel;
    const uint32_t *d_inv;      /* [T] original index -> table row (only with the fine leaves: the wide kernels) */
    hrt_kaccel kaccel;           /* device pointers of the acceleration structure */
    void *d_rxt;                 /* per-RX direction tables (device blob), or NULL */
    hrt_krxt krxt;
    uint64_t rxt_entries;        /* total list entries over all (rx, cell) */
    uint32_t *h_fuse_flag, *d_fuse_flag;   /* pinned words a fused launch / the chain kernel sets when it gives up (hrt_kparams.host_flag) */
    void *aux_stream;            /* second stream of a trace: the records kernels run beside the bounce kernels */
    void *aux_ev[2];             /* fork (live list complete) / join (records done): ordering-only events */
    void *d_patch;               /* patch tables (device blob), or NULL */
    hrt_kpatch kpatch;
    int sort_rays;               /* re-sort the live list between bounces (hrt_ksort) */
    float scene_lo[3], scene_hi[3];   /* bounding box of the (finite) vertices */
};

/* hrt_problem_create with the number of rays the problem will trace (decides whether the direction
 * tables are worth building: problem.c) */
int hrt_problem_create_for(const Scene *scene, const Vec3 *rx_pos, const Vec3 *tx_pos,
                           const Vec3 *rx_vel, const Vec3 *tx_vel, float f_ghz, size_t num_rx,
                           size_t num_tx, int device, uint64_t rays_hint, hrt_problem **out);

/* error plumbing: set the thread's last-error text and return `code` */
int hrt_fail(int code, const char *fmt, ...);
int hrt_fail_hip(int hip_err, const char *what);
#define HRT_HIP(call, what)                                   \
    do {                                                      \
        int e__ = (call);                                     \
        if (e__ != 0) return hrt_fail_hip(e__, (what));       \
    } while (0)

double hrt_now_s(void);
/* fused launches off for the rest of the process (a fused launch timed out: the GPU is shared with other fused
 * kernels); hrt_trace looks at it, and at the problem's pinned flag word, before every trace */
void hrt_fuse_disable(void);
int hrt_fuse_disabled(void);
void hrt_chain_disable(void);
int hrt_chain_disabled(void);
int hrt_void_step_retry(uint32_t err_word);
#define HRT_POOL_MAX_DEFAULT (5ull << 30)   /* HRT_POOL_MAX_BYTES: what a worker may keep between calls */
uint64_t hrt_worker_held_bytes(uint64_t ws_bytes, uint64_t dirs_rows, uint64_t cap);
int hrt_batch_fits_pool(uint64_t ws_bytes, uint64_t dirs_rows, uint64_t cap);
void hrt_list_cache_clear(void);   /* path_list.c: the blocks kept from the last freed path list */

/* ---- buffers of one device worker of the drop-in calls (compute_paths.c), pooled between calls ---- */
typedef struct {
    void *d_dirs, *d_ws, *d_order;
    uint32_t *h_order;      /* coherent launch order of one batch */
    float *h_dirs;          /* launch directions of the whole sphere, [np][3] */
    uint32_t *h_counts;
    float *h_los;
    uint32_t *ray, *tri;    /* per-bounce downloads */
    uint32_t *ray2, *tri2;  /* ... of the NEXT bounce, requested while the last block of this one is written */
    float *fs0, *fs02;      /* launch Doppler term of the hits (path-list writer), this bounce's and the next's */
    float *st[6];           /* o, d of the hits */
    float *hs[4], *hs2[4];  /* origin and delay of the hits after the bounce (this bounce's / the next's): the dense
                             * writer forms directions_rx and tau of the records from them */
    float *rec[HRT_REC_FIELDS];
    uint64_t *mask;
    float *rec2[HRT_REC_FIELDS];   /* second staging set: the copy of the next (bounce, rx) block */
    uint64_t *mask2;              /* overlaps the dense scatter of the current one */
    void *copy_stream, *copy_stream2;   /* two streams: two DMA engines (one engine moves ~28 GB/s) */
    Ray *cur_rays;          /* RaysInfo emulation: state of every ray of the current batch, [ntx][n_loc] */
    uint8_t *active, *next_active;   /* (unused since the snapshots are per batch; kept for the pool's layout) */
    float *dirs_batch;      /* gathered launch directions of one batch */
    uint64_t *run_start;    /* per bounce: runs of equal TX in the hit list */
    uint32_t *run_tx;
    int device;
} work_t;

/* everything one device worker needs */
typedef struct {
    /* the call (shared, read-only) */
    Scene *scene;
    const Vec3 *rx_pos, *tx_pos, *rx_vel, *tx_vel;
    float f_ghz;
    size_t nrx, ntx, np, nb, nq;
    ChannelInfo *los, *scat;
    RaysInfo *los_rays, *scat_rays;
    uint32_t G;                 /* batches = round-robin shards of the launch set */
    uint8_t *act_all;           /* RaysInfo: [nb + 1][nq / 8 + 1] active bits of every ray before bounce 0 .. after the
                                 * last one, shared by the workers (each sets the bits of its own paths) */
    int host_launch, scatter_threads, use_pool;
    size_t amp_stride;          /* floats between consecutive amplitude entries: 1 (the reference's planes) or
                                 * 2 (re/im interleaved: hrt_compute_paths_interleaved) */
    /* this worker */
    int index, count;           /* handles batches index, index + count, ... */
    int device;
    hrt_problem *prob;
    work_t w;
    uint64_t cap_alloc, ws_alloc, dirs_rows_alloc;
    hrt_stats st;
    double t_dev, t_rb, t_launch;
    int rc;
    char err[512];
} dev_ctx;
int hrt_pool_begin(void);             /* 1 if this call owns the pool of kept buffers */
void hrt_pool_end(int taken);
int hrt_worker_alloc(dev_ctx *c);     /* needs c->prob, nrx, ntx, np, nb, G, index, device, use_pool, scat_rays */
void hrt_worker_release(dev_ctx *c);  /* back to the pool (when c->rc == HRT_OK) or freed */

/* host helpers shared by the dense writer and the path-list writer (compute_paths.c) */
#define HRT_MAX_SCATTER_THREADS 32
typedef void (*hrt_range_fn)(void *ctx, uint64_t i0, uint64_t i1, int tid);
void hrt_parallel_ranges(hrt_range_fn fn, void *ctx, uint64_t n, int threads);   /* tid < 32 */
int hrt_host_threads(void);                      /* HRT_HOST_THREADS or min(16, cores) */
void hrt_parallel_release(void);                 /* ends the calling thread's parked helper threads */
int hrt_launch_cache_enabled(uint64_t np);
int hrt_launch_cache_get(uint64_t np, float *dirs, uint32_t *order);   /* copies; 1 if served */
void hrt_launch_cache_put(uint64_t np, const float *dirs, const uint32_t *order);

#endif
