/* hrt_device.h -- device-resident form of the compute_paths hot path (C ABI).
 *
 * compute_paths() (hermespy_rt.h) takes and returns HOST arrays in the reference's dense
 * layout.  The functions here are the same path with everything resident in HBM and the
 * result in COMPACT form -- what a multi-GPU caller (one process per GPU, ray-sharded) and
 * bench.py drive.  compute_paths() itself is built on them.
 *
 * The caller owns the big buffers (launch directions, workspace): they are plain device
 * pointers, e.g. torch tensors' data_ptr(), and `stream` is a hipStream_t passed as void*
 * (NULL = the default stream).  No torch or HIP types appear in any signature.
 *
 * Reference lines each piece replaces (relative to the reference repository):
 *   hrt_problem_create   src/compute_paths.c:437-438 (precompute_materials/_normals, :171-224)
 *                        + the scene/endpoint arguments of compute_paths (:419-429)
 *   hrt_launch_dirs_host src/compute_paths.c:443-451 (Fibonacci sphere, double libm)
 *   hrt_trace            src/compute_paths.c:460-472 (state init), :515-577 (LoS),
 *                        :591-729 (bounce loop: moeller_trumbore :237-287, refl_coefs
 *                        :300-344, scat_coefs :359-415)
 */
#ifndef HRT_DEVICE_H
#define HRT_DEVICE_H

#include "hermespy_rt.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Everything small and read-only, uploaded once: flattened triangle table (v1, e1, e2, unit
 * normal, mesh id; (mesh, face) order), per-mesh material/velocity, the eta table for the
 * carrier frequency, RX/TX positions and velocities. */
typedef struct hrt_problem hrt_problem;

int hrt_problem_create(const Scene *scene, const Vec3 *rx_pos, const Vec3 *tx_pos,
                       const Vec3 *rx_vel, const Vec3 *tx_vel, float carrier_frequency_GHz,
                       size_t num_rx, size_t num_tx, int device, hrt_problem **out);
void hrt_problem_destroy(hrt_problem *p);
/* Fused launches and the chain kernel wait for each other's workgroups; when a wait runs out (the GPU is shared
 * with other such kernels) the step is void (HRT_ERR_FUSE_TIMEOUT / HRT_ERR_CHAIN_TIMEOUT in counts[nb + 1]), the
 * library switches that kernel off for the rest of the process and callers trace again (the drop-in calls do so
 * themselves).  hrt_fallback_state: bit 0 = fused launches are off, bit 1 = the chain kernel is off. */
int hrt_fallback_state(void);
#ifndef HRT_ERR_FUSE_TIMEOUT   /* (the same values in csrc/hrt_kparams.h, for the kernels) */
#define HRT_ERR_FUSE_TIMEOUT 0x100u
#define HRT_ERR_CHAIN_TIMEOUT 0x200u
#define HRT_ERR_VOID (HRT_ERR_FUSE_TIMEOUT | HRT_ERR_CHAIN_TIMEOUT)
#endif
uint32_t hrt_problem_num_triangles(const hrt_problem *p);
uint32_t hrt_problem_num_rx(const hrt_problem *p);
uint32_t hrt_problem_num_tx(const hrt_problem *p);
int hrt_problem_device(const hrt_problem *p);

/* Host copies of what was uploaded (for tests): eta table [17][12] in the reference's
 * MaterialPrecomputed field order (src/compute_paths.c:125-132); normals [T][3]; for row j of
 * the device table its (mesh, face) pair. */
int hrt_problem_eta_table(const hrt_problem *p, float *out17x12);
int hrt_problem_normals(const hrt_problem *p, float *outTx3);   /* in the reference's loop order */
int hrt_problem_tri_ids(const hrt_problem *p, uint32_t *mesh_out, uint32_t *face_out);
/* The device table is kept in a spatial (Morton) order for the acceleration structure
 * (csrc/host/accel.c); HRT_HIT_TRI values are rows of THAT table.  out[j] = position of row j in
 * the reference's (mesh, face) loop order (src/compute_paths.c:253-254), i.e. the flat index the
 * reference's scan would give the same triangle.  Environment: HRT_NO_REORDER=1 keeps the
 * reference's order (the identity here). */
int hrt_problem_tri_order(const hrt_problem *p, uint32_t *orig_of_row);

/* A shard of the launch set.  The N = num_paths launch directions of every TX are cut into
 * granules of `chunk` consecutive path indices dealt round-robin to `count` shards, so every
 * shard sees the whole sphere (a contiguous block would be a polar cap with a very different
 * hit rate).  Local ray i of shard r is global path
 *     p = ((i / chunk) * count + r) * chunk + i % chunk .
 * count = 1 is the unsharded problem. */
typedef struct {
    uint64_t num_paths;    /* global N per TX (the N in k/N of the Fibonacci sphere) */
    uint32_t rank, count;  /* shard r of G */
    uint32_t chunk;        /* granule; multiple of 64; 0 = 4096 */
    uint32_t num_bounces;
} hrt_shard;

uint64_t hrt_shard_num_local(const hrt_shard *s);
uint64_t hrt_shard_global_path(const hrt_shard *s, uint64_t local_i);

/* Launch directions of this shard's local rays, [num_local][3] floats, on the HOST with the
 * host libm (bit-identical to the reference's by construction).  num_threads <= 0: all
 * cores. */
int hrt_launch_dirs_host(const hrt_shard *s, float *out, int num_threads);

/* The same directions generated ON THE DEVICE into d_dirs (device [num_local][3] floats),
 * bit-identical to hrt_launch_dirs_host: the device evaluates the reference's formula with its
 * own double math library and flags every value whose rounding to float could depend on that
 * library's last bits (closer than 2^-44 relative to a float rounding boundary; about one ray
 * in a million); the flagged rays are recomputed with the host libm and patched in.  Blocks
 * until done; *num_patched (may be NULL) returns how many rays were patched. */
int hrt_launch_dirs_device(const hrt_shard *s, float *d_dirs, int device, void *stream,
                           uint64_t *num_patched);

/* Coherent launch order of this shard: a permutation of 0..num_local-1 (order[i] = local ray
 * launched by lane i) that walks the sphere in z-bands, serpentine in azimuth, so that 64
 * consecutive lanes -- one wavefront -- form a narrow ray packet (what the packet culling of
 * the trace kernel feeds on).  Pure function of `dirs` ([num_local][3], from
 * hrt_launch_dirs_host); dirs == NULL derives the keys from the path indices alone (for
 * directions generated on the device).  Results do not depend on the order (records carry ray ids); only
 * speed does. */
int hrt_launch_order_host(const hrt_shard *s, const float *dirs, uint32_t *order_out);

/* The same kind of order computed ON THE DEVICE into d_order (device [num_local] u32): bands are
 * ranges of the local index on a Fibonacci sphere, so one workgroup sorts one band segment by
 * azimuth in LDS.  Not the same permutation as hrt_launch_order_host (coarser azimuth keys), equally
 * coherent; what compute_paths() uses.  Blocks until done. */
int hrt_launch_order_device(const hrt_shard *s, uint32_t *d_order, int device, void *stream);

/* ---- workspace layout ----
 * cap = num_tx * num_local rounded up to 256 entries.  Every array below holds `cap`
 * elements of 4 bytes unless noted, so a field is a contiguous, coalesced run.
 *
 *   counts      u32[num_bounces + 2]   counts[b] (b >= 1) = rays that hit at bounce b-1
 *                                      = live rays entering bounce b; counts[0] unused;
 *                                      counts[num_bounces+1] = internal error flags (0)
 *   los         num_rx*num_tx entries of HRT_LOS_FLOATS floats
 *   hit block b (b = 0 .. num_bounces-1), fields HRT_HIT_*:
 *       the rays that hit something at bounce b, in the (stable) order of the live list they
 *       came from, with their state AFTER the bounce.  It is also the live list of bounce b+1.
 *   rec block b: for rx in 0..num_rx-1, fields HRT_REC_*: the scatter record of
 *       (hit i of block b, rx) at element i; plus one bit per (rx, i) "unblocked".
 *       A blocked record has a_* = tau = 0 and its dir/dfs elements are not written.
 */
enum {
    HRT_HIT_RAY = 0,   /* u32: tx * num_local + local_i */
    HRT_HIT_TRI,       /* u32: flat triangle index of the hit */
    HRT_HIT_THETA,     /* incidence angle */
    HRT_HIT_FS0,       /* launch Doppler term dot(tx_vel, d_launch) * f/c */
    HRT_HIT_OX, HRT_HIT_OY, HRT_HIT_OZ, HRT_HIT_DX, HRT_HIT_DY, HRT_HIT_DZ,
    HRT_HIT_A_TE_RE, HRT_HIT_A_TE_IM, HRT_HIT_A_TM_RE, HRT_HIT_A_TM_IM,
    HRT_HIT_TAU,
    HRT_HIT_FIELDS
};
enum {
    HRT_REC_A_TE_RE = 0, HRT_REC_A_TE_IM, HRT_REC_A_TM_RE, HRT_REC_A_TM_IM,
    HRT_REC_TAU,
    HRT_REC_DIRX, HRT_REC_DIRY, HRT_REC_DIRZ,   /* directions_rx */
    HRT_REC_DFS,       /* dot(d_to_rx - d_reflected, mesh_velocity) * f/c; the record's
                          freq_shift is FS0 - DFS */
    HRT_REC_FIELDS
};
enum {
    HRT_LOS_STATUS = 0,  /* as float bits of u32: 0 coincident, 1 blocked, 2 clear */
    HRT_LOS_A, HRT_LOS_TAU, HRT_LOS_DIRX, HRT_LOS_DIRY, HRT_LOS_DIRZ, HRT_LOS_FS,
    HRT_LOS_FLOATS = 8
};

typedef struct {
    uint64_t total_bytes;
    uint64_t cap;               /* elements per array */
    uint64_t off_counts;
    uint64_t off_los;
    uint64_t off_hits;          /* field f of block b at off_hits + b*hit_block_bytes + f*cap*4 */
    uint64_t hit_block_bytes;
    uint64_t off_recs;          /* field f of (b, rx) at off_recs + b*rec_block_bytes
                                                       + (rx*HRT_REC_FIELDS + f)*cap*4 */
    uint64_t rec_block_bytes;
    uint64_t off_masks;         /* u64 words of (b, rx) at off_masks + (b*num_rx + rx)*(cap/64)*8 */
    /* scratch of the stable compaction: survivor counts per WAVE of every 256-entry chunk (4 u32 per chunk), and per bounce the
     * sums over every 32 chunks (u32 [num_bounces][num_super]) */
    uint64_t off_chunk_cnt, off_super_cnt;
    /* trace results of one launch (internal scratch between the two kernels): for trace kind k
     * (0..num_rx-1 shadow to rx k, num_rx the bounce itself) a block of 2*cap words at
     * off_res + 2k*cap*4 -- the bounce: u32 triangle[cap] then f32 distance[cap]; a shadow trace:
     * one packed word per entry (triangle | blocked << 31), a half word on tables of fewer than
     * 32 767 triangles (triangle | blocked << 15) */
    uint64_t off_res;
    uint64_t num_super;
    /* re-sorting of the live list between bounces (only when the problem has it on, HRT_SORT_RAYS):
     * a hit-block-sized scratch, 4 x cap u32 of keys / indices, the sort's temporary storage */
    uint64_t off_sort_scratch, off_sort_keys, off_sort_tmp, sort_tmp_bytes;
    /* status words of the fused kernels (stable compaction inside ONE kernel: per-chunk, per-group
     * and per-supergroup survivor counts): u32 [num_bounces + 1][lb_stride], zeroed with the counts
     * at the start of every trace */
    uint64_t off_lb, lb_stride;
    /* queue of the packets too wide to cull (tables of more than 1 024 triangles): wide_cap entries of
     * 8 bytes, then 64 keys of 8 bytes per entry; wide_cap = 0: none */
    uint64_t off_wide_q, off_wide_key, wide_cap;
} hrt_layout;

/* HRT_E_CAPACITY when num_tx * (local rays) exceeds 2^32 / (HRT_HIT_FIELDS * 4) - 512
 * (~71.5 M): the kernels address the field arrays of a block by 32-bit offsets from the block's
 * buffer descriptor.  Use more shards (the drop-in compute_paths does so by itself). */
int hrt_layout_query(const hrt_problem *p, const hrt_shard *s, hrt_layout *out);

/* Per-launch device times of one hrt_trace call, filled only when requested. */
typedef struct {
    float los_ms;
    float trace_ms[33];         /* trace kernel of launch b = 0..num_bounces (patch tables: the primary rays only;
                                 * a fused launch: its one kernel) */
    float shade_ms[33];         /* shade kernel (Fresnel, reflect, compaction; records unless records_ms) of launch b */
    float records_ms[33];       /* hrt_records_kernel of launch b (patch tables: shadow traces + scatter records);
                                 * 0 where the trace and shade kernels do that work */
    uint32_t num_bounce_launches;
} hrt_kernel_times;

/* Enqueue the whole path on `stream`: zero the counters, LoS kernel, then num_bounces + 1
 * launches; launch b = trace kernel (the num_rx shadow rays of every hit of bounce b-1 and the
 * rays of bounce b: intersection only) + a one-workgroup scan (stable compaction offsets) +
 * shade kernel (scatter records of bounce b-1; Fresnel, delay, reflection of bounce b;
 * survivors written in order into the next live list).
 * Asynchronous unless `times` != NULL, in which case HIP events are recorded around every
 * launch on `stream` and the call returns after the stream drained.
 * d_dirs:  device [num_local][3] floats (this shard's launch directions).
 * d_order: device [num_local] u32 coherent launch order (hrt_launch_order_host), or NULL for
 *          index order (same results, less coherent packets). */
int hrt_trace(const hrt_problem *p, const hrt_shard *s, const float *d_dirs,
              const uint32_t *d_order, void *d_workspace, uint64_t workspace_bytes, void *stream,
              hrt_kernel_times *times);

/* Per-kernel timing WITHOUT a synchronisation per call: a timer owns the HIP events of one
 * hrt_trace; hrt_trace_timed records them on `stream` and returns at once (asynchronous, like
 * hrt_trace with times == NULL); hrt_timer_read waits for the timer's last event and converts.
 * Use one timer per call in flight (bench.py: one per timed step, read after the timed region). */
typedef struct hrt_timer hrt_timer;
int hrt_timer_create(uint32_t num_bounces, hrt_timer **out);
void hrt_timer_destroy(hrt_timer *t);
int hrt_trace_timed(const hrt_problem *p, const hrt_shard *s, const float *d_dirs,
                    const uint32_t *d_order, void *d_workspace, uint64_t workspace_bytes,
                    void *stream, hrt_timer *timer);

/* hrt_trace / hrt_trace_timed (timer may be NULL) with options.  HRT_DIRS_IN_LAUNCH_ORDER: d_dirs
 * is already permuted into launch order, d_dirs[i] = direction of ray d_order[i] (a caller that
 * traces the same launch set repeatedly permutes once; the launch kernels then read contiguous
 * runs instead of gathering 12-byte rows).  Results are identical. */
#define HRT_DIRS_IN_LAUNCH_ORDER 1u
int hrt_trace_flags(const hrt_problem *p, const hrt_shard *s, const float *d_dirs,
                    const uint32_t *d_order, void *d_workspace, uint64_t workspace_bytes,
                    void *stream, hrt_timer *timer, uint32_t flags);
int hrt_timer_read(hrt_timer *t, hrt_kernel_times *out);

/* Algorithmic work of a finished trace from its (host copy of) counts: see hrt_stats. */
void hrt_work_from_counts(const hrt_problem *p, const hrt_shard *s, const uint32_t *counts,
                          hrt_stats *out);

/* ---- packed export and gather: the exchange step of the sharded path (one process per GPU) ----
 * Every rank packs its compact result into ONE run of 32-bit words in HBM (floats bit-cast); the runs are
 * gathered to one rank.  A rank's META block (hrt_export_meta_words(nb, nrx) u32) holds what sizes the run:
 *     meta[0 .. nb + 1]                 the counts block of the trace (meta[b + 1] = H_b, the hits of bounce b)
 *     meta[nb + 2 + b * nrx + rx]       U_(b, rx): the unblocked records of (bounce b, rx)
 * Layout of the run, bounce after bounce (H = H_b):
 *     hit rows   [4][H]                 HRT_HIT_RAY, _TRI, _THETA, _FS0
 *     then per rx -- HRT_EXPORT_FULL:        records [HRT_REC_FIELDS][H], mask [2 * ceil(H / 64)] ("unblocked" bits)
 *                 -- HRT_EXPORT_UNBLOCKED:   index [U] (hit i of each kept record), records [HRT_REC_FIELDS][U]
 *                    (blocked records are all zero: 14 % of C3's; they need not travel)
 * hermespy-rt_amd/sharding.py is the torch.distributed binding of the same layout. */
#define HRT_EXPORT_FULL 0u
#define HRT_EXPORT_UNBLOCKED 1u
typedef struct {
    uint64_t hits, unblocked, records;      /* H_b, U_(b, rx), records in the run (H or U) */
    uint64_t off_hit;                       /* word offsets into the run: hit rows [4][hits] */
    uint64_t off_index;                     /* HRT_EXPORT_UNBLOCKED: [records] hit indices */
    uint64_t off_rec;                       /* [HRT_REC_FIELDS][records] */
    uint64_t off_mask;                      /* HRT_EXPORT_FULL: [2 * ceil(hits / 64)] */
} hrt_export_part;
uint32_t hrt_export_meta_words(uint32_t num_bounces, uint32_t num_rx);
uint64_t hrt_export_words(const uint32_t *meta, uint32_t num_bounces, uint32_t num_rx, uint32_t flags);
int hrt_export_locate(const uint32_t *meta, uint32_t num_bounces, uint32_t num_rx, uint32_t flags, uint32_t bounce,
                      uint32_t rx, hrt_export_part *out);

/* The gather of one shard's rank (s->rank of s->count) to `root`; buffers are kept between steps.
 * With a transport of the caller's own (sharding.py: torch.distributed):
 *     hrt_gather_prepare      meta block of the finished trace in d_workspace, on the device (hrt_gather_meta_device)
 *     (exchange the meta blocks)   hrt_gather_set_meta for every rank whose block is known here
 *     hrt_gather_pack         this rank's run;   hrt_gather_recv_buffer: where the root receives rank r's
 * or, over RCCL (librccl is bound at run time; `comm` is the caller's ncclComm_t, e.g. from
 * hrt_rccl_comm_create): hrt_gather_rccl does all of it -- ncclAllGather of the meta blocks, one group of
 * ncclSend / ncclRecv -- and returns when the root holds every run (hrt_gather_export; the other ranks hold the
 * meta blocks). */
typedef struct hrt_gather hrt_gather;
int hrt_gather_create(const hrt_problem *p, const hrt_shard *s, int root, uint32_t flags, hrt_gather **out);
void hrt_gather_destroy(hrt_gather *g);
uint32_t hrt_gather_meta_words(const hrt_gather *g);
const uint32_t *hrt_gather_meta_device(const hrt_gather *g);
int hrt_gather_prepare(hrt_gather *g, const void *d_workspace, void *stream);
int hrt_gather_set_meta(hrt_gather *g, uint32_t rank, const uint32_t *host_meta);
const uint32_t *hrt_gather_meta(const hrt_gather *g, uint32_t rank);   /* host; NULL if not known */
int hrt_gather_pack(hrt_gather *g, const void *d_workspace, void *stream, const void **d_run, uint64_t *words);
int hrt_gather_recv_buffer(hrt_gather *g, uint32_t rank, void **d_buf, uint64_t *words);
int hrt_gather_export(const hrt_gather *g, uint32_t rank, const void **d_run, uint64_t *words);
const void *hrt_gather_received(const hrt_gather *g, uint32_t rank);
typedef struct { char internal[128]; } hrt_rccl_id;   /* ncclUniqueId */
int hrt_rccl_unique_id(hrt_rccl_id *out);
int hrt_rccl_comm_create(const hrt_rccl_id *id, int world, int rank, int device, void **comm);
int hrt_rccl_comm_destroy(void *comm);
int hrt_gather_rccl(hrt_gather *g, void *comm, const void *d_workspace, void *stream, int self_loop);

/* sizes of the structs this build writes in full (a binding compares them with its own mirror) */
uint64_t hrt_stats_size(void);
uint64_t hrt_layout_size(void);

/* ---- thin helpers over the HIP runtime for C callers without one ---- */
int hrt_device_count(int *out);
int hrt_device_malloc(int device, void **out, uint64_t bytes);
int hrt_device_free(int device, void *ptr);
int hrt_device_upload(int device, void *dst, const void *src, uint64_t bytes);
int hrt_device_download(int device, void *dst, const void *src, uint64_t bytes);
int hrt_device_sync(int device, void *stream);
/* total/free HBM bytes */
int hrt_device_mem_info(int device, uint64_t *free_bytes, uint64_t *total_bytes);

/* Sionna / Mitsuba scene (scene.xml, the PLY files it names, optional scene.csv) -> Scene, the job of
 * the reference's importer tool (src/scene_fromSionna.c:103-488) as a library call.  `out` is
 * filled with malloc()ed meshes (release with free_scene); the file names box.xml and
 * simple_reflector.xml select the two built-in scenes.  See csrc/host/sionna_import.c for the
 * accepted formats.  CLI: lib/hrt_import_sionna. */
int hrt_scene_import_sionna(const char *xml_path, Scene *out);

/* Device self test: evaluates on the GPU, over n host floats, one of the float libm
 * restatements the shading code uses -- fn 0 sinf, 1 cosf, 2 expf, 3 acosf (csrc/hrt_libm.h)
 * -- or, fn 4, the incidence angle of src/compute_paths.c:281-283 for dot(n, d) = in[i]
 * ((float)acos((double)x) folded to [0, pi/2]); fn 5 / 6 the sine / cosine of the fused
 * hrt_sincosf and fn 7 hrt_cosf_nb (the branch-free forms the shade kernel calls).  Tests compare
 * the result with the host libm. */
int hrt_selftest_math(int device, int fn, const float *in, float *out, uint64_t n);

/* Diagnostic counters of the trace kernel; all zero unless the library was built with
 * `make STATS=1`.  out48 = [3 kinds][16]: kind 0 primary traces of launch 0, 1 primary traces
 * of later launches, 2 shadow traces; columns: 0 wave-traces, 1 packets usable as a whole,
 * 2 candidate triangles after packet culling, 3 / 4 / 5 staged bodies reaching stage 2, stage 3,
 * the exact divisions, 6 (sub-)packets walked [flat: heavy packets], 7 packet-culling rounds of near
 * leaves [flat: candidates of heavy packets], 8 sphere-node rounds, 9 plane-node rounds, 10 plane
 * leaves judged, 11 of them with flagged triangles, 12 flagged triangles, 13-15 unused. */
int hrt_debug_kernel_stats(int device, uint64_t *out48, int reset);

#ifdef __cplusplus
}
#endif
#endif /* HRT_DEVICE_H */
