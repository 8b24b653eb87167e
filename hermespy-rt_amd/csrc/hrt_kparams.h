/* hrt_kparams.h -- internal contract between the host C code and the HIP shim.
 * Plain C structs; passed to the kernels by value. */
#ifndef HRT_KPARAMS_H
#define HRT_KPARAMS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HRT_NO_HIT 0xFFFFFFFFu
#ifndef HRT_ERR_FUSE_TIMEOUT          /* (public: include/hrt_device.h carries the same three values) */
#define HRT_ERR_FUSE_TIMEOUT 0x100u   /* error word of a trace (counts[nb + 1]): a fused launch timed out waiting for its
                                       * prefix (GPU shared with other fused kernels); the step is void, redo it unfused */
#define HRT_ERR_CHAIN_TIMEOUT 0x200u  /* ... the kernel that runs launches 1 .. nb as one (hrt_chain_kernel) found its grid not
                                       * resident, or timed out in a grid barrier; the step is void, redo it launch by launch */
#define HRT_ERR_VOID (HRT_ERR_FUSE_TIMEOUT | HRT_ERR_CHAIN_TIMEOUT)
#endif
#define HRT_NUM_MATERIALS 17
#define HRT_TRI_FLOATS 20  /* v1(3) e1(3) e2(3) n(3) E_d c_uv c_w (culling tolerances) |e1| |e2| |e2-e1| |e1xe2| mesh_id(u32) */
#define HRT_MAT_FLOATS 16  /* 12 MaterialPrecomputed fields, s, s1_alpha, pad(2) */
#define HRT_MESH_FLOATS 4  /* velocity(3), material_index(u32) */
#define HRT_BLOCK 256
/* a super-chunk = 2^HRT_SUPER_SHIFT chunks: the trace kernel adds every chunk's survivor count to
 * its super-chunk's (one atomic per chunk: fewer chunks per super-chunk = less contention), the
 * shade kernel sums the earlier super-chunks and the earlier chunks of its own */
#ifndef HRT_SUPER_SHIFT
#define HRT_SUPER_SHIFT 5
#endif
#define HRT_SUPER_CHUNKS (1u << HRT_SUPER_SHIFT)
#define HRT_TRACE_GRID 8192   /* persistent workgroups of the trace kernel */
#define HRT_SHADE_GRID 16384
/* triangle tables up to this many bytes are staged in LDS (160 KiB per CU on gfx950, minus
 * the material/endpoint tables); larger scenes read the table through the scalar cache. */
#define HRT_LDS_TRI_BYTES_MAX (40u * 1024u)   /* larger tables are read from global memory / L2: staged in
                                              * LDS they would leave fewer than 4 workgroups per CU, and
                                              * occupancy is worth more than LDS latency (T = 1104: 7.3 ms
                                              * staged, 2.4 ms from L2); env HRT_LDS_TRI_BYTES_MAX overrides,
                                              * up to 144 KiB */

/* ---- acceleration structure (host: csrc/host/accel.c; proofs: DESIGN.md section 9) ---- */
#define HRT_NODE_FLOATS 8           /* a sphere node: c.xyz, R, Lambda, -, -, -; a cone node: nu.xyz,
                                     * sin(beta+g), cos(beta+g); a guard record: p1.xyz, l, n.xyz, qs */
#define HRT_ACCEL_MAX_LEVELS 3      /* 64-ary levels above the leaves: up to 64^4 = 16.7 M triangles */
#define HRT_ACCEL_BIG 524288u        /* tables with more triangles get inner levels + the plane tree ... */
#define HRT_ACCEL_SPARSE 0.05       /* ... if the median leaf radius is below this fraction of the scene's */
#define HRT_GUARD_SF 4.0            /* safety factor on the reference's noise bound 1e-5 l (S + l) */
#define HRT_GUARD_MU 0.0625         /* big tables: a sphere is "far" when missed by mu * S and Lambda / 2 */

typedef struct {
    const uint32_t *orig;           /* [T] table row -> index in the reference's loop order */
    const float *tg;                /* [T][2]: qs, longest edge */
    const float *leaf;              /* [num_leaf][HRT_NODE_FLOATS] */
    uint32_t num_leaf;
    uint32_t big;                   /* inner levels + plane tree present */
    uint32_t num_levels;
    uint32_t node_count[HRT_ACCEL_MAX_LEVELS];
    const float *node[HRT_ACCEL_MAX_LEVELS];
    uint32_t pl_levels;
    uint32_t pl_count[HRT_ACCEL_MAX_LEVELS];
    const float *pl_node[HRT_ACCEL_MAX_LEVELS];
    const uint32_t *pl_index;
    const float *pl_rec;
    /* fine leaves: one sphere record (c.xyz, R, Lambda) per HRT_FINE_ROWS consecutive table rows, scanned
     * FLAT (64 at a time) by closest_hit_fine, with the plane tree as the guard: tables of more than
     * HRT_FINE_MIN_TRI triangles without the big-table trees; NULL otherwise */
    const float *fine;
    uint32_t num_fine;
} hrt_kaccel;
#define HRT_FINE_ROWS 16u
#define HRT_FINE_MIN_TRI 1024u
#define HRT_WIDE_COS 0.995f    /* half-angle 5.7 deg.  Measured (ms per step, 1 M rays; city 25 k / 100 k, room 24 k):
                                * 0.5 15.4 / 56.7 / 22.2, 0.95 8.4 / 31.7 / 10.9, 0.99 7.5 / 27.8 / 11.0, 0.995 7.2 / 26.4,
                                * 0.999 7.0 / 25.0 / 11.5, every packet 10.7 / 38.2 / 16.6 */
#define HRT_WIDE_COS_BIG 0.998f      /* ... on tables of HRT_WIDE_COS_BIG_TRI triangles or more (final code, city 25 k / 100 k / 199 k, room
                                       * 24 k: 0.995 6.45 / 22.1 / 32.1 / 11.2, 0.998 6.02 / 21.1 / 28.7 / 10.9, 0.999 5.98 / 20.4 / 29.0 / 11.3; the
                                       * smaller tables lose 2 % at 0.998) */
#define HRT_WIDE_COS_BIG_TRI 4096u   /* (with 8 192 workgroups in the wide kernel: 6 012 triangles 3.73 -> 3.67 ms at 0.998, 1 212: 1.61 -> 1.63) */

/* ---- per-RX direction tables for the shadow rays (host: problem.c; kernels: closest_hit_packet) ----
 * All shadow rays of a trace kind converge on one RX, so which triangles can possibly be met is a
 * function of the DIRECTION seen from that RX: the directions are binned on a cube map (6 faces x
 * n x n cells), and per (RX, cell) the list of triangles that packet culling cannot reject for the
 * whole cell (widened by the largest packet half-angle served, HRT_RXT_SIN_AQ) is computed once
 * per problem, on the device, with the very same test.  A shadow packet then culls only its cell's
 * list -- typically one round instead of T / 64. */
#define HRT_RXT_N 48                /* cells per cube-face edge */
#define HRT_RXT_BINS (6 * HRT_RXT_N * HRT_RXT_N)
#define HRT_RXT_SIN_AQ 0.05233596f  /* sin(3 deg): packets wider than this use the whole table */
#define HRT_SORT_MIN_TRI 1024u      /* tables beyond this re-sort the live list between bounces by default */
#define HRT_RXT_MAX_TRI 1024u       /* tables are built for scenes up to this many triangles */
typedef struct {
    uint32_t enabled;
    uint32_t num_txt;               /* tables of the TXs follow those of the RXs (the launch set leaves a TX
                                     * as the shadow rays arrive at an RX): num_tx, or 0 = none */
    float cx, cy, cz, region_r;     /* the ball every ray origin of the scene lies in */
    const float *ro_bin;            /* [num_rx + num_txt] line-point radius the lists were built for */
    const uint32_t *off;            /* [(num_rx + num_txt) * HRT_RXT_BINS + 1] */
    const uint16_t *idx;            /* table rows */
    /* tables of at most 64 triangles: instead of lists, ONE 64-bit candidate mask per (apex, cell),
     * built for the cell alone (no packet widening) -- a ray looks up its own cell, the wave ORs
     * the masks of its lanes and walks the union: no origin ball, no cone, no culling round.
     * [(num_rx + num_tx) * HRT_RXT_BINS], or NULL (then `enabled` says whether there are lists) */
    const unsigned long long *cell_mask;
} hrt_krxt;

/* ---- patch tables: candidate masks keyed by WHERE A RAY STARTS (host: problem.c patch_build; kernels:
 * closest_hit_patch, hrt_patch_build_kernel) ----
 * Every ray of launch b >= 1 starts on the triangle it just hit.  Each triangle carries a (nu x nv) grid
 * of cells ("patches") in its (e1, e2) basis; per (apex, patch) ONE candidate mask of the whole table,
 * built once per problem on the device by the packet test itself (packet_culls) with the packet
 * { origins in the patch's ball, lines that meet ball(apex, ro) }:
 *   apex k < num_rx            shadow rays towards RX k (they arrive at the apex);
 *   apex num_rx + tx           the bounce rays of launch 1: they left TX tx and were mirrored by the patch's
 *                              triangle, so they leave the IMAGE of the TX in that triangle's plane.
 * A lane finds its patch from its own origin (and is only served when that origin provably lies in the
 * patch's ball, and -- image apexes -- its line provably meets the apex ball), the wave ORs the masks of
 * its lanes and walks the union through the staged test: no origin ball, no cone, no culling round, and a
 * wave whose rays left different surfaces pays for the union of a few small sets, not for the table.
 * Tables of 65 .. 256 triangles (4 mask words per entry). */
#define HRT_PATCH_WORDS 4u
#define HRT_PATCH_MAX_TRI (64u * HRT_PATCH_WORDS)
#define HRT_PATCH_MARGIN 0.0625f    /* a cell's ball covers this fraction of a cell beyond its outline */
#define HRT_PATCH_ACCEPT 0.03125f   /* an origin up to this fraction of a cell outside the grid is clamped into it */
typedef struct {
    const unsigned long long *mask; /* [num_apex][num_patch][HRT_PATCH_WORDS], or NULL (no tables) */
    const float *pdef;              /* [T][8]: g1 * nu (xyz), bits(base) | g2 * nv (xyz), bits(nu | nv << 16) */
    const unsigned long long *txcell; /* [num_tx][HRT_RXT_BINS][HRT_PATCH_WORDS]: the launch rays leave a TX exactly, so the
                                     * candidates of a ray are a function of the cube-map cell of its direction (built for
                                     * origin = the TX, lines through it, the cell's cone): launch 0 by per-lane lookups
                                     * too -- 2.4 candidates per wave on C3 against 1.9 of the packet test, without its
                                     * ball, cone and culling round; or NULL */
    uint32_t num_patch;
    uint32_t num_img;               /* image-apex tables present for this many TXs (0 or num_tx) */
    float hmax;                     /* served origins lie within this distance of their triangle's plane */
    float ro_rx, ro_img;            /* line-point radii the tables were built for */
} hrt_kpatch;

/* ---- re-sorting of the live list between bounces ----
 * After a bounce the rays of a wave may have left different surfaces in different directions: the
 * wave is then a wide packet and culls nothing.  With `enabled`, the shade kernel writes the
 * survivors of bounce b into a scratch block, they are sorted by (TX, cell of the new origin, bin of
 * the new direction) -- stable, so launch-order coherence survives inside a key -- and written in
 * that order into hit block b, which every later consumer (next launch, record export) reads.
 * Results do not depend on the order; only speed does. */
typedef struct {
    uint32_t enabled;
    uint32_t key_bits;              /* bits of the key that are in use (sort passes) */
    uint32_t tx_shift;              /* key = tx | coarse cell | direction bin | fine cell (most to least significant) */
    uint32_t dir_res, dir_bits;     /* direction bins: 6 faces x dir_res^2, in dir_bits bits */
    uint32_t bits[3], nfine;        /* cell bits per axis (15 in all, interleaved); how many of the code's low bits sort behind the direction */
    float lo[3], inv_cell[3];       /* cell = (o - lo) * inv_cell */
    uint64_t off_scratch;           /* workspace: a hit-block-sized scratch */
    uint64_t off_keys;              /* workspace: 4 arrays of cap u32 (keys in/out, index in/out) */
    uint64_t off_tmp, tmp_bytes;    /* workspace: digit histograms of the radix sort ([256][tiles] + 256 totals) */
} hrt_ksort;

/* developer / test switches that reach the launch shims (csrc/host/tune.c: HRT_TUNE, read once per problem) */
typedef struct {
    int32_t variant;               /* intersection loop: HRT_TRACE_VARIANT_DEFAULT = auto */
    uint64_t lds_tri_bytes_max;    /* tables up to this many bytes are staged in LDS */
    uint32_t trace_grid, shade_grid, wide_grid;
    uint32_t los_big_min_tri, fuse_staged_max_tri, shade_global_normals;
    uint32_t lb_max_polls;         /* fused launches: polls of the prefix words before the step is declared void (~10 ms) */
} hrt_ktune;
#define HRT_TRACE_VARIANT_DEFAULT 7

typedef struct {
    /* scene (device pointers) */
    const float *tri;     /* [num_tri][HRT_TRI_FLOATS] in (mesh, face) order */
    const float *mesh;    /* [num_mesh][HRT_MESH_FLOATS] */
    const float *mat;     /* [17][HRT_MAT_FLOATS] */
    uint32_t num_tri, num_mesh;
    hrt_kaccel acc;
    hrt_krxt rxt;
    hrt_kpatch patch;
    hrt_ksort sort;
    hrt_ktune tune;
    /* endpoints (device pointers, [n][3]) */
    const float *rx_pos, *tx_pos, *rx_vel, *tx_vel;
    uint32_t num_rx, num_tx;
    /* src/compute_paths.c:483-488 */
    float fsl_mult;       /* 4 pi f / c */
    float dop_mult;       /* f / c */
    /* shard */
    const float *dirs;    /* [num_local][3] */
    const uint32_t *order;/* [num_local] coherent launch order (local ray of position i), or NULL */
    uint32_t num_local;
    uint32_t num_bounces;
    uint32_t n0;          /* num_tx * num_local */
    uint32_t dirs_in_launch_order;   /* dirs[i] is the direction of launch position i (not of ray i) */
    /* workspace (see include/hrt_device.h) */
    uint8_t *ws;
    uint64_t cap;
    uint64_t off_counts, off_los, off_hits, hit_block_bytes, off_recs, rec_block_bytes,
        off_masks, off_chunk_cnt, off_super_cnt, off_res;
    uint32_t num_super;   /* super-chunks per bounce */
    /* status words of the fused kernels' stable compaction (hrt_kernels.hip, "counted sums"), zeroed
     * with the counts: per launch b, from off_lb + b * lb_stride * 4: lb_chunks u32 (one per
     * 256-entry chunk), lb_chunks / 64 u32 (groups), 128 u32 (supergroups) */
    uint64_t off_lb;
    uint32_t lb_stride, lb_chunks;
    uint32_t los_blocks;  /* fused launch 0: the last los_blocks workgroups of the grid do the LoS pass (0: own kernel) */
    uint32_t phase_bounce;   /* -DHRT_PHASE_STATS builds: the launch whose workgroups record their time stamps */
    uint32_t fuse;        /* HRT_FUSE_*: which launches run as ONE kernel (trace + shade + compaction) */
    uint32_t cnt_stride;  /* HRT_CNT_STRIDE(num_bounces): bytes between the parts of the counter block */
    uint32_t *host_flag;  /* two words in page-locked host memory (device address): [0] set to 1 when a fused launch gives
                           * up waiting (HRT_ERR_FUSE_TIMEOUT), [1] when the chain kernel does (HRT_ERR_CHAIN_TIMEOUT) --
                           * the host sees them without synchronising; or NULL */
    uint32_t records_done; /* set by the shade shim: hrt_records_kernel wrote this launch's records (patch tables) */
    /* queue of the packets that are too wide to cull (big tables, hrt_wide_kernel): wide_cap entries of
     * 8 bytes at off_wide_q, 64 keys of 8 bytes per entry at off_wide_key; the per-launch entry counts are
     * the u32 at off_counts + 2 cnt_stride + 4 b (zeroed with the counts).  wide_cap 0: no queue */
    uint64_t off_wide_q, off_wide_key;
    uint32_t wide_cap;
    float wide_cos;          /* fine walk: packets whose cone is wider than this cosine go to the wide kernels */
    const uint32_t *wide_inv;   /* [T] index in the reference's loop order -> table row (the inverse of acc.orig) */
    uint32_t cnt_per_wave;   /* survivor counts at off_chunk_cnt: 1 = one word per WAVE of a chunk (the fine walk, whose
                              * waves do not wait for each other), 0 = one per chunk; set by the launch shims */
} hrt_kparams;

/* the counter block at off_counts (zeroed at the start of every trace), four parts of cnt_stride bytes
 * (hrt_cnt_stride(num_bounces): 256 up to 62 bounces): counts[nb + 2] | one work-unit counter per launch
 * (trees) | one wide-queue entry count per launch | LoS words {max of ~distance bits, waves done} per
 * (rx, tx) pair, up to 32 pairs (256 bytes) */
#define HRT_CNT_STRIDE(nb) ((((uint64_t)(nb) + 2u) * 4u + 255u) & ~255ull)
#define HRT_CNT_BYTES(nb) (3u * HRT_CNT_STRIDE(nb) + 256u)
#define HRT_WIDE_SLICE 1024u  /* table rows of one (packet, slice) item of hrt_wide_kernel: 64 fine spheres */

/* hrt_kparams.fuse */
#define HRT_FUSE_LAUNCH0 1u   /* launch 0 (no shadow rays, state generated in registers) */
#define HRT_FUSE_BOUNCES 2u   /* launches 1..num_bounces too (tables of one culling block) */
#define HRT_FUSE_MAX_TRI 64u  /* default: whole bounces are fused on tables of at most this many triangles */

/* ---- the shim (hrt_kernels.hip).  All return 0 or a positive hipError_t. ---- */
int hrt_hip_device_count(int *n);
int hrt_hip_set_device(int dev);
int hrt_hip_malloc(void **p, uint64_t bytes);
int hrt_hip_free(void *p);
int hrt_hip_host_malloc(void **p, uint64_t bytes);   /* page-locked host memory */
int hrt_hip_host_free(void *p);
int hrt_hip_host_malloc_mapped(void **host, void **dev, uint64_t bytes);
int hrt_hip_h2d(void *dst, const void *src, uint64_t bytes);
int hrt_hip_d2h(void *dst, const void *src, uint64_t bytes);
int hrt_hip_memset_async(void *dst, int value, uint64_t bytes, void *stream);
int hrt_hip_stream_sync(void *stream);
int hrt_hip_d2h_async(void *dst, const void *src, uint64_t bytes, void *stream);
int hrt_hip_stream_create(void **stream);
int hrt_hip_stream_destroy(void *stream);
int hrt_hip_mem_info(uint64_t *free_b, uint64_t *total_b);
int hrt_hip_launch_los(const hrt_kparams *P, void *stream);
int hrt_hip_launch_trace(const hrt_kparams *P, uint32_t bounce, void *stream);
int hrt_hip_launch_shade(const hrt_kparams *P, uint32_t bounce, void *stream);

int hrt_hip_launch_fused(const hrt_kparams *P, uint32_t bounce, void *stream);   /* -1: not fusable, nothing launched */
int hrt_hip_launch_records(const hrt_kparams *P, uint32_t bounce, void *stream); /* -1: not applicable, nothing launched */
int hrt_hip_launch_chain(const hrt_kparams *P, uint32_t b0, void *stream);        /* launches b0 .. nb as one kernel; -1: not applicable */
int hrt_hip_event_create_sync(void **ev);
int hrt_hip_stream_wait_event(void *stream, void *ev);
int hrt_hip_launch_dirs(uint64_t num_paths, uint32_t rank, uint32_t count, uint32_t chunk,
                        uint64_t num_local, float *d_dirs, uint32_t *d_fix_count,
                        uint32_t *d_fix_list, uint32_t fix_cap, void *stream);
int hrt_hip_rxt_build(const float *d_tri, uint32_t num_tri, const float *d_rx_pos, uint32_t num_rx,
                      const float *d_bin_dir4, const float *d_bin_cs2, const float *d_ro_bin,
                      float cx, float cy, float cz, float region_r, unsigned long long *d_masks, void *stream);
int hrt_hip_txcell_build(const float *d_tri, uint32_t num_tri, const float *d_tx_pos, uint32_t num_tx, const float *d_bin_dir4,
                         const float *d_bin_cs2, unsigned long long *d_masks, void *stream);
int hrt_hip_patch_build(const float *d_tri, uint32_t num_tri, const float *d_pdef, const uint32_t *d_patch_tri,
                        uint32_t num_patch, const float *d_apex, uint32_t num_rx, uint32_t num_img, float hball,
                        float ro_rx, float ro_img, unsigned long long *d_masks, void *stream);
uint64_t hrt_hip_sort_temp_bytes(uint64_t cap);
int hrt_hip_sort_hits(const hrt_kparams *P, uint32_t bounce, void *stream);
int hrt_hip_export_copy(const void *d_ws, const void *d_segs, uint32_t num_segs, uint64_t max_words, void *d_out, void *stream);
int hrt_hip_export_prefix(const void *d_ws, uint64_t off_counts, uint64_t off_masks, uint64_t cap, uint32_t num_bounces,
                          uint32_t num_rx, uint32_t *d_prefix, uint32_t *d_totals, void *stream);
int hrt_hip_export_compact(const void *d_ws, uint64_t off_counts, uint64_t off_masks, uint64_t off_recs, uint64_t rec_block_bytes,
                           uint64_t cap, uint32_t num_bounces, uint32_t num_rx, uint64_t max_hits, const uint32_t *d_prefix,
                           const uint32_t *d_totals, const uint64_t *d_dst_word, void *d_out, void *stream);
int hrt_hip_d2d_async(void *dst, const void *src, uint64_t bytes, void *stream);
int hrt_hip_launch_fs0(const float *d_dirs, uint64_t n, const float *tx_vel3, float mult, float *d_out, void *stream);
int hrt_hip_launch_order(const uint32_t *d_seg_start, const uint32_t *d_seg_band, uint32_t num_seg,
                         uint64_t num_paths, uint32_t rank, uint32_t count, uint32_t chunk,
                         uint32_t *d_order, void *stream);
int hrt_hip_h2d_async(void *dst, const void *src, uint64_t bytes, void *stream);
int hrt_hip_selftest_math(int fn, const float *d_in, float *d_out, uint64_t n, void *stream);
#define HRT_STATS_COLS 16
int hrt_hip_read_stats(unsigned long long *out3xCOLS, int reset);
/* events: opaque handles */
int hrt_hip_event_create(void **ev);
int hrt_hip_event_destroy(void *ev);
int hrt_hip_event_record(void *ev, void *stream);
int hrt_hip_event_elapsed_ms(void *start, void *stop, float *ms);
int hrt_hip_event_sync(void *ev);
const char *hrt_hip_error_string(int err);

#ifdef __cplusplus
}
#endif
#endif
