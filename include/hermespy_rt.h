/* hermespy_rt.h -- the drop-in C ABI of the compute_paths hot path.
 *
 * libhermespy_rt_amd.so exports the three entry points the reference's callers bind
 * (its pybind11 module, compute_paths_pybind11.cpp:155-170, and its C demo, test/test.c:62-72):
 *
 *     compute_paths   replaces  inc/compute_paths.h:59-74  (impl src/compute_paths.c:419-757)
 *     scene_load      replaces  inc/scene.h:105            (impl src/scene.c:36-83)
 *     scene_save      replaces  inc/scene.h:95             (impl src/scene.c:7-34)
 *
 * with the same names, argument meaning, ownership and error behaviour, and the structs
 * below are byte-compatible with inc/vec3.h:6-8, inc/ray.h:6-9, inc/scene.h:10-32 and
 * inc/compute_paths.h:13-30.  A program compiled against the reference headers can be
 * re-linked against this library unchanged; a program may also include this header instead.
 *
 * What is different behind the boundary: the ray launch / triangle intersection / specular
 * bounce / scatter-to-RX loop runs as hand-written HIP kernels on an MI355X (gfx950).  There
 * is no CPU implementation in this library: without a usable HIP device compute_paths()
 * reports the HIP error on stderr and exits with status 70 (the reference's own "cannot
 * continue" status, src/compute_paths.c:504), and hrt_compute_paths_ex() returns the error.
 *
 * Plain C, no torch / HIP types in any signature.
 */
#ifndef HERMESPY_RT_H
#define HERMESPY_RT_H

#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- types (layout-identical to the reference headers) ---- */

typedef struct { float x, y, z; } Vec3;            /* inc/vec3.h:6-8   (12 bytes) */
typedef struct { Vec3 o, d; } Ray;                 /* inc/ray.h:6-9    (24 bytes) */

typedef struct {                                   /* inc/scene.h:10-27 */
    uint32_t num_vertices;
    Vec3 *vs;                 /* [num_vertices] */
    uint32_t num_triangles;
    uint32_t *is;             /* [num_triangles * 3] vertex indices */
    uint32_t material_index;  /* 0..16, ITU-R P.2040-3 table 3 row */
    Vec3 velocity;
    Vec3 *ns;                 /* [num_triangles] unit normals; NULL after scene_load.  As in
                                 the reference (src/compute_paths.c:212) compute_paths()
                                 malloc()s it and leaves it for free_scene(); like the
                                 reference it OVERWRITES the old pointer without freeing it
                                 (a hand-built Mesh may leave `ns` uninitialised): a caller
                                 that calls compute_paths() repeatedly on one Scene frees
                                 `ns` between the calls, or reloads the scene. */
} Mesh;

typedef struct { uint32_t num_meshes; Mesh *meshes; } Scene;   /* inc/scene.h:29-32 */

typedef struct {                                   /* inc/compute_paths.h:13-23 */
    uint32_t num_rays;
    Vec3 *directions_rx;      /* (num_rx, num_tx, num_rays) */
    Vec3 *directions_tx;      /* LoS: (num_rx, num_tx).  Scatter: NEVER written (as in the
                                 reference; callers allocate as little as num_rays Vec3) */
    float *a_te_re, *a_te_im, *a_tm_re, *a_tm_im;   /* (num_rx, num_tx, num_rays) */
    float *tau;               /* s */
    float *freq_shift;        /* Hz */
} ChannelInfo;

typedef struct {                                   /* inc/compute_paths.h:26-30 */
    uint32_t num_bounces, num_rays;
    Ray *rays;                /* scatter: >= num_tx*(num_bounces+1)*num_rays entries */
    uint8_t *rays_active;     /* scatter: >= (num_tx*num_bounces+1)*(num_rays/8+1) bytes */
} RaysInfo;

/* inc/scene.h:72-86 (static inline there too; everything the library stores in a Scene is
 * malloc()-compatible) */
static inline void free_mesh(Mesh *mesh) { free(mesh->vs); free(mesh->is); free(mesh->ns); }
static inline void free_scene(Scene *scene)
{
    for (uint32_t i = 0; i < scene->num_meshes; i++) free_mesh(&scene->meshes[i]);
    free(scene->meshes);
}

/* ---- the reference's entry points ---- */

/* Read a .hrt file.  Errors: perror + exit(8), as src/scene.c:36-83. */
Scene scene_load(const char *filepath);

/* Write a .hrt file (normals are not stored).  Errors: perror + exit(8). */
void scene_save(Scene *scene, const char *filepath);

/* Trace.  All out-arrays are caller-allocated; only the slots the reference writes are
 * written (dead rays' slots, blocked records' directions/freq_shift and the scatter
 * directions_tx stay untouched).  Scatter arrays are indexed
 * ((rx*num_tx + tx)*num_bounces + bounce)*num_rays + ray.  Blocking; not re-entrant. */
void compute_paths(Scene *scene, Vec3 *rx_pos, Vec3 *tx_pos, Vec3 *rx_vel, Vec3 *tx_vel,
                   float carrier_frequency_GHz, size_t num_rx, size_t num_tx, size_t num_rays,
                   size_t num_bounces, ChannelInfo *chanInfo_los, RaysInfo *raysInfo_los,
                   ChannelInfo *chanInfo_scat, RaysInfo *raysInfo_scat);

/* ---- additions (not in the reference) ---- */

/* Work counters of one call. */
typedef struct {
    uint64_t live[34];        /* live[b] = rays entering bounce b; live[num_bounces] = hits of
                                 the last bounce (valid for num_bounces <= 32) */
    uint64_t records;         /* scatter records written (hit x rx, blocked ones included) */
    uint64_t records_unblocked;
    uint64_t tests;           /* algorithmic ray-triangle tests of the brute-force reference:
                                 nrx*ntx*T + sum_b T*(live[b] + nrx*hits[b]) */
    double t_setup_s, t_launch_dirs_s, t_device_s, t_readback_s, t_total_s;   /* device / readback: the
                                 slowest device's */
    int device;               /* the first device used */
    int num_devices;          /* devices the batches were dealt to (HRT_DEVICES) */
    uint32_t num_batches;     /* round-robin shards of the launch set the call was cut into */
    /* per device d < num_devices (HRT_DEVICES order; at most 16): its id, the batches it took, the
     * time its thread spent waiting for its kernels and in readback + dense scatter -- on a node
     * with several GPUs these show the balance of the call */
    int dev_id[16];
    uint32_t dev_batches[16];
    double dev_t_device_s[16], dev_t_readback_s[16];
} hrt_stats;

/* Same as compute_paths() but returns 0 / a negative HRT_E_* code instead of exiting, takes
 * optional NULL for raysInfo_los / raysInfo_scat (skips their fill and transfer), and
 * reports counters.  `stats` may be NULL. */
int hrt_compute_paths_ex(Scene *scene, const Vec3 *rx_pos, const Vec3 *tx_pos,
                         const Vec3 *rx_vel, const Vec3 *tx_vel, float carrier_frequency_GHz,
                         size_t num_rx, size_t num_tx, size_t num_rays, size_t num_bounces,
                         ChannelInfo *chanInfo_los, RaysInfo *raysInfo_los,
                         ChannelInfo *chanInfo_scat, RaysInfo *raysInfo_scat, hrt_stats *stats);

/* hrt_compute_paths_ex for callers whose amplitudes are COMPLEX arrays (numpy complex64, C99 float
 * _Complex): a_te_re / a_te_im (a_tm_re / a_tm_im) of both ChannelInfo point at the real and the
 * imaginary part of element 0 of an interleaved array, element i being at [2 i] of each pointer.
 * The dense writer fills the complex arrays in place (the reference's planes would have to be
 * interleaved by the binding afterwards: compute_paths_pybind11.cpp:44-97 does).  Everything else
 * -- arguments, layout quirks, errors -- as hrt_compute_paths_ex. */
int hrt_compute_paths_interleaved(Scene *scene, const Vec3 *rx_pos, const Vec3 *tx_pos,
                                  const Vec3 *rx_vel, const Vec3 *tx_vel, float carrier_frequency_GHz,
                                  size_t num_rx, size_t num_tx, size_t num_rays, size_t num_bounces,
                                  ChannelInfo *chanInfo_los, RaysInfo *raysInfo_los,
                                  ChannelInfo *chanInfo_scat, RaysInfo *raysInfo_scat, hrt_stats *stats);

#define HRT_OK 0
#define HRT_E_INVALID (-1)   /* bad argument (zero count, material_index > 16, ...) */
#define HRT_E_NOMEM (-2)     /* host allocation failed */
#define HRT_E_HIP (-3)       /* HIP runtime error; text via hrt_last_error() */
#define HRT_E_CAPACITY (-4)  /* problem does not fit the device / > 71.5 M rays in one shard */

/* compute_paths() with the result as ONE list of path records instead of dense arrays: same inputs,
 * same tracing, the same values bit for bit (src/compute_paths.c:419-757), but only the records
 * that exist -- the dense [rx][tx][bounce][path] form is > 95 % unwritten slots.  Entry n is the
 * record the reference would write at scat.*[((rx[n]*num_tx + tx[n])*num_bounces + bounce[n])*
 * num_rays + path[n]]; order: by bounce, then rx, then the device's live-list order.
 * freq_shift = launch Doppler term of the ray minus the record's (the dense array's value for one
 * TX; the reference's dense fill is undefined for more).  All arrays are malloc'ed by the
 * call and released by hrt_path_list_free(). */
typedef struct {
    uint64_t num;
    uint32_t num_rx, num_tx;
    uint32_t *rx, *tx, *bounce;      /* [num] */
    uint64_t *path;                  /* [num] */
    float *a_te_re, *a_te_im, *a_tm_re, *a_tm_im, *tau;   /* [num]; a blocked record has zeros */
    Vec3 *direction_rx;              /* [num]; undefined for blocked records */
    float *freq_shift;               /* [num]; undefined for blocked records */
    uint8_t *unblocked;              /* [num] */
    uint32_t *mesh, *face;           /* [num] the triangle the ray left towards the RX */
    float *los;                      /* [num_rx*num_tx][8]: u32 status (0 coincident, 1 blocked,
                                      * 2 clear), a, tau, dir_tx xyz, freq_shift, - (HRT_LOS_*) */
} hrt_path_list;

/* include_blocked = 0 drops the blocked records (the reference writes zeros there). */
int hrt_compute_paths_list(Scene *scene, const Vec3 *rx_positions, const Vec3 *tx_positions,
                           const Vec3 *rx_velocities, const Vec3 *tx_velocities,
                           float carrier_frequency_GHz, size_t num_rx, size_t num_tx, size_t num_rays,
                           size_t num_bounces, int include_blocked, hrt_path_list *out,
                           hrt_stats *stats);
void hrt_path_list_free(hrt_path_list *list);

/* Human-readable description of the last error on this thread ("" if none). */
const char *hrt_last_error(void);

/* "hermespy-rt_amd <version> (gfx950)" */
const char *hrt_version(void);

/* Between calls compute_paths keeps: (with HRT_HOST_LAUNCH=1) the launch-direction table and launch
 * order of the last num_rays; the device workspace and page-locked staging of the last call; and
 * the calling thread's helper threads of the dense writer, parked (csrc/host/compute_paths.c).
 * This releases all three.  Environment: HRT_NO_CACHE=1 keeps no buffers. */
void hrt_cache_clear(void);

#ifdef __cplusplus
}
#endif
#endif /* HERMESPY_RT_H */
