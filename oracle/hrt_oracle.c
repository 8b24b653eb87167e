/* hrt_oracle.c -- CPU restatement of the reference's compute_paths() hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle for the HIP product path.
 * Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may build, load or
 * call it, and only as the checker -- it is never linked into, imported by, or executed
 * from hermespy-rt_amd/ (the product fails loudly when its HIP library is missing).
 *
 * Parity status: PINNED.  The reference holds no golden values of its own (its two tests
 * assert shapes only, test/test.py:61-87), so this restatement is pinned against outputs of
 * the reference itself: oracle/Makefile builds oracle/_ref/libhrt_ref.so from the sources
 * where they lie under /root/reference, tests/golden/make_golden.py records its outputs as
 * fixtures, and tests/test_oracle_*.py require bit-identity of EVERY output array (written
 * slots, unwritten slots, RaysInfo snapshots, active masks) on all four bundled scenes.
 *
 * Structure (deliberately not the reference's): the scene arrives flattened (one triangle
 * table in (mesh, face) order with edges pre-subtracted), rays are advanced bounce by bounce
 * over an explicit live list (wavefront form -- the same decomposition the HIP kernels use),
 * the per-ray work of one (bounce, tx) is data-parallel (OpenMP), and the few order-dependent
 * side effects of the reference (freq_shift read-modify-writes, snapshots) are replayed
 * serially afterwards in the reference's order.  Extra outputs the reference does not have:
 * hit triangle per (bounce, ray), live counts per bounce, algorithmic test count.
 *
 * Each function cites the reference lines (relative to /root/reference) it restates.
 *
 * Build: gcc -O3 -ffp-contract=off -fopenmp   (no -march=native/-mfma: contraction changes
 * result bits).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* src/compute_paths.c:18-19 */
#define HRT_PI_F 3.14159265358979323846f
#define HRT_C_F 299792458.0f
/* FLT_EPSILON = 2^-23; 1+2^-23 is the float after 1 */
#define HRT_EPS 1.1920928955078125e-07f
#define HRT_ONE_PLUS_EPS 1.00000011920928955078125f

#define HRT_NUM_MATERIALS 17
#define HRT_NO_HIT 0xFFFFFFFFu

/* ---- ITU-R P.2040-3 table 3 parameters + scattering parameters, per material index.
 * Data restated from src/materials.c:3-89 as a flat numeric table
 * {a, b, c, d, s, s1_alpha}; the names and the unused s1/s2/s3/s3_alpha are omitted. */
static const float k_mat[HRT_NUM_MATERIALS][6] = {
    /* 0 air               */ {1.f, 0.f, 0.f, 0.001f, 0.1f, 2.f},
    /* 1 concrete          */ {5.24f, 0.f, 0.0462f, 0.7822f, 0.5f, 4.f},
    /* 2 brick             */ {3.91f, 0.f, 0.0238f, 0.16f, 0.4f, 3.f},
    /* 3 plasterboard      */ {2.73f, 0.f, 0.0085f, 0.9395f, 0.3f, 3.f},
    /* 4 wood              */ {1.99f, 0.f, 0.0047f, 1.0718f, 0.2f, 2.f},
    /* 5 glass (1)         */ {6.31f, 0.f, 0.0036f, 1.3394f, 0.3f, 3.f},
    /* 6 glass (2)         */ {5.79f, 0.f, 0.0004f, 1.658f, 0.3f, 3.f},
    /* 7 ceiling board (1) */ {1.48f, 0.f, 0.0011f, 1.0750f, 0.2f, 2.f},
    /* 8 ceiling board (2) */ {1.52f, 0.f, 0.0029f, 1.029f, 0.2f, 2.f},
    /* 9 chipboard         */ {2.58f, 0.f, 0.0217f, 0.7800f, 0.4f, 3.f},
    /* 10 plywood          */ {2.71f, 0.f, 0.33f, 0.f, 0.3f, 3.f},
    /* 11 marble           */ {7.074f, 0.f, 0.0055f, 0.9262f, 0.3f, 3.f},
    /* 12 floorboard       */ {3.66f, 0.f, 0.0044f, 1.3515f, 0.3f, 3.f},
    /* 13 metal            */ {1.f, 0.f, 10000000.f, 0.f, 0.f, 1.f},
    /* 14 very dry ground  */ {3.f, 0.f, 0.00015f, 2.52f, 0.4f, 4.f},
    /* 15 medium dry ground*/ {15.f, -0.1f, 0.035f, 1.63f, 0.5f, 4.f},
    /* 16 wet ground       */ {30.f, -0.4f, 0.15f, 1.30f, 0.5f, 4.f},
};

typedef struct { float x, y, z; } v3;

/* inc/vec3.h:10-43 -- operand order is part of the contract (bit-exactness) */
static inline v3 sub3(v3 a, v3 b) { v3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
static inline v3 add3(v3 a, v3 b) { v3 r = {a.x + b.x, a.y + b.y, a.z + b.z}; return r; }
static inline v3 mul3(v3 a, float s) { v3 r = {a.x * s, a.y * s, a.z * s}; return r; }
static inline float dot3(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3 cross3(v3 a, v3 b)
{
    v3 r = {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
    return r;
}
static inline v3 unit3(v3 a)
{
    float n = sqrtf((a.x * a.x + a.y * a.y) + a.z * a.z);
    v3 r = {a.x / n, a.y / n, a.z / n};
    return r;
}

/* ---- public structs (mirrored with ctypes in oracle/oracle.py) ---- */

/* Flattened scene: triangles in (mesh, face) order. */
typedef struct {
    uint32_t num_meshes;
    uint32_t num_tri;
    const float *tri_vtx;         /* [num_tri][9]  v1 v2 v3 */
    const uint32_t *tri_mesh;     /* [num_tri]     owning mesh */
    const uint32_t *mesh_material;/* [num_meshes] */
    const float *mesh_velocity;   /* [num_meshes][3] */
} hrt_oracle_scene;

/* One ChannelInfo worth of caller-allocated arrays (inc/compute_paths.h:13-23). */
typedef struct {
    float *directions_rx, *directions_tx;
    float *a_te_re, *a_te_im, *a_tm_re, *a_tm_im, *tau, *freq_shift;
} hrt_oracle_chan;

typedef struct {
    float *rays;          /* Ray = 6 floats */
    uint8_t *rays_active;
} hrt_oracle_rays;

typedef struct {
    /* subset of the path index processed: p = p_begin + k*p_stride < p_end.  The launch
     * directions always use the GLOBAL index and GLOBAL num_paths.  Outside (0, np, 1) the
     * dense freq_shift fill / RaysInfo snapshots are still done for all paths so that the
     * result equals the full run restricted to the subset. */
    uint64_t p_begin, p_end, p_stride;
    int num_threads;      /* <=0: OpenMP default */
    /* optional extras (may be NULL) */
    uint32_t *hit_tri;    /* [nb][ntx*np], HRT_NO_HIT where the ray did not hit at that bounce */
    float *hit_theta;     /* [nb][ntx*np] incidence angle of the hit */
    uint64_t *live;       /* [nb+1] rays entering bounce b (live[nb] = hits of last bounce) */
    uint64_t *tests;      /* [1] algorithmic ray-triangle tests */
    float *eta_table;     /* [17][12] MaterialPrecomputed rows, reference field order */
    float *normals;       /* [num_tri][3] */
    float *launch_dirs;   /* [np][3] */
    /* compact subset (test infrastructure for launch sets whose dense arrays do not fit the host,
     * C5 at 8 x 8 x 8 x 8M): every array indexed by the path -- scatter outputs, hit_tri, hit_theta
     * -- has extent npo = number of subset paths and slot (p - p_begin) / p_stride instead of np
     * and p; freq_shift is the record's own value (launch term minus the record's: the dense
     * array's value for one TX, without the reference's replication quirks Q9/Q10); RaysInfo
     * snapshots are not taken (scat_rays may be NULL). */
    int compact;
} hrt_oracle_opts;

/* src/compute_paths.c:125-132 field order */
typedef struct {
    float eta_re, eta_sqrt_re, eta_inv_re, eta_inv_sqrt_re;
    float eta_im, eta_sqrt_im, eta_inv_im, eta_inv_sqrt_im;
    float eta_abs, eta_abs_pow2, eta_abs_inv_sqrt;
    float r;
} mat_pre;

/* src/compute_paths.c:136-151 */
static void complex_sqrt(float re, float im, float mag, float *o_re, float *o_im)
{
    *o_re = sqrtf((re + mag) / 2.f);
    if (fabsf(im) < HRT_EPS && re >= -HRT_EPS) {
        *o_im = 0.f;
    } else {
        float s = sqrtf((mag - re) / 2.f);
        *o_im = (im < 0.f) ? -s : s;
    }
}

/* src/compute_paths.c:152-164 */
static inline void complex_div(float ar, float ai, float br, float bi, float *cr, float *ci)
{
    float den = br * br + bi * bi;
    *cr = (ar * br + ai * bi) / den;
    *ci = (ai * br - ar * bi) / den;
}

/* src/compute_paths.c:171-206 (one material) */
static void material_eta(uint32_t idx, float f_ghz, mat_pre *m)
{
    const float *p = k_mat[idx];
    m->eta_re = p[0] * powf(f_ghz, p[1]);
    m->eta_im = (p[2] * powf(f_ghz, p[3])) / (0.0556325027352135f * f_ghz);
    m->eta_abs_pow2 = m->eta_re * m->eta_re + m->eta_im * m->eta_im;
    m->eta_abs = sqrtf(m->eta_abs_pow2);
    m->eta_abs_inv_sqrt = 1.f / sqrtf(m->eta_abs);
    complex_sqrt(m->eta_re, m->eta_im, m->eta_abs, &m->eta_sqrt_re, &m->eta_sqrt_im);
    m->eta_inv_re = m->eta_re / m->eta_abs_pow2;
    m->eta_inv_im = -m->eta_im / m->eta_abs_pow2;
    complex_sqrt(m->eta_inv_re, m->eta_inv_im, 1.f / m->eta_abs, &m->eta_inv_sqrt_re,
                 &m->eta_inv_sqrt_im);
    m->r = 1.f - p[4];
}

/* Prepared triangle: v1 and the two edges (e1 = v2-v1, e2 = v3-v1, :259-260) and the unit
 * normal (:208-224).  Precomputing the edges is the same subtraction on the same inputs. */
typedef struct { v3 v1, e1, e2, n; } tri_t;

typedef struct {
    uint32_t tri;   /* flat index, HRT_NO_HIT on miss */
    float t;
    float theta;    /* valid on hit only */
} hit_t;

/* src/compute_paths.c:237-287.  Closest hit over all triangles in (mesh, face) order; the
 * strict '<' keeps the lowest index on equal distance.  theta depends only on the final
 * winner, so it is evaluated once after the loop. */
static hit_t closest_hit(const tri_t *tris, uint32_t nt, v3 o, v3 d)
{
    float best = 1e9f;
    uint32_t who = HRT_NO_HIT;
    for (uint32_t j = 0; j < nt; ++j) {
        const tri_t *T = &tris[j];
        v3 pv = cross3(d, T->e2);
        float det = dot3(T->e1, pv);
        if (det > -HRT_EPS && det < HRT_EPS) continue;
        v3 s = sub3(o, T->v1);
        float u = dot3(s, pv) / det;
        if (u < -HRT_EPS || u > HRT_ONE_PLUS_EPS) continue;
        v3 q = cross3(s, T->e1);
        float v = dot3(d, q) / det;
        float w = u + v;
        if (v < -HRT_EPS || w > HRT_ONE_PLUS_EPS) continue;
        float dist = dot3(T->e2, q) / det;
        if (dist > HRT_EPS && dist < best) { best = dist; who = j; }
    }
    hit_t h = {who, best, 0.f};
    if (who != HRT_NO_HIT) {
        /* :281-283 -- acos in double, stored to float, folded with the FLOAT pi */
        float th = (float)acos((double)dot3(tris[who].n, d));
        if ((double)th > (double)HRT_PI_F / 2.) th = HRT_PI_F - th;
        h.theta = th;
    }
    return h;
}

/* src/compute_paths.c:300-344 */
static void fresnel(const mat_pre *m, float th, float out[4])
{
    float s1 = sinf(th);
    if (m->eta_abs_inv_sqrt * s1 > 1.f - HRT_EPS) {
        out[0] = out[2] = 1.f;
        out[1] = out[3] = 0.f;
        return;
    }
    float s2 = s1 * s1;
    float c2r = sqrtf(1.f + m->eta_inv_re / m->eta_abs_pow2 * s2);
    float c2i = sqrtf(1.f - m->eta_inv_im / m->eta_abs_pow2 * s2);
    float pr = m->eta_sqrt_re * c2r - m->eta_sqrt_im * c2i;
    float pi = m->eta_sqrt_re * c2i + m->eta_sqrt_im * c2r;
    float c1 = cosf(th);
    complex_div(c1 - pr, -pi, c1 + pr, pi, &out[0], &out[1]);
    float qr = m->eta_sqrt_re * c1;
    float qi = m->eta_sqrt_im * c1;
    complex_div(qr - c2r, qi - c2i, qr + c2r, qi + c2i, &out[2], &out[3]);
    out[0] *= m->r; out[1] *= m->r; out[2] *= m->r; out[3] *= m->r;
}

/* src/compute_paths.c:359-415.  alpha is the uint8 s1_alpha promoted to int. */
static void scatter_pattern(float th_s, float th_i, uint32_t mat, float out[4])
{
    float s = k_mat[mat][4];
    int alpha = (int)k_mat[mat][5];
    float cs = cosf(th_s), ci = cosf(th_i), si = sinf(th_i);
    float dth = fabsf(th_s - th_i);
    float f = s * expf((float)(-alpha) * dth);
    float rough = 1.0f / (1.0f + (float)alpha);
    float spec = rough * cs;
    float diff = (1.0f - rough) * cs;
    float te = f * (spec + diff);
    float tm = f * (spec * ci + diff);
    float ph = (float)alpha * si * 0.1f;
    float tei = te * sinf(ph);
    float tmi = tm * sinf(ph);
    float nrm = sqrtf(te * te + tei * tei + tm * tm + tmi * tmi);
    if (nrm > 1e-6f) { te /= nrm; tei /= nrm; tm /= nrm; tmi /= nrm; }
    out[0] = te; out[1] = tei; out[2] = tm; out[3] = tmi;
}

/* src/compute_paths.c:443-451: Fibonacci sphere, float inputs, double trig. */
static v3 launch_dir(uint64_t p, uint64_t np)
{
    float k = (float)p + .5f;
    float phi = (float)acos((double)(1.f - 2.f * k / (float)np));
    float th = HRT_PI_F * (1.f + sqrtf(5.f)) * k;
    v3 d = {(float)(cos((double)th) * sin((double)phi)),
            (float)(sin((double)th) * sin((double)phi)),
            (float)cos((double)phi)};
    return d;
}

static inline v3 ld3(const float *p) { v3 r = {p[0], p[1], p[2]}; return r; }
static inline void st3(float *p, v3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }

int hrt_oracle_version(void) { return 1; }

/* cap on the threads of every parallel region (a container may show all the machine's cores
 * while owning a share of them: oracle.py passes min(16, cores)) */
void hrt_oracle_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int hrt_oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* Whole path: src/compute_paths.c:419-757.  Returns 0, or -1 on allocation failure /
 * invalid material index. */
int hrt_oracle_compute_paths(const hrt_oracle_scene *sc,
                             const float *rx_pos, const float *tx_pos,
                             const float *rx_vel, const float *tx_vel,
                             float f_ghz, size_t nrx, size_t ntx, size_t np, size_t nb,
                             hrt_oracle_chan *los, hrt_oracle_rays *los_rays,
                             hrt_oracle_chan *scat, hrt_oracle_rays *scat_rays,
                             const hrt_oracle_opts *opt)
{
    const uint32_t T = sc->num_tri;
    const size_t nq = ntx * np;
    uint64_t p_begin = 0, p_end = np, p_stride = 1;
    if (opt && opt->p_stride) { p_begin = opt->p_begin; p_end = opt->p_end; p_stride = opt->p_stride; }
    if (p_end > np) p_end = np;
    const int compact = opt && opt->compact;
    const size_t npo = compact ? (size_t)((p_end - p_begin + p_stride - 1) / p_stride) : np;
#define PO(p) (compact ? (size_t)(((p) - p_begin) / p_stride) : (size_t)(p))
#ifdef _OPENMP
    int nthr = (opt && opt->num_threads > 0) ? opt->num_threads : omp_get_max_threads();
#endif

    /* ---- materials (:437) ---- */
    mat_pre mats[HRT_NUM_MATERIALS];
    memset(mats, 0, sizeof mats);
    for (uint32_t i = 0; i < sc->num_meshes; ++i) {
        if (sc->mesh_material[i] >= HRT_NUM_MATERIALS) return -1;
        material_eta(sc->mesh_material[i], f_ghz, &mats[sc->mesh_material[i]]);
    }
    if (opt && opt->eta_table) memcpy(opt->eta_table, mats, sizeof mats);

    /* ---- triangles + normals (:438, :208-224) ---- */
    tri_t *tris = (tri_t *)malloc((size_t)(T ? T : 1) * sizeof(tri_t));
    v3 *dirs = (v3 *)malloc(np * sizeof(v3));
    float *st = (float *)malloc(nq * 11 * sizeof(float)); /* o d a[4] tau per ray */
    uint8_t *active = (uint8_t *)malloc(nq / 8 + 1);
    uint32_t *live = (uint32_t *)malloc((np ? np : 1) * sizeof(uint32_t));
    hit_t *hits = (hit_t *)malloc((np ? np : 1) * sizeof(hit_t));
    float *dfs = (float *)malloc((np ? np : 1) * (nrx + 1) * sizeof(float));
    uint8_t *unblocked = (uint8_t *)malloc((np ? np : 1) * nrx);
    uint32_t *live_tx = (uint32_t *)malloc(nq * sizeof(uint32_t));
    size_t *live_cnt = (size_t *)calloc(ntx, sizeof(size_t));
    if (!tris || !dirs || !st || !active || !live || !hits || !dfs || !unblocked || !live_tx ||
        !live_cnt) {
        free(tris); free(dirs); free(st); free(active); free(live); free(hits); free(dfs);
        free(unblocked); free(live_tx); free(live_cnt);
        return -1;
    }
    for (uint32_t j = 0; j < T; ++j) {
        v3 a = ld3(sc->tri_vtx + 9 * j), b = ld3(sc->tri_vtx + 9 * j + 3),
           c = ld3(sc->tri_vtx + 9 * j + 6);
        tris[j].v1 = a;
        tris[j].e1 = sub3(b, a);
        tris[j].e2 = sub3(c, a);
        tris[j].n = unit3(cross3(tris[j].e1, tris[j].e2));
        if (opt && opt->normals) st3(opt->normals + 3 * j, tris[j].n);
    }

    /* ---- launch directions (:443-456): one per path, shared by all tx ---- */
#pragma omp parallel for schedule(static) num_threads(nthr)
    for (size_t p = 0; p < np; ++p) dirs[p] = launch_dir(p, np);
    if (opt && opt->launch_dirs) memcpy(opt->launch_dirs, dirs, np * sizeof(v3));

    /* ---- per-ray state (:460-472) ---- */
    for (size_t tx = 0; tx < ntx; ++tx)
        for (size_t p = 0; p < np; ++p) {
            float *s = st + 11 * (tx * np + p);
            st3(s, ld3(tx_pos + 3 * tx));
            st3(s + 3, dirs[p]);
            s[6] = 1.f; s[7] = 0.f; s[8] = 1.f; s[9] = 0.f; s[10] = 0.f;
        }
    for (size_t i = 0; i < nq / 8 + 1; ++i) active[i] = 0xff;
    if (!compact) for (size_t i = 0; i < nq / 8 + 1; ++i) scat_rays->rays_active[i] = 0xff;

    /* ---- multipliers (:483-488) ---- */
    float f_hz = (float)((double)f_ghz * 1e9);
    float fsl_mult = 4.f * HRT_PI_F * f_hz / HRT_C_F;
    float dop_mult = f_hz / HRT_C_F;

    /* ---- scatter Doppler launch term and its replication (:494-508, quirk Q9) ---- */
    if (compact) {
        for (size_t rx = 0; rx < nrx; ++rx)
            for (size_t tx = 0; tx < ntx; ++tx)
                for (size_t b = 0; b < nb; ++b)
                    for (uint64_t p = p_begin; p < p_end; p += p_stride) {
                        float v = dot3(ld3(tx_vel + 3 * tx), dirs[p]);
                        scat->freq_shift[((rx * ntx + tx) * nb + b) * npo + PO(p)] = v * dop_mult;
                    }
    } else {
        for (size_t tx = 0; tx < ntx; ++tx)
            for (size_t p = 0; p < np; ++p) {
                float v = dot3(ld3(tx_vel + 3 * tx), dirs[p]);
                scat->freq_shift[tx * np * nb + p] = v * dop_mult;
            }
        for (size_t b = 1; b < nb; ++b)
            memcpy(scat->freq_shift + nq * b, scat->freq_shift, nq * sizeof(float));
        for (size_t rx = 1; rx < nrx; ++rx)
            memcpy(scat->freq_shift + nq * nb * rx, scat->freq_shift, nq * nb * sizeof(float));
    }

    /* algorithmic count: the LoS pass is priced at nrx*ntx*T */
    uint64_t n_tests = (uint64_t)nrx * ntx * T;

    /* ---- LoS (:515-577) ---- */
    for (size_t off = 0; off < nrx * ntx; ++off) los->a_te_im[off] = los->a_tm_im[off] = 0.f;
    for (size_t rx = 0, off = 0; rx < nrx; ++rx)
        for (size_t tx = 0; tx < ntx; ++tx, ++off) {
            v3 o = ld3(tx_pos + 3 * tx);
            v3 d = sub3(ld3(rx_pos + 3 * rx), o);
            st3(los_rays->rays + 6 * off, o);
            st3(los_rays->rays + 6 * off + 3, d);
            uint8_t bit = (uint8_t)(1u << (off % 8));
            if (dot3(d, d) < HRT_EPS) {
                v3 ex = {1.f, 0.f, 0.f}, mex = {-1.f, 0.f, 0.f};
                st3(los->directions_rx + 3 * off, ex);
                st3(los->directions_tx + 3 * off, mex);
                los->a_te_re[off] = los->a_tm_re[off] = 1.f;
                los->tau[off] = 0.f;
                los->freq_shift[off] = 0.f;
                los_rays->rays_active[off / 8] |= bit;
                continue;
            }
            hit_t h = closest_hit(tris, T, o, d);
            if (h.tri != HRT_NO_HIT && h.t <= 1.f) {
                los->a_te_re[off] = los->a_tm_re[off] = los->tau[off] = 0.f;
                los_rays->rays_active[off / 8] &= (uint8_t)~bit;
                continue;
            }
            float dist = sqrtf(dot3(d, d));
            v3 u = {d.x / dist, d.y / dist, d.z / dist};
            v3 mu = {-u.x, -u.y, -u.z};
            st3(los->directions_tx + 3 * off, u);
            st3(los->directions_rx + 3 * off, mu);
            float fsl = fsl_mult * dist;
            los->a_te_re[off] = los->a_tm_re[off] = (fsl > 1.f) ? 1.f / fsl : 1.f;   /* Q4 */
            los->tau[off] = dist / HRT_C_F;
            /* Q5: always tx_vel[0] / rx_vel[0] */
            float fs = dot3(ld3(tx_vel), u) - dot3(ld3(rx_vel), u);
            los->freq_shift[off] = fs * (f_hz / HRT_C_F);
            los_rays->rays_active[off / 8] |= bit;
        }

    /* ---- launch snapshot (:589) ---- */
    if (!compact)
        for (size_t q = 0; q < nq; ++q) memcpy(scat_rays->rays + 6 * q, st + 11 * q, 6 * sizeof(float));

    /* ---- live lists per tx, in path order ---- */
    for (size_t tx = 0; tx < ntx; ++tx) {
        size_t n = 0;
        for (uint64_t p = p_begin; p < p_end; p += p_stride) live_tx[tx * np + n++] = (uint32_t)p;
        live_cnt[tx] = n;
    }
    if (opt && opt->live) memset(opt->live, 0, (nb + 1) * sizeof(uint64_t));
    if (opt && opt->hit_tri) memset(opt->hit_tri, 0xff, nb * (compact ? ntx * npo : nq) * sizeof(uint32_t));

    /* rays outside the processed subset are "never live": clear their bits so that the
     * active masks equal a run in which they all missed at bounce 0.  (Full runs: no-op.) */
    if (!(p_begin == 0 && p_end == np && p_stride == 1)) {
        for (size_t q = 0; q < nq; ++q) active[q / 8] &= (uint8_t)~(1u << (q % 8));
        for (size_t tx = 0; tx < ntx; ++tx)
            for (size_t i = 0; i < live_cnt[tx]; ++i) {
                size_t q = tx * np + live_tx[tx * np + i];
                active[q / 8] |= (uint8_t)(1u << (q % 8));
            }
    }

    /* ---- bounces (:591-745) ---- */
    for (size_t b = 0; b < nb; ++b) {
        for (size_t tx = 0; tx < ntx; ++tx) {
            const size_t n_live = live_cnt[tx];
            const uint32_t *lv = live_tx + tx * np;
            if (opt && opt->live) opt->live[b] += n_live;
            uint64_t n_hit = 0;

            /* data-parallel part: every live ray of this tx */
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : n_hit) num_threads(nthr)
            for (size_t i = 0; i < n_live; ++i) {
                const size_t p = lv[i], q = tx * np + p;
                float *s = st + 11 * q;
                v3 o = ld3(s), d = ld3(s + 3);
                hit_t h = closest_hit(tris, T, o, d);                         /* :615 */
                hits[i] = h;
                if (h.tri == HRT_NO_HIT) continue;                            /* :616-620 */
                ++n_hit;
                const tri_t *tr = &tris[h.tri];
                const uint32_t mesh = sc->tri_mesh[h.tri];
                const uint32_t mat = sc->mesh_material[mesh];
                float R[4];
                fresnel(&mats[mat], h.theta, R);                              /* :622-625 */
                float fsl = fsl_mult * h.t;                                   /* :627-634 */
                fsl *= fsl;
                if (fsl > 1.f) { R[0] /= fsl; R[1] /= fsl; R[2] /= fsl; R[3] /= fsl; }
                float a0 = s[6] * R[0] - s[7] * R[1];                         /* :636-643 */
                float a1 = s[6] * R[1] + s[7] * R[0];
                float a2 = s[8] * R[2] - s[9] * R[3];
                float a3 = s[8] * R[3] + s[9] * R[2];
                s[6] = a0; s[7] = a1; s[8] = a2; s[9] = a3;
                s[10] += h.t / HRT_C_F;                                       /* :645 */
                o = add3(mul3(d, h.t), o);                                    /* :650-651 */
                float dn = dot3(d, tr->n);                                    /* :654 */
                d = sub3(d, mul3(tr->n, 2.f * dn));                           /* :655-656 */
                o = add3(o, mul3(d, 1e-4f));                                  /* :658-659 */
                st3(s, o);
                st3(s + 3, d);
                const v3 mvel = ld3(sc->mesh_velocity + 3 * mesh);
                /* :663-664 (Q10): r aliases rays[off_tx_path] -> d - d */
                v3 zz = sub3(d, d);
                dfs[i * (nrx + 1) + nrx] = dot3(zz, mvel) * dop_mult;

                /* scatter to every rx IN ORDER, carrying theta (Q7)           :671-723 */
                float theta = h.theta;
                for (size_t rx = 0; rx < nrx; ++rx) {
                    const size_t off = ((rx * ntx + tx) * nb + b) * npo + PO(p);   /* :674 */
                    v3 w = sub3(ld3(rx_pos + 3 * rx), o);
                    float d2rx = sqrtf(dot3(w, w));
                    w = unit3(w);
                    hit_t sh = closest_hit(tris, T, o, w);
                    if (sh.tri != HRT_NO_HIT) theta = sh.theta;               /* Q7 */
                    if (sh.tri != HRT_NO_HIT && sh.t <= 1.f) {                /* Q6 */
                        scat->a_te_re[off] = scat->a_te_im[off] = scat->a_tm_re[off] =
                            scat->a_tm_im[off] = scat->tau[off] = 0.f;
                        unblocked[i * nrx + rx] = 0;
                        continue;
                    }
                    float th_s = acosf(dot3(w, tr->n));                       /* :694 */
                    float S[4];
                    scatter_pattern(th_s, theta, mat, S);
                    float o0 = a0 * S[0] - a1 * S[1];                         /* :698-705, Q8 */
                    float o1 = a0 * S[1] + a1 * S[0];
                    float o2 = a2 * S[2] - a3 * S[3];
                    float o3 = a2 * S[3] + a3 * S[2];
                    v3 mw = {-w.x, -w.y, -w.z};
                    st3(scat->directions_rx + 3 * off, mw);                   /* :707 */
                    scat->tau[off] = s[10] + d2rx / HRT_C_F;                  /* :709 */
                    float f2 = fsl_mult * d2rx;                               /* :711-718 */
                    f2 *= f2;
                    if (f2 > 1.f) { o0 /= f2; o1 /= f2; o2 /= f2; o3 /= f2; }
                    scat->a_te_re[off] = o0; scat->a_te_im[off] = o1;
                    scat->a_tm_re[off] = o2; scat->a_tm_im[off] = o3;
                    dfs[i * (nrx + 1) + rx] = dot3(sub3(w, d), mvel) * dop_mult; /* :720-721 */
                    unblocked[i * nrx + rx] = 1;
                }
            }

            /* order-dependent side effects, replayed in the reference's (path) order */
            size_t n_next = 0;
            uint32_t *nx = live_tx + tx * np; /* in-place compaction keeps path order */
            for (size_t i = 0; i < n_live; ++i) {
                const size_t p = lv[i], q = tx * np + p;
                if (hits[i].tri == HRT_NO_HIT) {
                    active[q / 8] &= (uint8_t)~(1u << (q % 8));               /* :617 */
                    continue;
                }
                const size_t qo = compact ? (b * ntx + tx) * npo + PO(p) : b * nq + q;
                if (opt && opt->hit_tri) opt->hit_tri[qo] = hits[i].tri;
                if (opt && opt->hit_theta) opt->hit_theta[qo] = hits[i].theta;
                if (!compact) scat->freq_shift[q] += dfs[i * (nrx + 1) + nrx];   /* :664, Q10 */
                for (size_t rx = 0; rx < nrx; ++rx)
                    if (unblocked[i * nrx + rx]) {
                        const size_t off = ((rx * ntx + tx) * nb + b) * npo + PO(p);
                        scat->freq_shift[off] -= dfs[i * (nrx + 1) + rx];     /* :722 */
                    }
                nx[n_next++] = (uint32_t)p;
            }
            live_cnt[tx] = n_next;
            n_tests += (uint64_t)T * (n_live + (uint64_t)nrx * n_hit);
            if (opt && opt->live && b + 1 == nb) opt->live[nb] += n_hit;

            /* snapshots (:732-743, quirks Q11/Q12/Q14: stride nb, mask always from byte 0) */
            if (!compact) {
                size_t off_rays = (tx * nb + (b + 1)) * np;
                size_t off_act = (tx * nb + (b + 1)) * (np / 8 + 1);
                for (size_t p = 0; p < np; ++p)
                    memcpy(scat_rays->rays + 6 * (off_rays + p), st + 11 * (tx * np + p),
                           6 * sizeof(float));
                memcpy(scat_rays->rays_active + off_act, active, np / 8 + 1);
            }
        }
    }
    if (opt && opt->tests) *opt->tests = n_tests;
#undef PO

    free(tris); free(dirs); free(st); free(active); free(live); free(hits); free(dfs);
    free(unblocked); free(live_tx); free(live_cnt);
    return 0;
}

/* Host libm over an array, for tests/test_gpu_libm.py: fn 0 sinf, 1 cosf, 2 expf, 3 acosf,
 * 4 the incidence angle (src/compute_paths.c:281-283) for dot = in[i]. */
void hrt_oracle_libm(int fn, const float *in, float *out, size_t n)
{
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        float x = in[i], y;
        switch (fn) {
        case 0: y = sinf(x); break;
        case 1: y = cosf(x); break;
        case 2: y = expf(x); break;
        case 3: y = acosf(x); break;
        default:
            y = (float)acos((double)x);
            if ((double)y > (double)HRT_PI_F / 2.) y = HRT_PI_F - y;
        }
        out[i] = y;
    }
}
