/* problem.c -- host orchestration of the device-resident path (include/hrt_device.h).
 *
 * Host-side, once-per-call work of the reference that stays on the host:
 *   precompute_materials  src/compute_paths.c:171-206   (17 x 12 floats, powf/sqrtf)
 *   precompute_normals    src/compute_paths.c:208-224   (one cross/normalise per triangle)
 *   Fibonacci launch set  src/compute_paths.c:443-451   (double libm; threads)
 * plus what the reference has no need for: flattening the Scene into one 64-byte-per-
 * triangle table in (mesh, face) order, the workspace layout, shard arithmetic, and the
 * launch sequence of the HIP kernels (through the shim in ../hrt_kernels.hip).
 */
#include <math.h>
#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "hrt_internal.h"

/* ------------------------------------------------------------------ errors */

static __thread char g_err[512];

const char *hrt_last_error(void) { return g_err; }
const char *hrt_version(void) { return "hermespy-rt_amd 0.1 (gfx950)"; }

int hrt_fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

int hrt_fail_hip(int hip_err, const char *what)
{
    snprintf(g_err, sizeof g_err, "%s: %s", what, hrt_hip_error_string(hip_err));
    return HRT_E_HIP;
}

double hrt_now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ------------------------------------------------------------------ problem */

static uint64_t round_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

static inline Vec3 v_sub(Vec3 a, Vec3 b) { Vec3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }

/* unit normal of (e1, e2): cross then divide by sqrtf of the squared length, in the operand
 * order of inc/vec3.h:20-43 of the reference */
static Vec3 unit_normal(Vec3 e1, Vec3 e2)
{
    Vec3 c = {e1.y * e2.z - e1.z * e2.y, e1.z * e2.x - e1.x * e2.z, e1.x * e2.y - e1.y * e2.x};
    float len = sqrtf(c.x * c.x + c.y * c.y + c.z * c.z);
    Vec3 n = {c.x / len, c.y / len, c.z / len};
    return n;
}

void hrt_problem_destroy(hrt_problem *p)
{
    if (!p) return;
    if (p->d_blob) {
        hrt_hip_set_device(p->device);
        hrt_hip_free(p->d_blob);
        if (p->d_rxt) hrt_hip_free(p->d_rxt);
        if (p->d_patch) hrt_hip_free(p->d_patch);
        if (p->h_fuse_flag) hrt_hip_host_free(p->h_fuse_flag);
        if (p->aux_stream) hrt_hip_stream_destroy(p->aux_stream);
        for (int k = 0; k < 2; ++k) if (p->aux_ev[k]) hrt_hip_event_destroy(p->aux_ev[k]);
    }
    free(p->h_tri); free(p->h_mesh); free(p->h_mat); free(p->h_tri_mesh); free(p->h_tri_face);
    hrt_accel_free(&p->accel);
    free(p);
}

/* ---- per-RX direction tables for the shadow rays (hrt_kparams.h, hrt_krxt) ----
 * cube-map cell -> direction: the inverse of rxt_cell() in hrt_kernels.hip */
static void rxt_dir(uint32_t f, double u, double v, double out[3])
{
    const uint32_t m = f % 3u;
    const double sgn = f >= 3u ? -1.0 : 1.0;
    double c[3];
    c[m] = sgn;
    c[(m + 1u) % 3u] = u * sgn;
    c[(m + 2u) % 3u] = v * sgn;
    const double l = sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    out[0] = c[0] / l; out[1] = c[1] / l; out[2] = c[2] / l;
}

/* cones of the 6 x 48 x 48 cube-map cells (axis as the device gets it, cos / sin of the half-angle),
 * widened by `aq` (the widest packet served; 0 for per-ray lookups) and by the rounding of the
 * device's cell lookup.  A pure function of aq: computed once per process for each of the two
 * values in use. */
static pthread_mutex_t g_bins_lock = PTHREAD_MUTEX_INITIALIZER;
static float *g_bins[2];   /* [0] aq = asin(HRT_RXT_SIN_AQ), [1] aq = 0: NB * 6 floats (dir4, cs2) */
static const float *rxt_bins(int per_ray)
{
    pthread_mutex_lock(&g_bins_lock);
    float *b = g_bins[per_ray];
    if (!b) {
        const uint32_t NB = HRT_RXT_BINS;
        b = (float *)malloc((size_t)NB * 6 * sizeof(float));
        if (b) {
            float *bin_dir = b, *bin_cs = b + (size_t)NB * 4;
            const double aq = per_ray ? 0.0 : asin((double)HRT_RXT_SIN_AQ);
            for (uint32_t f = 0; f < 6; ++f)
                for (uint32_t iv = 0; iv < HRT_RXT_N; ++iv)
                    for (uint32_t iu = 0; iu < HRT_RXT_N; ++iu) {
                        const uint32_t cell = (f * HRT_RXT_N + iv) * HRT_RXT_N + iu;
                        const double u0 = 2.0 * iu / HRT_RXT_N - 1.0, u1 = 2.0 * (iu + 1) / HRT_RXT_N - 1.0;
                        const double v0 = 2.0 * iv / HRT_RXT_N - 1.0, v1 = 2.0 * (iv + 1) / HRT_RXT_N - 1.0;
                        double ctr[3], q[3], cmin = 1.0;
                        rxt_dir(f, 0.5 * (u0 + u1), 0.5 * (v0 + v1), ctr);
                        const float cfl[3] = {(float)ctr[0], (float)ctr[1], (float)ctr[2]};   /* what the device gets */
                        const double fl = sqrt((double)cfl[0] * cfl[0] + (double)cfl[1] * cfl[1] + (double)cfl[2] * cfl[2]);
                        const double us[2] = {u0, u1}, vs[2] = {v0, v1};
                        for (int a = 0; a < 2; ++a)
                            for (int c2 = 0; c2 < 2; ++c2) {
                                rxt_dir(f, us[a], vs[c2], q);
                                const double cc = (q[0] * cfl[0] + q[1] * cfl[1] + q[2] * cfl[2]) / fl;
                                if (cc < cmin) cmin = cc;
                            }
                        /* cone of the cell + the widest packet served + rounding of the cell lookup */
                        const double ang = acos(cmin > 1 ? 1 : cmin) + aq + 3e-4;
                        bin_dir[4 * cell] = cfl[0]; bin_dir[4 * cell + 1] = cfl[1]; bin_dir[4 * cell + 2] = cfl[2]; bin_dir[4 * cell + 3] = 0.f;
                        float cs = (float)cos(ang), sn = (float)sin(ang);
                        if ((double)cs > cos(ang)) cs = nextafterf(cs, -1.f);      /* wider, never narrower */
                        if ((double)sn < sin(ang)) sn = nextafterf(sn, 2.f);
                        bin_cs[2 * cell] = cs; bin_cs[2 * cell + 1] = sn;
                    }
            g_bins[per_ray] = b;
        }
    }
    pthread_mutex_unlock(&g_bins_lock);
    return b;
}

static int rxt_build(hrt_problem *p, const Vec3 *rx_pos, const Vec3 *tx_pos)
{
    /* apexes: the RXs (shadow rays converge on them), then the TXs (the launch set leaves them) */
    const uint32_t T = p->num_tri, n_rx = p->num_rx;
    uint32_t max_tri = p->tune.rxt_max_tri;
    if (max_tri > 65535u) max_tri = 65535u;
    /* Up to 64 triangles the whole table is ONE culling round, a list cannot be cheaper: there the
     * tables are one 64-bit candidate MASK per (apex, cell), built for the cell alone, looked up per
     * ray (closest_hit_masked) -- RXs and TXs alike, up to 256 apexes (110 KB each). */
    const int per_ray = T <= 64;
    if (T > max_tri || p->tune.no_rxt) return HRT_OK;
    if (per_ray ? (n_rx + p->num_tx > 256) : (n_rx > 64)) return HRT_OK;
    const uint32_t n_txt = per_ray ? p->num_tx
                                   : ((n_rx + p->num_tx <= 64 && !p->tune.no_txt) ? p->num_tx : 0u);
    const uint32_t nrx = n_rx + n_txt;
    /* the ball every ray origin lies in: hit points are on triangles (+ 1e-4 along the new direction) */
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t j = 0; j < T; ++j) {
        const float *r = p->h_tri + (size_t)j * HRT_TRI_FLOATS;
        for (int v = 0; v < 3; ++v)
            for (int k = 0; k < 3; ++k) {
                const double x = (double)r[k] + (v == 1 ? (double)r[3 + k] : (v == 2 ? (double)r[6 + k] : 0.0));
                if (!isfinite(x)) return HRT_OK;   /* non-finite geometry: no tables, the whole table is walked */
                if (x < lo[k]) lo[k] = x;
                if (x > hi[k]) hi[k] = x;
            }
    }
    for (uint32_t t = 0; t < n_txt; ++t) {   /* ... and the launch set starts at the TXs */
        const double x[3] = {tx_pos[t].x, tx_pos[t].y, tx_pos[t].z};
        for (int k = 0; k < 3; ++k) {
            if (!isfinite(x[k])) return HRT_OK;
            if (x[k] < lo[k]) lo[k] = x[k];
            if (x[k] > hi[k]) hi[k] = x[k];
        }
    }
    const double c[3] = {0.5 * (lo[0] + hi[0]), 0.5 * (lo[1] + hi[1]), 0.5 * (lo[2] + hi[2])};
    const double half = 0.5 * sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2]));
    const float cf[3] = {(float)c[0], (float)c[1], (float)c[2]};
    const float region_r = (float)(half * 1.01 + 0.01 + 1e-5 * (fabs(c[0]) + fabs(c[1]) + fabs(c[2])));
    if (!isfinite(region_r)) return HRT_OK;

    const uint32_t NB = HRT_RXT_BINS, W = (T + 63u) / 64u;
    const float *bins = rxt_bins(per_ray);
    if (!bins) return hrt_fail(HRT_E_NOMEM, "out of host memory");
    const float *bin_dir = bins, *bin_cs = bins + (size_t)NB * 4;
    float *ro_bin = (float *)malloc((size_t)nrx * sizeof(float));
    float *apex = (float *)malloc((size_t)nrx * 3 * sizeof(float));
    uint64_t *masks = per_ray ? NULL : (uint64_t *)malloc((size_t)nrx * NB * W * 8);
    uint32_t *off = per_ray ? NULL : (uint32_t *)malloc(((size_t)nrx * NB + 1) * 4);
    int rc = HRT_OK, e;
    void *d_tmp = NULL;
    uint16_t *idx = NULL;
    if (!ro_bin || !apex || (!per_ray && (!masks || !off))) { rc = hrt_fail(HRT_E_NOMEM, "out of host memory"); goto out; }
    for (uint32_t r = 0; r < nrx; ++r) {
        const Vec3 a = r < n_rx ? rx_pos[r] : tx_pos[r - n_rx];
        apex[3 * r] = a.x; apex[3 * r + 1] = a.y; apex[3 * r + 2] = a.z;
        const double dx = (double)a.x - c[0], dy = (double)a.y - c[1], dz = (double)a.z - c[2];
        const double lmax = (sqrt(dx * dx + dy * dy + dz * dz) + 2.0 * (double)region_r) * 1.001;
        ro_bin[r] = (float)(8.0 * 0.5 * 1.1920928955078125e-07 * lmax * 1.001 + 2e-7);
        /* a launch packet: every origin IS the TX; the kernel takes the radius of its origin ball
         * (1e-6 of the 1-norm of its centre, origin_ball()) plus the centre's distance from the TX */
        if (r >= n_rx) ro_bin[r] = (float)(4e-6 * (fabs((double)a.x) + fabs((double)a.y) + fabs((double)a.z)) + 4e-6);
        if (!isfinite(ro_bin[r])) goto out;   /* an RX at infinity: no tables */
    }
    {
        const uint64_t b_dir = (uint64_t)NB * 16, b_cs = (uint64_t)NB * 8, b_ro = (uint64_t)nrx * 4 + 252;
        const uint64_t b_mask = (uint64_t)nrx * NB * W * 8, b_apex = round_up((uint64_t)nrx * 12, 256);
        if ((e = hrt_hip_malloc(&d_tmp, b_dir + b_cs + (b_ro & ~255ull) + 256 + b_apex + (per_ray ? 0 : b_mask)))) { rc = hrt_fail_hip(e, "hipMalloc(rxt build)"); goto out; }
        uint8_t *q = (uint8_t *)d_tmp;
        float *d_dir = (float *)q; q += b_dir;
        float *d_cs = (float *)q; q += b_cs;
        float *d_ro = (float *)q; q += (b_ro & ~255ull) + 256;
        float *d_apex = (float *)q; q += b_apex;
        unsigned long long *d_masks = (unsigned long long *)q;
        if ((e = hrt_hip_h2d(d_dir, bin_dir, b_dir)) || (e = hrt_hip_h2d(d_cs, bin_cs, b_cs)) ||
            (e = hrt_hip_h2d(d_ro, ro_bin, (uint64_t)nrx * 4)) ||
            (e = hrt_hip_h2d(d_apex, apex, (uint64_t)nrx * 12))) { rc = hrt_fail_hip(e, "hipMemcpy(rxt build)"); goto out; }
        if (per_ray) {
            /* the masks ARE the table: built straight into the problem's own allocation */
            if ((e = hrt_hip_malloc(&p->d_rxt, b_mask))) { p->d_rxt = NULL; rc = hrt_fail_hip(e, "hipMalloc(rxt)"); goto out; }
            d_masks = (unsigned long long *)p->d_rxt;
        }
        if ((e = hrt_hip_rxt_build(p->d_tri, T, d_apex, nrx, d_dir, d_cs, d_ro, cf[0], cf[1], cf[2], region_r,
                                   d_masks, NULL))) { rc = hrt_fail_hip(e, "hrt_rxt_build_kernel"); goto out; }
        if ((e = hrt_hip_stream_sync(NULL))) { rc = hrt_fail_hip(e, "hipStreamSynchronize"); goto out; }
        if (per_ray) {
            p->krxt.cell_mask = d_masks;
            p->krxt.num_txt = n_txt;
            p->krxt.cx = cf[0]; p->krxt.cy = cf[1]; p->krxt.cz = cf[2]; p->krxt.region_r = region_r;
            goto out;
        }
        if ((e = hrt_hip_d2h(masks, d_masks, b_mask))) { rc = hrt_fail_hip(e, "hipMemcpy D2H"); goto out; }
    }
    {
        uint64_t total = 0;
        for (uint64_t k = 0; k < (uint64_t)nrx * NB; ++k) {
            off[k] = (uint32_t)total;
            for (uint32_t w = 0; w < W; ++w) total += (uint64_t)__builtin_popcountll(masks[k * W + w]);
        }
        off[(uint64_t)nrx * NB] = (uint32_t)total;
        if (total > 0xfffffff0ull) goto out;
        idx = (uint16_t *)malloc((size_t)(total ? total : 1) * 2);
        if (!idx) { rc = hrt_fail(HRT_E_NOMEM, "out of host memory"); goto out; }
        uint64_t n = 0;
        for (uint64_t k = 0; k < (uint64_t)nrx * NB; ++k)
            for (uint32_t w = 0; w < W; ++w) {
                uint64_t m = masks[k * W + w];
                while (m) {
                    idx[n++] = (uint16_t)(w * 64u + (uint32_t)__builtin_ctzll(m));
                    m &= m - 1;
                }
            }
        const uint64_t b_ro = round_up((uint64_t)nrx * 4, 256), b_off = round_up(((uint64_t)nrx * NB + 1) * 4, 256);
        const uint64_t b_idx = round_up((total ? total : 1) * 2, 256);
        if ((e = hrt_hip_malloc(&p->d_rxt, b_ro + b_off + b_idx))) { p->d_rxt = NULL; rc = hrt_fail_hip(e, "hipMalloc(rxt)"); goto out; }
        uint8_t *q = (uint8_t *)p->d_rxt;
        if ((e = hrt_hip_h2d(q, ro_bin, (uint64_t)nrx * 4)) || (e = hrt_hip_h2d(q + b_ro, off, ((uint64_t)nrx * NB + 1) * 4)) ||
            (total && (e = hrt_hip_h2d(q + b_ro + b_off, idx, total * 2)))) { rc = hrt_fail_hip(e, "hipMemcpy(rxt)"); goto out; }
        p->krxt.enabled = 1u;
        p->krxt.num_txt = n_txt;
        p->krxt.cx = cf[0]; p->krxt.cy = cf[1]; p->krxt.cz = cf[2]; p->krxt.region_r = region_r;
        p->krxt.ro_bin = (const float *)q;
        p->krxt.off = (const uint32_t *)(q + b_ro);
        p->krxt.idx = (const uint16_t *)(q + b_ro + b_off);
        p->rxt_entries = total;
    }
out:
    if (d_tmp) hrt_hip_free(d_tmp);
    free(ro_bin); free(apex); free(masks); free(off); free(idx);
    if (rc && p->d_rxt) { hrt_hip_free(p->d_rxt); p->d_rxt = NULL; p->krxt.cell_mask = NULL; }
    return rc;
}

/* ---- patch tables (hrt_kpatch, hrt_kparams.h): the grid of every triangle, the list patch -> triangle,
 * and the device pass that fills the candidate masks.  Tables of 65 .. HRT_PATCH_MAX_TRI triangles. ----
 * Patch edge: 0.5 m (HRT_TUNE patch_size: C3 4.5 candidates per shadow ray against 7.4 at 1 m and
 * 17.7 at 2 m, profiles/study/), enlarged until all tables fit HRT_PATCH_MAX_BYTES (default 1 GiB). */
static int patch_build(hrt_problem *p, const Vec3 *rx_pos, const Vec3 *tx_pos)
{
    const uint32_t T = p->num_tri, n_rx = p->num_rx, n_tx = p->num_tx;
    if (p->tune.no_patch) return HRT_OK;
    /* (up to 64 triangles the fused kernels with their per-cell masks are faster: patch tables + a records
     * kernel measured 0.86 against 0.66 ms on C4, 0.080 / 0.050 on C2, 0.047 / 0.022 on C1) */
    if (T <= 64u || T > HRT_PATCH_MAX_TRI || n_rx + n_tx > 4096u) return HRT_OK;
    double size = p->tune.patch_size > 0.0 ? p->tune.patch_size : 0.5, max_bytes = 1024.0 * 1024.0 * 1024.0;
    { const char *v = getenv("HRT_PATCH_MAX_BYTES"); if (v && *v && atof(v) > 0.0) max_bytes = atof(v); }
    if (max_bytes > 3.5e9) max_bytes = 3.5e9;   /* the kernels address the masks with 32-bit byte offsets */
    /* extent of everything a ray origin or an apex can be */
    double ext1 = 0.0, cmax = 0.0;
    for (int k = 0; k < 3; ++k) {
        ext1 += (double)p->scene_hi[k] - (double)p->scene_lo[k];
        cmax += fmax(fabs((double)p->scene_hi[k]), fabs((double)p->scene_lo[k]));
    }
    if (!isfinite(ext1) || !isfinite(cmax)) return HRT_OK;
    const double u = 5.9604644775390625e-08;   /* 2^-24 */
    const double Smax = ext1 + 1.0;            /* >= |o - v1| for any origin within hmax of the scene's box */
    const float hmax = (float)(4e-4 + 4e-6 * cmax);
    const uint32_t n_apex = n_rx + n_tx;
    float *pdef = (float *)calloc((size_t)T * 8, sizeof(float));
    uint32_t *nuv = (uint32_t *)calloc((size_t)T * 2, sizeof(uint32_t));
    uint32_t *ptri = NULL;
    float *apex = (float *)malloc((size_t)n_apex * 3 * sizeof(float));
    void *d_tmp = NULL;
    int rc = HRT_OK, e;
    uint64_t npatch = 0;
    if (!pdef || !nuv || !apex) { rc = hrt_fail(HRT_E_NOMEM, "out of host memory"); goto out; }
    for (int pass = 0; pass < 32; ++pass) {
        npatch = 0;
        for (uint32_t j = 0; j < T; ++j) {
            const float *r = p->h_tri + (size_t)j * HRT_TRI_FLOATS;
            const double e1[3] = {r[3], r[4], r[5]}, e2[3] = {r[6], r[7], r[8]};
            const double a11 = e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2];
            const double a22 = e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2];
            const double a12 = e1[0] * e2[0] + e1[1] * e2[1] + e1[2] * e2[2];
            const double det = a11 * a22 - a12 * a12;
            nuv[2 * j] = nuv[2 * j + 1] = 0u;
            if (!(det > 1e-30) || !isfinite(det) || !isfinite(r[9] + r[10] + r[11])) continue;   /* degenerate: not served */
            double nu = ceil(sqrt(a11) / size), nv = ceil(sqrt(a22) / size);
            if (nu < 1) nu = 1;
            if (nv < 1) nv = 1;
            if (nu > 2048) nu = 2048;
            if (nv > 2048) nv = 2048;
            double g1[3], g2[3], l1 = 0, l2 = 0;
            for (int k = 0; k < 3; ++k) {
                g1[k] = (a22 * e1[k] - a12 * e2[k]) / det * nu;
                g2[k] = (a11 * e2[k] - a12 * e1[k]) / det * nv;
                l1 += g1[k] * g1[k];
                l2 += g2[k] * g2[k];
            }
            /* rounding of the kernel's cell coordinates (s = o - v1, three fused terms): must stay below
             * 1/64 of a cell -- with the clamp of HRT_PATCH_ACCEPT an origin is then never further than
             * HRT_PATCH_MARGIN outside its cell (needle triangles fail this and are not served) */
            const double v1n = fabs((double)r[0]) + fabs((double)r[1]) + fabs((double)r[2]);
            if (!(16.0 * u * (Smax + v1n) * sqrt(l1 > l2 ? l1 : l2) <= 1.0 / 64.0)) continue;
            nuv[2 * j] = (uint32_t)nu;
            nuv[2 * j + 1] = (uint32_t)nv;
            float *q = pdef + (size_t)j * 8;
            const uint32_t base = (uint32_t)npatch, bits = (uint32_t)nu | ((uint32_t)nv << 16);
            q[0] = (float)g1[0]; q[1] = (float)g1[1]; q[2] = (float)g1[2]; memcpy(&q[3], &base, 4);
            q[4] = (float)g2[0]; q[5] = (float)g2[1]; q[6] = (float)g2[2]; memcpy(&q[7], &bits, 4);
            npatch += (uint64_t)(nu * nv);
        }
        if (npatch == 0) goto out;
        const double bytes = (double)npatch * n_apex * HRT_PATCH_WORDS * 8.0;
        if (bytes <= max_bytes && npatch < 0x7fffffffull) break;
        size *= fmax(1.05, sqrt(bytes / max_bytes) * 1.02);
        if (pass == 31) goto out;   /* (cannot happen: the size grows geometrically) */
    }
    for (uint32_t j = 0; j < T; ++j)   /* unserved triangles: nu = nv = 0 in the table */
        if (nuv[2 * j] == 0u) { uint32_t z = 0; float *q = pdef + (size_t)j * 8; memcpy(&q[3], &z, 4); memcpy(&q[7], &z, 4); }
    ptri = (uint32_t *)malloc((size_t)npatch * 4);
    if (!ptri) { rc = hrt_fail(HRT_E_NOMEM, "out of host memory"); goto out; }
    {
        uint64_t k = 0;
        for (uint32_t j = 0; j < T; ++j)
            for (uint64_t c = 0; c < (uint64_t)nuv[2 * j] * nuv[2 * j + 1]; ++c) ptri[k++] = j;
    }
    for (uint32_t a = 0; a < n_apex; ++a) {
        const Vec3 v = a < n_rx ? rx_pos[a] : tx_pos[a - n_rx];
        if (!isfinite(v.x) || !isfinite(v.y) || !isfinite(v.z)) goto out;   /* an apex at infinity: no tables */
        apex[3 * a] = v.x; apex[3 * a + 1] = v.y; apex[3 * a + 2] = v.z;
        cmax = fmax(cmax, fabs((double)v.x) + fabs((double)v.y) + fabs((double)v.z));
    }
    {
        /* a shadow ray built as normalise(rx - o) passes within 4 u L of the RX (packet_bounds): L <= apex
         * distance from anywhere in the box; an image-apex lane is CHECKED against ro_img by the kernel */
        const float ro_rx = (float)(8.0 * u * (2.0 * cmax + ext1) * 1.001 + 2e-7);
        const float ro_img = (float)(1e-3 + 1e-5 * (cmax + ext1));
        /* a served lane's COMPUTED plane distance is <= hmax; the true one then <= hball */
        const float hball = (float)((double)hmax * 1.01 + 16.0 * u * (Smax + cmax));
        const uint64_t b_pdef = round_up((uint64_t)T * 32, 256), b_mask = round_up(npatch * n_apex * HRT_PATCH_WORDS * 8, 256);
        const uint64_t b_txc = (uint64_t)n_tx * HRT_RXT_BINS * HRT_PATCH_WORDS * 8;   /* the TX cell masks (launch 0) */
        const float *bins = rxt_bins(1);   /* the cells' cones alone (+ the rounding of the lookup) */
        if (!bins) { rc = hrt_fail(HRT_E_NOMEM, "out of host memory"); goto out; }
        const uint64_t b_ptri = round_up(npatch * 4, 256), b_apex = round_up((uint64_t)n_apex * 12, 256);
        const uint64_t b_bins = (uint64_t)HRT_RXT_BINS * 24;
        if ((e = hrt_hip_malloc(&p->d_patch, b_pdef + b_mask + b_txc))) { p->d_patch = NULL; rc = hrt_fail_hip(e, "hipMalloc(patch tables)"); goto out; }
        if ((e = hrt_hip_malloc(&d_tmp, b_ptri + b_apex + b_bins))) { rc = hrt_fail_hip(e, "hipMalloc(patch build)"); goto out; }
        uint8_t *q = (uint8_t *)p->d_patch;
        if ((e = hrt_hip_h2d(q, pdef, (uint64_t)T * 32)) || (e = hrt_hip_h2d(d_tmp, ptri, npatch * 4)) ||
            (e = hrt_hip_h2d((uint8_t *)d_tmp + b_ptri, apex, (uint64_t)n_apex * 12))) { rc = hrt_fail_hip(e, "hipMemcpy(patch build)"); goto out; }
        if ((e = hrt_hip_patch_build(p->d_tri, T, (const float *)q, (const uint32_t *)d_tmp, (uint32_t)npatch,
                                     (const float *)((uint8_t *)d_tmp + b_ptri), n_rx, n_tx, hball, ro_rx, ro_img,
                                     (unsigned long long *)(q + b_pdef), NULL))) { rc = hrt_fail_hip(e, "hrt_patch_build_kernel"); goto out; }
        {   /* TX cell masks: bins = [NB][4] axes then [NB][2] (cos, sin) */
            uint8_t *db = (uint8_t *)d_tmp + b_ptri + b_apex;
            if ((e = hrt_hip_h2d(db, bins, b_bins))) { rc = hrt_fail_hip(e, "hipMemcpy(patch build)"); goto out; }
            if ((e = hrt_hip_txcell_build(p->d_tri, T, (const float *)((uint8_t *)d_tmp + b_ptri) + 3 * (size_t)n_rx, n_tx, (const float *)db,
                                          (const float *)(db + (uint64_t)HRT_RXT_BINS * 16), (unsigned long long *)(q + b_pdef + b_mask), NULL))) {
                rc = hrt_fail_hip(e, "hrt_txcell_build_kernel"); goto out;
            }
        }
        if ((e = hrt_hip_stream_sync(NULL))) { rc = hrt_fail_hip(e, "hipStreamSynchronize"); goto out; }
        p->kpatch.txcell = (const unsigned long long *)(q + b_pdef + b_mask);
        p->kpatch.mask = (const unsigned long long *)(q + b_pdef);
        p->kpatch.pdef = (const float *)q;
        p->kpatch.num_patch = (uint32_t)npatch;
        p->kpatch.num_img = n_tx;
        p->kpatch.hmax = hmax;
        p->kpatch.ro_rx = ro_rx;
        p->kpatch.ro_img = ro_img;
    }
out:
    if (d_tmp) hrt_hip_free(d_tmp);
    free(pdef); free(nuv); free(ptri); free(apex);
    if (rc && p->d_patch) { hrt_hip_free(p->d_patch); p->d_patch = NULL; memset(&p->kpatch, 0, sizeof p->kpatch); }
    return rc;
}

int hrt_problem_create(const Scene *scene, const Vec3 *rx_pos, const Vec3 *tx_pos,
                       const Vec3 *rx_vel, const Vec3 *tx_vel, float f_ghz, size_t num_rx,
                       size_t num_tx, int device, hrt_problem **out)
{
    /* a problem made through the device API is traced many times: the direction tables pay */
    return hrt_problem_create_for(scene, rx_pos, tx_pos, rx_vel, tx_vel, f_ghz, num_rx, num_tx, device, UINT64_MAX, out);
}

/* `rays_hint`: how many rays (all TX together) the problem will trace in its lifetime, as far as the
 * caller knows.  The per-RX / per-TX direction tables cost ~3 ms to build (C3: 5 apexes) and save
 * ~0.03 us per ray: a one-shot drop-in call builds them only from HRT_RXT_MIN_RAYS (default 2^26) on. */
int hrt_problem_create_for(const Scene *scene, const Vec3 *rx_pos, const Vec3 *tx_pos,
                           const Vec3 *rx_vel, const Vec3 *tx_vel, float f_ghz, size_t num_rx,
                           size_t num_tx, int device, uint64_t rays_hint, hrt_problem **out)
{
    if (!scene || !rx_pos || !tx_pos || !rx_vel || !tx_vel || !out)
        return hrt_fail(HRT_E_INVALID, "hrt_problem_create: NULL argument");
    if (num_rx == 0 || num_tx == 0 || num_rx > 65535 || num_tx > 65535)
        return hrt_fail(HRT_E_INVALID, "hrt_problem_create: num_rx/num_tx must be in 1..65535");
    if (!(f_ghz > 0.f)) return hrt_fail(HRT_E_INVALID, "carrier frequency must be > 0 GHz");
    if (scene->num_meshes == 0) return hrt_fail(HRT_E_INVALID, "scene has no meshes");

    uint64_t T = 0;
    for (uint32_t i = 0; i < scene->num_meshes; ++i) {
        const Mesh *m = &scene->meshes[i];
        if (m->material_index >= HRT_NUM_MATERIALS)
            return hrt_fail(HRT_E_INVALID, "mesh %u: material_index %u out of range (0..16)", i,
                            m->material_index);
        for (uint64_t k = 0; k < (uint64_t)m->num_triangles * 3; ++k)
            if (m->is[k] >= m->num_vertices)
                return hrt_fail(HRT_E_INVALID, "mesh %u: vertex index %u >= num_vertices %u", i,
                                m->is[k], m->num_vertices);
        T += m->num_triangles;
    }
    if (T > 0x7fffffffu) return hrt_fail(HRT_E_CAPACITY, "too many triangles");

    hrt_problem *p = (hrt_problem *)calloc(1, sizeof *p);
    if (!p) return hrt_fail(HRT_E_NOMEM, "out of host memory");
    {   /* developer / test switches, once per problem (tune.c) */
        const int rct = hrt_tune_load(&p->tune);
        if (rct) { free(p); return rct; }
    }
    p->device = device;
    p->num_tri = (uint32_t)T;
    p->num_mesh = scene->num_meshes;
    p->num_rx = (uint32_t)num_rx;
    p->num_tx = (uint32_t)num_tx;
    p->f_ghz = f_ghz;
    /* src/compute_paths.c:483-488: f_hz = f_GHz * 1e9 (double product stored to float), then
     * float arithmetic left to right */
    float f_hz = (float)((double)f_ghz * 1e9);
    p->fsl_mult = 4.f * HRT_PI_F * f_hz / HRT_C_F;
    p->dop_mult = f_hz / HRT_C_F;

    p->h_tri = (float *)calloc((size_t)(T ? T : 1) * HRT_TRI_FLOATS, sizeof(float));
    p->h_mesh = (float *)calloc((size_t)p->num_mesh * HRT_MESH_FLOATS, sizeof(float));
    p->h_mat = (float *)calloc(HRT_NUM_MATERIALS * HRT_MAT_FLOATS, sizeof(float));
    p->h_tri_mesh = (uint32_t *)malloc((size_t)(T ? T : 1) * sizeof(uint32_t));
    p->h_tri_face = (uint32_t *)malloc((size_t)(T ? T : 1) * sizeof(uint32_t));
    if (!p->h_tri || !p->h_mesh || !p->h_mat || !p->h_tri_mesh || !p->h_tri_face) {
        hrt_problem_destroy(p);
        return hrt_fail(HRT_E_NOMEM, "out of host memory");
    }

    /* flatten: one row per triangle, (mesh, face) order = the reference's loop order, which
     * is what resolves equal-distance ties (src/compute_paths.c:253-275) */
    uint32_t j = 0;
    for (uint32_t i = 0; i < scene->num_meshes; ++i) {
        const Mesh *m = &scene->meshes[i];
        for (uint32_t f = 0; f < m->num_triangles; ++f, ++j) {
            Vec3 v1 = m->vs[m->is[3 * f]], v2 = m->vs[m->is[3 * f + 1]],
                 v3 = m->vs[m->is[3 * f + 2]];
            Vec3 e1 = v_sub(v2, v1), e2 = v_sub(v3, v1);
            Vec3 n = unit_normal(e1, e2);
            float *row = p->h_tri + (size_t)j * HRT_TRI_FLOATS;
            row[0] = v1.x; row[1] = v1.y; row[2] = v1.z;
            row[3] = e1.x; row[4] = e1.y; row[5] = e1.z;
            row[6] = e2.x; row[7] = e2.y; row[8] = e2.z;
            row[9] = n.x; row[10] = n.y; row[11] = n.z;
            memcpy(&row[19], &i, 4);   /* mesh id: last word of the row */
            /* lengths for the packet-culling tolerances, rounded UP (they scale error bounds) */
            {
                Vec3 e3 = v_sub(e2, e1);
                Vec3 c = {e1.y * e2.z - e1.z * e2.y, e1.z * e2.x - e1.x * e2.z, e1.x * e2.y - e1.y * e2.x};
                float *cr = row + 15;   /* |e1| |e2| |e2-e1| |e1 x e2| */
                cr[0] = sqrtf(e1.x * e1.x + e1.y * e1.y + e1.z * e1.z) * 1.000001f;
                cr[1] = sqrtf(e2.x * e2.x + e2.y * e2.y + e2.z * e2.z) * 1.000001f;
                cr[2] = sqrtf(e3.x * e3.x + e3.y * e3.y + e3.z * e3.z) * 1.000001f;
                cr[3] = sqrtf(c.x * c.x + c.y * c.y + c.z * c.z) * 1.000001f;
                /* triangle-only parts of the culling tolerances (hrt_kernels.hip packet_culls),
                 * rounded up: E_d = 16 eps |e1||e2|, 4 eps a_N with a_N = 1.0001 |N| + E_d, and
                 * c_w = 8 eps a_N + 2 E_d + 4e-6 |N| */
                const double eps = 1.1920928955078125e-07, up = 1.000001;
                const double Ed = 16.0 * eps * (double)cr[0] * (double)cr[1];
                const double c2 = 4.0 * eps * (1.0001 * (double)cr[3] + Ed);
                row[12] = (float)(Ed * up);
                row[13] = (float)(c2 * up);
                row[14] = (float)((2.0 * c2 + 2.0 * Ed + 4e-6 * (double)cr[3]) * up);
            }
            p->h_tri_mesh[j] = i;
            p->h_tri_face[j] = f;
        }
        float *mr = p->h_mesh + (size_t)i * HRT_MESH_FLOATS;
        mr[0] = m->velocity.x; mr[1] = m->velocity.y; mr[2] = m->velocity.z;
        memcpy(&mr[3], &m->material_index, 4);
        /* eta only for materials the scene uses, like the reference */
        hrt_eta *e = &p->eta[m->material_index];
        hrt_material_eta(m->material_index, f_ghz, e);
        float *mt = p->h_mat + (size_t)m->material_index * HRT_MAT_FLOATS;
        memcpy(mt, e, 12 * sizeof(float));
        mt[12] = hrt_materials[m->material_index].s;
        mt[13] = (float)hrt_materials[m->material_index].s1_alpha;
    }

    /* ---- acceleration structure: Morton order of the rows (the reference's loop order is kept
     * in accel.orig for the tie-break), leaf spheres, guard numbers, inner levels, plane tree ---- */
    {
        int rc2 = hrt_accel_order(&p->accel, p->h_tri, (uint32_t)T, !p->tune.no_reorder);
        if (rc2) { hrt_problem_destroy(p); return rc2; }
        float *rows = (float *)malloc((size_t)(T ? T : 1) * HRT_TRI_FLOATS * sizeof(float));
        uint32_t *tm = (uint32_t *)malloc((size_t)(T ? T : 1) * 4), *tf = (uint32_t *)malloc((size_t)(T ? T : 1) * 4);
        if (!rows || !tm || !tf) {
            free(rows); free(tm); free(tf);
            hrt_problem_destroy(p);
            return hrt_fail(HRT_E_NOMEM, "out of host memory");
        }
        for (uint32_t k = 0; k < (uint32_t)T; ++k) {
            const uint32_t o = p->accel.orig[k];
            memcpy(rows + (size_t)k * HRT_TRI_FLOATS, p->h_tri + (size_t)o * HRT_TRI_FLOATS, HRT_TRI_FLOATS * sizeof(float));
            tm[k] = p->h_tri_mesh[o];
            tf[k] = p->h_tri_face[o];
        }
        free(p->h_tri); free(p->h_tri_mesh); free(p->h_tri_face);
        p->h_tri = rows; p->h_tri_mesh = tm; p->h_tri_face = tf;
        rc2 = hrt_accel_build(&p->accel, p->h_tri, &p->tune);
        if (rc2) { hrt_problem_destroy(p); return rc2; }
    }

    /* one device blob (every part 256-B aligned): tri | mesh | mat | rx_pos | tx_pos | rx_vel |
     * tx_vel | orig | tg | leaf | inner levels | plane-tree levels | plane index | plane records */
    const hrt_accel *A = &p->accel;
    const void *src[32];
    uint64_t len[32], offs[32];
    int np = 0;
#define PART(ptr, bytes) (src[np] = (ptr), len[np] = (uint64_t)(bytes), np++)
    const int i_tri = PART(p->h_tri, (uint64_t)(T ? T : 1) * HRT_TRI_FLOATS * 4);
    const int i_mesh = PART(p->h_mesh, (uint64_t)p->num_mesh * HRT_MESH_FLOATS * 4);
    const int i_mat = PART(p->h_mat, HRT_NUM_MATERIALS * HRT_MAT_FLOATS * 4);
    const int i_rxp = PART(rx_pos, (uint64_t)num_rx * 12);
    const int i_txp = PART(tx_pos, (uint64_t)num_tx * 12);
    const int i_rxv = PART(rx_vel, (uint64_t)num_rx * 12);
    const int i_txv = PART(tx_vel, (uint64_t)num_tx * 12);
    const int i_orig = PART(A->orig, (uint64_t)(T ? T : 1) * 4);
    const int i_tg = PART(A->tg, (uint64_t)(T ? T : 1) * 8);
    const int i_leaf = PART(A->leaf, (uint64_t)(A->num_leaf ? A->num_leaf : 1) * HRT_NODE_FLOATS * 4);
    int i_node[HRT_ACCEL_MAX_LEVELS], i_pl[HRT_ACCEL_MAX_LEVELS], i_pli = -1, i_plr = -1;
    for (uint32_t k = 0; k < A->num_levels; ++k)
        i_node[k] = PART(A->node[k], (uint64_t)A->node_count[k] * HRT_NODE_FLOATS * 4);
    for (uint32_t k = 0; k < A->pl_levels; ++k)
        i_pl[k] = PART(A->pl_node[k], (uint64_t)A->pl_count[k] * HRT_NODE_FLOATS * 4);
    int i_fine = -1;
    if (A->planes) {
        i_pli = PART(A->pl_index, (uint64_t)A->pl_num_leaf * 64 * 4);
        i_plr = PART(A->pl_rec, (uint64_t)A->pl_num_leaf * 64 * HRT_NODE_FLOATS * 4);
    }
    if (A->fine) i_fine = PART(A->fine, (uint64_t)A->num_fine * HRT_NODE_FLOATS * 4);
    int i_inv = -1;
    uint32_t *inv = NULL;
    if (A->fine) {   /* original index -> row (the wide kernels carry the original index in their keys) */
        inv = (uint32_t *)malloc((size_t)(T ? T : 1) * 4);
        if (!inv) { hrt_problem_destroy(p); return hrt_fail(HRT_E_NOMEM, "out of host memory"); }
        for (uint32_t k = 0; k < (uint32_t)T; ++k) inv[A->orig[k]] = k;
        i_inv = PART(inv, (uint64_t)(T ? T : 1) * 4);
    }
#undef PART
    uint64_t total = 0;
    for (int k = 0; k < np; ++k) { offs[k] = total; total += round_up(len[k] ? len[k] : 1, 256); }
    int rc;
    if ((rc = hrt_hip_set_device(device)) != 0) {
        free(inv);
        hrt_problem_destroy(p);
        return hrt_fail_hip(rc, "hipSetDevice");
    }
    if ((rc = hrt_hip_malloc(&p->d_blob, total)) != 0) {
        free(inv);
        p->d_blob = NULL;
        hrt_problem_destroy(p);
        return hrt_fail_hip(rc, "hipMalloc(problem)");
    }
    {   /* the word a fused launch sets when it gives up waiting (best effort: without it the error word alone tells) */
        void *hf = NULL, *df = NULL;
        if (hrt_hip_host_malloc_mapped(&hf, &df, 64) == 0) {
            p->h_fuse_flag = (uint32_t *)hf;
            p->d_fuse_flag = (uint32_t *)df;
            p->h_fuse_flag[0] = p->h_fuse_flag[1] = 0u;
        }
    }
    uint8_t *b = (uint8_t *)p->d_blob;
    for (int k = 0; k < np; ++k)
        if (len[k] && (rc = hrt_hip_h2d(b + offs[k], src[k], len[k]))) {
            free(inv);
            hrt_problem_destroy(p);
            return hrt_fail_hip(rc, "hipMemcpy(problem)");
        }
    free(inv);
    p->d_tri = (const float *)(b + offs[i_tri]);
    p->d_mesh = (const float *)(b + offs[i_mesh]);
    p->d_mat = (const float *)(b + offs[i_mat]);
    p->d_rx_pos = (const float *)(b + offs[i_rxp]);
    p->d_tx_pos = (const float *)(b + offs[i_txp]);
    p->d_rx_vel = (const float *)(b + offs[i_rxv]);
    p->d_tx_vel = (const float *)(b + offs[i_txv]);
    hrt_kaccel *ka = &p->kaccel;
    memset(ka, 0, sizeof *ka);
    ka->orig = (const uint32_t *)(b + offs[i_orig]);
    ka->tg = (const float *)(b + offs[i_tg]);
    ka->leaf = (const float *)(b + offs[i_leaf]);
    ka->num_leaf = A->num_leaf;
    ka->big = A->big ? 1u : 0u;
    ka->num_levels = A->num_levels;
    for (uint32_t k = 0; k < A->num_levels; ++k) {
        ka->node_count[k] = A->node_count[k];
        ka->node[k] = (const float *)(b + offs[i_node[k]]);
    }
    ka->pl_levels = A->pl_levels;
    for (uint32_t k = 0; k < A->pl_levels; ++k) {
        ka->pl_count[k] = A->pl_count[k];
        ka->pl_node[k] = (const float *)(b + offs[i_pl[k]]);
    }
    if (A->planes) {
        ka->pl_index = (const uint32_t *)(b + offs[i_pli]);
        ka->pl_rec = (const float *)(b + offs[i_plr]);
    }
    if (A->fine) {
        ka->fine = (const float *)(b + offs[i_fine]);
        ka->num_fine = A->num_fine;
        p->d_inv = (const uint32_t *)(b + offs[i_inv]);
    }
    {   /* bounding box of the finite vertices (cells of the re-sort keys), and whether to re-sort */
        double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t jj = 0; jj < p->num_tri; ++jj) {
            const float *r = p->h_tri + (size_t)jj * HRT_TRI_FLOATS;
            for (int v = 0; v < 3; ++v)
                for (int k = 0; k < 3; ++k) {
                    const double x = (double)r[k] + (v == 1 ? (double)r[3 + k] : (v == 2 ? (double)r[6 + k] : 0.0));
                    if (!isfinite(x)) continue;
                    if (x < lo[k]) lo[k] = x;
                    if (x > hi[k]) hi[k] = x;
                }
        }
        for (int k = 0; k < 3; ++k) {
            p->scene_lo[k] = isfinite(lo[k]) ? (float)lo[k] : 0.f;
            p->scene_hi[k] = isfinite(hi[k]) ? (float)hi[k] : 1.f;
        }

        /* default: tables beyond HRT_SORT_MIN_TRI triangles -- there a wave that scattered costs a
         * staged pass over the whole table, and the sort (a few passes over the live list) is cheap
         * against it; on small tables (C3) the sort would cost more than the whole trace */
        p->sort_rays = p->tune.sort_rays >= 0 ? (p->tune.sort_rays != 0) : (p->num_tri > HRT_SORT_MIN_TRI);
        uint32_t txb = 0;
        while ((1u << txb) < p->num_tx) ++txb;
        if (24u + txb > 32u) p->sort_rays = 0;   /* 15 bits of cell + up to 9 of direction + TX */
    }
    {
        /* (the per-ray masks of tables of <= 64 triangles need no host pass over the lists: ~0.2 ms) */
        uint64_t min_rays = p->num_tri <= 64u ? 1ull << 18 : 1ull << 26;
        if (p->tune.rxt_min_rays != UINT64_MAX) min_rays = p->tune.rxt_min_rays;
        if (rays_hint >= min_rays) {
            int rcx = rxt_build(p, rx_pos, tx_pos);
            if (!rcx) rcx = patch_build(p, rx_pos, tx_pos);
            if (rcx) { hrt_problem_destroy(p); return rcx; }
        }
    }
    *out = p;
    return HRT_OK;
}

uint32_t hrt_problem_num_triangles(const hrt_problem *p) { return p->num_tri; }
uint32_t hrt_problem_num_rx(const hrt_problem *p) { return p->num_rx; }
uint32_t hrt_problem_num_tx(const hrt_problem *p) { return p->num_tx; }
int hrt_problem_device(const hrt_problem *p) { return p->device; }

int hrt_problem_eta_table(const hrt_problem *p, float *out)
{
    memcpy(out, p->eta, sizeof p->eta);
    return HRT_OK;
}

int hrt_problem_normals(const hrt_problem *p, float *out)
{
    /* in the reference's (mesh, face) loop order, whatever the order of the device table */
    for (uint32_t j = 0; j < p->num_tri; ++j)
        memcpy(out + 3 * (size_t)j, p->h_tri + (size_t)p->accel.newidx[j] * HRT_TRI_FLOATS + 9, 12);
    return HRT_OK;
}

int hrt_problem_tri_order(const hrt_problem *p, uint32_t *orig_of_row)
{
    memcpy(orig_of_row, p->accel.orig, (size_t)p->num_tri * 4);
    return HRT_OK;
}

int hrt_problem_tri_ids(const hrt_problem *p, uint32_t *mesh_out, uint32_t *face_out)
{
    memcpy(mesh_out, p->h_tri_mesh, (size_t)p->num_tri * 4);
    memcpy(face_out, p->h_tri_face, (size_t)p->num_tri * 4);
    return HRT_OK;
}

/* ------------------------------------------------------------------ shards */

static uint32_t shard_chunk(const hrt_shard *s) { return s->chunk ? s->chunk : 4096u; }

static int shard_check(const hrt_shard *s)
{
    if (!s || s->count == 0 || s->rank >= s->count || s->num_paths == 0)
        return hrt_fail(HRT_E_INVALID, "bad shard (rank %u of %u, %llu paths)", s ? s->rank : 0,
                        s ? s->count : 0, s ? (unsigned long long)s->num_paths : 0ull);
    if (shard_chunk(s) % 64u)
        return hrt_fail(HRT_E_INVALID, "shard chunk %u is not a multiple of 64", shard_chunk(s));
    if (s->num_bounces == 0 || s->num_bounces > 65535u)   /* (the reference's loop has no cap: src/compute_paths.c:591) */
        return hrt_fail(HRT_E_INVALID, "num_bounces must be in 1..65535");
    return HRT_OK;
}

uint64_t hrt_shard_num_local(const hrt_shard *s)
{
    if (!s || s->count == 0 || s->rank >= s->count) return 0;
    const uint64_t ch = shard_chunk(s), N = s->num_paths;
    const uint64_t n_chunks = (N + ch - 1) / ch, rem = N % ch;
    if (n_chunks <= s->rank) return 0;
    const uint64_t mine = (n_chunks - s->rank + s->count - 1) / s->count;
    uint64_t n = mine * ch;
    const int owns_last = ((n_chunks - 1) % s->count) == s->rank;
    if (owns_last && rem) n -= ch - rem;
    return n;
}

uint64_t hrt_shard_global_path(const hrt_shard *s, uint64_t i)
{
    const uint64_t ch = shard_chunk(s);
    return ((i / ch) * s->count + s->rank) * ch + i % ch;
}

/* ------------------------------------------------------------------ launch directions */

typedef struct {
    const hrt_shard *s;
    float *out;
    uint64_t i0, i1;
} dirs_job;

/* src/compute_paths.c:444-451.  k, phi, theta are FLOAT; acos/cos/sin are the DOUBLE libm
 * functions applied to them; the products are rounded to float on store. */
static void launch_dir_host(const hrt_shard *s, uint64_t i, float *out3)
{
    const float n_f = (float)s->num_paths;   /* size_t -> float in "2.f * k / num_paths" */
    const float golden = HRT_PI_F * (1.f + sqrtf(5.f));
    const uint64_t p = hrt_shard_global_path(s, i);
    const float k = (float)p + .5f;
    const float phi = (float)acos((double)(1.f - 2.f * k / n_f));
    const float theta = golden * k;
    const double sp = sin((double)phi);
    out3[0] = (float)(cos((double)theta) * sp);
    out3[1] = (float)(sin((double)theta) * sp);
    out3[2] = (float)cos((double)phi);
}

static void *dirs_worker(void *arg)
{
    dirs_job *jb = (dirs_job *)arg;
    for (uint64_t i = jb->i0; i < jb->i1; ++i) launch_dir_host(jb->s, i, jb->out + 3 * i);
    return NULL;
}

int hrt_launch_dirs_host(const hrt_shard *s, float *out, int num_threads)
{
    if (!s || !out || s->count == 0 || s->rank >= s->count || s->num_paths == 0)
        return hrt_fail(HRT_E_INVALID, "hrt_launch_dirs_host: bad shard");
    const uint64_t n = hrt_shard_num_local(s);
    if (num_threads <= 0) {
        long c = sysconf(_SC_NPROCESSORS_ONLN);
        num_threads = c > 0 ? (int)c : 1;
    }
    if (num_threads > 64) num_threads = 64;
    if ((uint64_t)num_threads > n / 4096 + 1) num_threads = (int)(n / 4096 + 1);
    pthread_t th[64];
    dirs_job jb[64];
    int started = 0;
    for (int t = 0; t < num_threads; ++t) {
        jb[t].s = s;
        jb[t].out = out;
        jb[t].i0 = n * (uint64_t)t / (uint64_t)num_threads;
        jb[t].i1 = n * (uint64_t)(t + 1) / (uint64_t)num_threads;
        if (t + 1 == num_threads || pthread_create(&th[t], NULL, dirs_worker, &jb[t]) != 0) {
            /* last slice (or a thread that would not start): run here, to the end */
            jb[t].i1 = n;
            dirs_worker(&jb[t]);
            break;
        }
        ++started;
    }
    for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
    return HRT_OK;
}

/* Launch directions generated ON THE DEVICE into d_dirs ([num_local][3] floats), bit-identical
 * to hrt_launch_dirs_host: the device evaluates the reference's formula with its own double
 * math library, flags every value whose rounding to float could depend on the last bits of
 * that library, and the flagged rays (about one in a million) are recomputed here with the host
 * libm and patched in.  Blocks until done.  *num_patched (may be NULL) reports the list length. */
#define HRT_DIRS_FIX_CAP 65536u
int hrt_launch_dirs_device(const hrt_shard *s, float *d_dirs, int device, void *stream,
                           uint64_t *num_patched)
{
    if (!s || !d_dirs || s->count == 0 || s->rank >= s->count || s->num_paths == 0)
        return hrt_fail(HRT_E_INVALID, "hrt_launch_dirs_device: bad argument");
    const uint32_t ch = shard_chunk(s);
    if (ch % 64u) return hrt_fail(HRT_E_INVALID, "shard chunk %u is not a multiple of 64", ch);
    const uint64_t n = hrt_shard_num_local(s);
    HRT_HIP(hrt_hip_set_device(device), "hipSetDevice");
    void *d_fix = NULL;
    HRT_HIP(hrt_hip_malloc(&d_fix, (uint64_t)(HRT_DIRS_FIX_CAP + 1) * 4), "hipMalloc(fix list)");
    int rc = HRT_OK, e;
    uint32_t *h_fix = (uint32_t *)malloc((size_t)(HRT_DIRS_FIX_CAP + 1) * 4);
    if (!h_fix) rc = hrt_fail(HRT_E_NOMEM, "out of host memory");
    if (!rc && (e = hrt_hip_memset_async(d_fix, 0, 4, stream))) rc = hrt_fail_hip(e, "hipMemsetAsync");
    if (!rc && (e = hrt_hip_launch_dirs(s->num_paths, s->rank, s->count, ch, n, d_dirs, (uint32_t *)d_fix,
                                        (uint32_t *)d_fix + 1, HRT_DIRS_FIX_CAP, stream)))
        rc = hrt_fail_hip(e, "hrt_launch_dirs_kernel");
    if (!rc && (e = hrt_hip_stream_sync(stream))) rc = hrt_fail_hip(e, "hipStreamSynchronize");
    if (!rc && (e = hrt_hip_d2h(h_fix, d_fix, 4))) rc = hrt_fail_hip(e, "hipMemcpy D2H");
    if (!rc && h_fix[0] > HRT_DIRS_FIX_CAP)
        rc = hrt_fail(HRT_E_CAPACITY, "launch-direction fix list overflow (%u): use the host generator", h_fix[0]);
    if (!rc && h_fix[0] && (e = hrt_hip_d2h(h_fix + 1, (uint32_t *)d_fix + 1, (uint64_t)h_fix[0] * 4)))
        rc = hrt_fail_hip(e, "hipMemcpy D2H");
    for (uint32_t k = 0; !rc && k < h_fix[0]; ++k) {
        float v[3];
        launch_dir_host(s, h_fix[1 + k], v);
        if ((e = hrt_hip_h2d(d_dirs + 3 * (uint64_t)h_fix[1 + k], v, 12))) rc = hrt_fail_hip(e, "hipMemcpy H2D");
    }
    if (!rc && num_patched) *num_patched = h_fix[0];
    free(h_fix);
    hrt_hip_free(d_fix);
    return rc;
}

/* Coherent launch order: z-bands of about sqrt(n/128) rows, serpentine in azimuth, so 64
 * consecutive positions are a roughly square patch of the sphere.  Keys are 32 bits (12 bits of
 * band, 20 of azimuth); a stable 2-pass LSD radix sort (16 bits per pass) orders them --
 * O(n), ~20 ms for 4M rays, and ties keep index order, so the result is deterministic. */
int hrt_launch_order_host(const hrt_shard *s, const float *dirs, uint32_t *order)
{
    if (!s || !order) return hrt_fail(HRT_E_INVALID, "hrt_launch_order_host: NULL argument");
    const uint64_t n = hrt_shard_num_local(s);
    if (n == 0 || n > 0xffffffffull) return hrt_fail(HRT_E_INVALID, "hrt_launch_order_host: bad shard");
    uint32_t *key = (uint32_t *)malloc(n * sizeof(uint32_t));
    uint32_t *tmp = (uint32_t *)malloc(n * sizeof(uint32_t));
    uint32_t *cnt = (uint32_t *)calloc(65537, sizeof(uint32_t));
    if (!key || !tmp || !cnt) {
        free(key); free(tmp); free(cnt);
        return hrt_fail(HRT_E_NOMEM, "out of host memory");
    }
    uint32_t nbands = (uint32_t)sqrt((double)n / 128.0);
    if (nbands < 1) nbands = 1;
    if (nbands > 4095) nbands = 4095;
    const float n_f = (float)s->num_paths, golden = HRT_PI_F * (1.f + sqrtf(5.f));
    for (uint64_t i = 0; i < n; ++i) {
        float dd[3];
        const float *d = dirs ? dirs + 3 * i : dd;
        if (!dirs) {
            /* no directions at hand (they were generated on the device): the key needs only
             * the polar coordinate z = 1 - 2k/N and the azimuth theta mod 2 pi, both available
             * without libm (fmod is exact); any monotone proxy gives the same order quality */
            const float k = (float)hrt_shard_global_path(s, i) + .5f;
            const double th = fmod((double)(golden * k), 6.283185307179586477);
            const double thq = th * (4.0 / 6.283185307179586477);   /* quadrant coordinate 0..4 */
            const int qd = (int)thq;
            const double fr = thq - qd;   /* diamond-angle-like proxy of (cos, sin) */
            dd[0] = (qd == 0 || qd == 3) ? (float)(qd == 0 ? 1.0 - fr : fr) : (float)(qd == 1 ? -fr : -(1.0 - fr));
            dd[1] = (qd == 0 || qd == 1) ? (float)(qd == 0 ? fr : 1.0 - fr) : (float)(qd == 2 ? -fr : -(1.0 - fr));
            dd[2] = 1.f - 2.f * k / n_f;
        }
        double z = d[2] > 1.f ? 1.0 : (d[2] < -1.f ? -1.0 : (double)d[2]);
        uint32_t band = (uint32_t)(0.5 * (1.0 - z) * nbands);
        if (band >= nbands) band = nbands - 1;
        /* "diamond angle": monotone in the true azimuth, no atan2 -- only the ORDER matters */
        const double ax = fabs((double)d[0]), ay = fabs((double)d[1]);
        double t = (ax + ay > 0.0) ? ay / (ax + ay) : 0.0;            /* 0..1 within a quadrant */
        double az = d[0] >= 0.f ? (d[1] >= 0.f ? t : 4.0 - t) : (d[1] >= 0.f ? 2.0 - t : 2.0 + t);
        az *= 0.25;
        if (!(az >= 0.0)) az = 0.0;
        if (az > 0.999999) az = 0.999999;
        if (band & 1u) az = 0.999999 - az;   /* serpentine */
        key[i] = (band << 20) | (uint32_t)(az * 1048576.0);
    }
    /* pass 1: by the low 16 key bits, index order -> tmp holds indices */
    for (uint64_t i = 0; i < n; ++i) cnt[(key[i] & 0xffffu) + 1]++;
    for (uint32_t k = 0; k < 65536; ++k) cnt[k + 1] += cnt[k];
    for (uint64_t i = 0; i < n; ++i) tmp[cnt[key[i] & 0xffffu]++] = (uint32_t)i;
    /* pass 2: by the high 16 bits, stable over pass 1's order */
    memset(cnt, 0, 65537 * sizeof(uint32_t));
    for (uint64_t i = 0; i < n; ++i) cnt[(key[i] >> 16) + 1]++;
    for (uint32_t k = 0; k < 65536; ++k) cnt[k + 1] += cnt[k];
    for (uint64_t i = 0; i < n; ++i) {
        const uint32_t idx = tmp[i];
        order[cnt[key[idx] >> 16]++] = idx;
    }
    free(key); free(tmp); free(cnt);
    return HRT_OK;
}

/* The same job on the device (hrt_launch_order_kernel): d_order[i] = local ray launched at
 * position i.  The band of a ray is monotone in its local index, so the host only finds the band
 * boundaries (bisection: bands x log n evaluations), cuts bands longer than 32768 rays, uploads
 * the segment table and launches one workgroup per segment.  Blocks until done.  The order is not
 * the host function's (17 instead of 20 azimuth bits, bands cut at 32768) -- any coherent order
 * serves; results never depend on it. */
static uint32_t order_band(const hrt_shard *s, uint64_t i, uint32_t nbands)
{
    const float n_f = (float)s->num_paths;
    const float k = (float)hrt_shard_global_path(s, i) + .5f;
    const float z = 1.f - 2.f * k / n_f;
    const double zc = z > 1.f ? 1.0 : (z < -1.f ? -1.0 : (double)z);
    uint32_t band = (uint32_t)(0.5 * (1.0 - zc) * nbands);
    return band >= nbands ? nbands - 1 : band;
}

int hrt_launch_order_device(const hrt_shard *s, uint32_t *d_order, int device, void *stream)
{
    if (!s || !d_order || s->count == 0 || s->rank >= s->count || s->num_paths == 0)
        return hrt_fail(HRT_E_INVALID, "hrt_launch_order_device: bad argument");
    const uint32_t ch = shard_chunk(s);
    if (ch % 64u) return hrt_fail(HRT_E_INVALID, "shard chunk %u is not a multiple of 64", ch);
    const uint64_t n = hrt_shard_num_local(s);
    if (n == 0 || n > 0xffffffffull) return hrt_fail(HRT_E_INVALID, "hrt_launch_order_device: bad shard");
    uint32_t nbands = (uint32_t)sqrt((double)n / 128.0);
    if (nbands < 1) nbands = 1;
    if (nbands > 4095) nbands = 4095;
    const uint64_t max_seg = (uint64_t)nbands + n / 32768u + 2;
    uint32_t *seg = (uint32_t *)malloc((max_seg + 1) * 2 * sizeof(uint32_t));
    if (!seg) return hrt_fail(HRT_E_NOMEM, "out of host memory");
    uint32_t *seg_start = seg, *seg_band = seg + max_seg + 1;
    uint64_t nseg = 0, pos = 0;
    while (pos < n) {
        const uint32_t b = order_band(s, pos, nbands);
        uint64_t lo = pos + 1, hi = n;      /* first index of a later band */
        while (lo < hi) {
            const uint64_t mid = (lo + hi) / 2;
            if (order_band(s, mid, nbands) > b) hi = mid; else lo = mid + 1;
        }
        for (uint64_t q = pos; q < lo; q += 32768u) {
            seg_start[nseg] = (uint32_t)q;
            seg_band[nseg] = b;
            ++nseg;
        }
        pos = lo;
    }
    seg_start[nseg] = (uint32_t)n;
    seg_band[nseg] = 0;
    int rc = HRT_OK, e;
    void *d_seg = NULL;
    if ((e = hrt_hip_set_device(device))) rc = hrt_fail_hip(e, "hipSetDevice");
    if (!rc && (e = hrt_hip_malloc(&d_seg, (nseg + 1) * 8))) rc = hrt_fail_hip(e, "hipMalloc(segments)");
    if (!rc && (e = hrt_hip_h2d(d_seg, seg_start, (nseg + 1) * 4))) rc = hrt_fail_hip(e, "hipMemcpy H2D");
    if (!rc && (e = hrt_hip_h2d((uint32_t *)d_seg + nseg + 1, seg_band, (nseg + 1) * 4))) rc = hrt_fail_hip(e, "hipMemcpy H2D");
    if (!rc && (e = hrt_hip_launch_order((const uint32_t *)d_seg, (const uint32_t *)d_seg + nseg + 1, (uint32_t)nseg,
                                         s->num_paths, s->rank, s->count, ch, d_order, stream)))
        rc = hrt_fail_hip(e, "hrt_launch_order_kernel");
    if (!rc && (e = hrt_hip_stream_sync(stream))) rc = hrt_fail_hip(e, "hipStreamSynchronize");
    if (d_seg) hrt_hip_free(d_seg);
    free(seg);
    return rc;
}

/* ------------------------------------------------------------------ layout + trace */

int hrt_layout_query(const hrt_problem *p, const hrt_shard *s, hrt_layout *L)
{
    if (!p || !L) return hrt_fail(HRT_E_INVALID, "hrt_layout_query: NULL argument");
    int rc = shard_check(s);
    if (rc) return rc;
    const uint64_t n0 = (uint64_t)p->num_tx * hrt_shard_num_local(s);
    if (n0 == 0) return hrt_fail(HRT_E_INVALID, "shard %u of %u is empty", s->rank, s->count);
    /* the kernels address the field arrays of a block by 32-bit byte offsets from the block's
     * buffer descriptor: HRT_HIT_FIELDS * cap * 4 < 2^32 */
    if (n0 > 0xffffffffull / (4u * HRT_HIT_FIELDS) - 2u * HRT_BLOCK)
        return hrt_fail(HRT_E_CAPACITY, "num_tx * local rays = %llu: more than %llu rays in one shard "
                        "(32-bit offsets inside a block of field arrays); use more shards",
                        (unsigned long long)n0,
                        (unsigned long long)(0xffffffffull / (4u * HRT_HIT_FIELDS) - 2u * HRT_BLOCK));
    const uint64_t nb = s->num_bounces, cap = round_up(n0, HRT_BLOCK);
    memset(L, 0, sizeof *L);
    L->cap = cap;
    uint64_t off = 0;
    /* counts[nb + 2], then (from byte 256) one work-unit counter per launch for the trace kernel's
     * dynamic unit distribution on big tables; zeroed together at the start of every trace */
    L->off_counts = off; off += HRT_CNT_BYTES(s->num_bounces);
    /* survivor counts per super-chunk (HRT_SUPER_CHUNKS chunks of 256 entries) and bounce: directly behind
     * the counts, zeroed with them at the start of every trace */
    L->num_super = cap / HRT_BLOCK / HRT_SUPER_CHUNKS + 1;
    L->off_super_cnt = off; off += round_up(nb * L->num_super * 4, 256);
    /* status words of the fused kernels' chained scan, one per chunk and launch: zeroed with the counts */
    {
        const uint64_t lbc = round_up(cap / HRT_BLOCK + 1, 64);
        L->lb_stride = lbc + lbc / 64 + 128;   /* u32 words: per chunk, per group of 64, per supergroup of 4096 */
    }
    L->off_lb = off;     off += round_up((nb + 1) * L->lb_stride * 4, 256);
    L->off_los = off;    off += round_up((uint64_t)p->num_rx * p->num_tx * HRT_LOS_FLOATS * 4, 256);
    L->off_hits = off;   L->hit_block_bytes = (uint64_t)HRT_HIT_FIELDS * cap * 4; off += nb * L->hit_block_bytes;
    L->off_recs = off;   L->rec_block_bytes = (uint64_t)p->num_rx * HRT_REC_FIELDS * cap * 4; off += nb * L->rec_block_bytes;
    L->off_masks = off;  off += round_up(nb * p->num_rx * (cap / 64) * 8, 256);
    L->off_chunk_cnt = off; off += round_up((cap / HRT_BLOCK + 1) * 16, 256);   /* one word per wave of a chunk */
    if (p->accel.fine) {
        /* the fine walk's queue of wide packets: half the wave-traces of the largest launch (what does
         * not fit is walked by the pushing wave itself) */
        const uint64_t traces = (cap / 64) * ((uint64_t)p->num_rx + 1);
        L->wide_cap = traces / 2 < 1024 ? 1024 : traces / 2;
        if (p->tune.wide_cap >= 0) L->wide_cap = (uint64_t)p->tune.wide_cap;
        if (L->wide_cap > 0x7fffffffull / 64) L->wide_cap = 0x7fffffffull / 64;
        L->off_wide_q = off;   off += round_up(L->wide_cap * 8 + 8, 256);
        L->off_wide_key = off; off += round_up(L->wide_cap * 64 * 8 + 8, 256);
    }
    L->off_res = off;    off += ((uint64_t)p->num_rx + 1) * 2 * cap * 4;
    if (p->sort_rays) {
        L->off_sort_scratch = off; off += L->hit_block_bytes;
        L->off_sort_keys = off;    off += round_up(4 * cap * 4, 256);
        L->sort_tmp_bytes = round_up(hrt_hip_sort_temp_bytes(cap) + 256, 256);
        L->off_sort_tmp = off;     off += L->sort_tmp_bytes;
    }
    L->total_bytes = off;
    return HRT_OK;
}

struct hrt_timer {
    uint32_t num_bounces;
    int recorded;
    void *ev[2 + 4 * 34];
};

int hrt_timer_create(uint32_t num_bounces, hrt_timer **out)
{
    /* (per-kernel times are recorded for up to 32 bounces: hrt_kernel_times has 33 slots; traces of more bounces run untimed) */
    if (!out || num_bounces == 0 || num_bounces > 32) return hrt_fail(HRT_E_INVALID, "hrt_timer_create: num_bounces must be in 1..32");
    hrt_timer *t = (hrt_timer *)calloc(1, sizeof *t);
    if (!t) return hrt_fail(HRT_E_NOMEM, "out of host memory");
    t->num_bounces = num_bounces;
    const uint32_t n = 2 + 4 * (num_bounces + 1);
    for (uint32_t i = 0; i < n; ++i) {
        int rc = hrt_hip_event_create(&t->ev[i]);
        if (rc) {
            for (uint32_t k = 0; k < i; ++k) hrt_hip_event_destroy(t->ev[k]);
            free(t);
            return hrt_fail_hip(rc, "hipEventCreate");
        }
    }
    *out = t;
    return HRT_OK;
}

void hrt_timer_destroy(hrt_timer *t)
{
    if (!t) return;
    for (uint32_t i = 0; i < 2 + 4 * (t->num_bounces + 1); ++i) hrt_hip_event_destroy(t->ev[i]);
    free(t);
}

int hrt_timer_read(hrt_timer *t, hrt_kernel_times *times)
{
    if (!t || !times || !t->recorded) return hrt_fail(HRT_E_INVALID, "hrt_timer_read: nothing recorded");
    const uint32_t nb = t->num_bounces;
    memset(times, 0, sizeof *times);
    int hip = hrt_hip_event_sync(t->ev[5 + 4 * nb]);
#define STEP(call) do { if (!hip) hip = (call); } while (0)
    STEP(hrt_hip_event_elapsed_ms(t->ev[0], t->ev[1], &times->los_ms));
    for (uint32_t b = 0; b <= nb && !hip; ++b) {
        STEP(hrt_hip_event_elapsed_ms(t->ev[2 + 4 * b], t->ev[4 + 4 * b], &times->records_ms[b]));
        STEP(hrt_hip_event_elapsed_ms(t->ev[4 + 4 * b], t->ev[3 + 4 * b], &times->trace_ms[b]));
        STEP(hrt_hip_event_elapsed_ms(t->ev[3 + 4 * b], t->ev[5 + 4 * b], &times->shade_ms[b]));
    }
#undef STEP
    times->num_bounce_launches = nb + 1;
    if (hip) return hrt_fail_hip(hip, "hrt_timer_read");
    return HRT_OK;
}

static int trace_impl(const hrt_problem *p, const hrt_shard *s, const float *d_dirs,
                      const uint32_t *d_order, void *d_ws, uint64_t ws_bytes, void *stream,
                      hrt_timer *timer, uint32_t flags);

int hrt_trace(const hrt_problem *p, const hrt_shard *s, const float *d_dirs,
              const uint32_t *d_order, void *d_ws, uint64_t ws_bytes, void *stream,
              hrt_kernel_times *times)
{
    if (!times) return trace_impl(p, s, d_dirs, d_order, d_ws, ws_bytes, stream, NULL, 0);
    if (!s) return hrt_fail(HRT_E_INVALID, "hrt_trace: NULL argument");
    hrt_timer *t = NULL;
    int rc = hrt_timer_create(s->num_bounces, &t);
    if (rc) return rc;
    rc = trace_impl(p, s, d_dirs, d_order, d_ws, ws_bytes, stream, t, 0);
    if (!rc) rc = hrt_timer_read(t, times);
    hrt_timer_destroy(t);
    return rc;
}

int hrt_trace_timed(const hrt_problem *p, const hrt_shard *s, const float *d_dirs,
                    const uint32_t *d_order, void *d_ws, uint64_t ws_bytes, void *stream,
                    hrt_timer *timer)
{
    if (!timer) return hrt_fail(HRT_E_INVALID, "hrt_trace_timed: NULL timer");
    return trace_impl(p, s, d_dirs, d_order, d_ws, ws_bytes, stream, timer, 0);
}

int hrt_trace_flags(const hrt_problem *p, const hrt_shard *s, const float *d_dirs,
                    const uint32_t *d_order, void *d_ws, uint64_t ws_bytes, void *stream,
                    hrt_timer *timer, uint32_t flags)
{
    if (flags & ~(uint32_t)HRT_DIRS_IN_LAUNCH_ORDER)
        return hrt_fail(HRT_E_INVALID, "hrt_trace_flags: unknown flag bits 0x%x", flags);
    if ((flags & HRT_DIRS_IN_LAUNCH_ORDER) && !d_order)
        return hrt_fail(HRT_E_INVALID, "HRT_DIRS_IN_LAUNCH_ORDER needs the launch order");
    return trace_impl(p, s, d_dirs, d_order, d_ws, ws_bytes, stream, timer, flags);
}

/* Which launches run as one fused kernel.  HRT_FUSE: unset = launch 0 on every table, whole bounces
 * on tables of at most HRT_FUSE_MAX_TRI triangles (one culling round: there the split into a
 * geometry kernel and a shading kernel only costs traffic and launches); 0 = never (two kernels per
 * launch); 1 = launch 0 only; 2 = every launch, on any table of one culling block (<= 1024). */
static int g_fuse_off, g_chain_off;   /* (ints written once: benign if two threads race to set them) */
void hrt_fuse_disable(void) { g_fuse_off = 1; }
int hrt_fuse_disabled(void) { return g_fuse_off; }
void hrt_chain_disable(void) { g_chain_off = 1; }
int hrt_chain_disabled(void) { return g_chain_off; }
/* which kernels this process has switched off after a timeout (bit 0: fused launches, bit 1: the chain kernel) */
int hrt_fallback_state(void) { return (g_fuse_off ? 1 : 0) | (g_chain_off ? 2 : 0); }
/* what the host does about a void step (error word of its counts): 1 = run it again (something was switched off) */
int hrt_void_step_retry(uint32_t err_word)
{
    int again = 0;
    if ((err_word & HRT_ERR_CHAIN_TIMEOUT) && !g_chain_off) { g_chain_off = 1; again = 1; }
    if ((err_word & HRT_ERR_FUSE_TIMEOUT) && !g_fuse_off) { g_fuse_off = 1; again = 1; }
    return again;
}

static uint32_t fuse_mode(const hrt_problem *p)
{
    /* a fused launch of this process gave up waiting once: the GPU is shared with other fused kernels */
    if (p->h_fuse_flag && ((volatile uint32_t *)p->h_fuse_flag)[0]) g_fuse_off = 1;
    if (p->h_fuse_flag && ((volatile uint32_t *)p->h_fuse_flag)[1]) g_chain_off = 1;
    if (g_fuse_off) return 0u;
    const char *v = getenv("HRT_FUSE");
    const int one_block = p->num_tri <= 1024u;
    if (v && *v) {
        const int m = atoi(v);
        if (m <= 0) return 0u;
        if (m == 1 || !one_block) return HRT_FUSE_LAUNCH0;
        return HRT_FUSE_LAUNCH0 | HRT_FUSE_BOUNCES;
    }
    return HRT_FUSE_LAUNCH0 | (p->num_tri <= HRT_FUSE_MAX_TRI ? HRT_FUSE_BOUNCES : 0u);
}

static int trace_impl(const hrt_problem *p, const hrt_shard *s, const float *d_dirs,
                      const uint32_t *d_order, void *d_ws, uint64_t ws_bytes, void *stream,
                      hrt_timer *timer, uint32_t flags)
{
    if (!p || !d_dirs || !d_ws) return hrt_fail(HRT_E_INVALID, "hrt_trace: NULL argument");
    hrt_layout L;
    int rc = hrt_layout_query(p, s, &L);
    if (rc) return rc;
    if (ws_bytes < L.total_bytes)
        return hrt_fail(HRT_E_INVALID, "workspace too small: %llu < %llu bytes",
                        (unsigned long long)ws_bytes, (unsigned long long)L.total_bytes);
    if (((uintptr_t)d_ws & 255u) || ((uintptr_t)d_dirs & 3u))
        return hrt_fail(HRT_E_INVALID, "workspace must be 256-byte aligned");

    hrt_kparams K;
    memset(&K, 0, sizeof K);
    K.tri = p->d_tri; K.mesh = p->d_mesh; K.mat = p->d_mat;
    K.num_tri = p->num_tri; K.num_mesh = p->num_mesh;
    K.acc = p->kaccel;
    K.rxt = p->krxt;
    K.patch = p->kpatch;
    K.rx_pos = p->d_rx_pos; K.tx_pos = p->d_tx_pos; K.rx_vel = p->d_rx_vel; K.tx_vel = p->d_tx_vel;
    K.num_rx = p->num_rx; K.num_tx = p->num_tx;
    K.fsl_mult = p->fsl_mult; K.dop_mult = p->dop_mult;
    K.dirs = d_dirs;
    K.order = d_order;
    K.dirs_in_launch_order = (flags & HRT_DIRS_IN_LAUNCH_ORDER) ? 1u : 0u;
    K.num_local = (uint32_t)hrt_shard_num_local(s);
    K.num_bounces = s->num_bounces;
    K.n0 = p->num_tx * K.num_local;
    K.ws = (uint8_t *)d_ws;
    K.cap = L.cap;
    K.off_counts = L.off_counts; K.off_los = L.off_los; K.off_hits = L.off_hits;
    K.hit_block_bytes = L.hit_block_bytes; K.off_recs = L.off_recs;
    K.rec_block_bytes = L.rec_block_bytes; K.off_masks = L.off_masks;
    K.off_chunk_cnt = L.off_chunk_cnt; K.off_super_cnt = L.off_super_cnt;
    K.num_super = (uint32_t)L.num_super;
    K.off_res = L.off_res;
    K.off_lb = L.off_lb;
    K.cnt_stride = (uint32_t)HRT_CNT_STRIDE(s->num_bounces);
    K.host_flag = p->d_fuse_flag;
    K.lb_stride = (uint32_t)L.lb_stride;
    K.off_wide_q = L.off_wide_q; K.off_wide_key = L.off_wide_key; K.wide_cap = (uint32_t)L.wide_cap;
    K.wide_inv = p->d_inv;
    K.wide_cos = p->tune.wide_cos != 0.0 ? (float)p->tune.wide_cos : (p->num_tri >= HRT_WIDE_COS_BIG_TRI ? HRT_WIDE_COS_BIG : HRT_WIDE_COS);
    K.tune = p->tune.k;
    K.lb_chunks = (uint32_t)round_up(L.cap / HRT_BLOCK + 1, 64);
    K.fuse = fuse_mode(p);
    if (p->sort_rays) {
        uint32_t txb = 0;
        while ((1u << txb) < p->num_tx) ++txb;
        K.sort.enabled = 1u;
        {   /* 15 bits of cell, dealt to the axes so that the cells come out about cubic (a bit at a
             * time to the axis whose cells are longest); HRT_SORT_DIR_RES: cells per cube-face edge
             * of the direction bins (default 2: 24 bins); sort_fine: how many of the 15 cell bits
             * sort BEHIND the direction (default 6: key = 512 coarse cells | 24 directions | 64 fine
             * cells -- shadow rays, the majority, only care about the origin, the bounce itself
             * about both; measured on the room of 6 012 triangles: 13.9 / 9.7 / 9.3 / 11.4 ms for
             * 0 / 6 / 9 / 15, on the city of 25 002: 24.4 / 23.0 / 24.7 / 22.4) */
            uint32_t res = (uint32_t)(p->tune.sort_dir_res > 0 ? p->tune.sort_dir_res : 2);
            uint32_t nfine = (uint32_t)(p->tune.sort_fine >= 0 ? p->tune.sort_fine : 6);
            if (res < 1u) res = 1u;
            if (res > 8u) res = 8u;
            if (nfine > 15u) nfine = 15u;
            uint32_t bits = 0;
            while ((1u << bits) < 6u * res * res) ++bits;
            float ext[3];
            for (int k = 0; k < 3; ++k) {
                ext[k] = p->scene_hi[k] - p->scene_lo[k];
                if (!(ext[k] > 0.f)) ext[k] = 1e-30f;
                K.sort.bits[k] = 0;
            }
            for (int n = 0; n < 15; ++n) {
                int best = 0;
                for (int k = 1; k < 3; ++k)
                    if (ext[k] / (float)(1u << K.sort.bits[k]) > ext[best] / (float)(1u << K.sort.bits[best])) best = k;
                K.sort.bits[best]++;
            }
            K.sort.nfine = nfine;
            for (int k = 0; k < 3; ++k) {
                K.sort.lo[k] = p->scene_lo[k];
                K.sort.inv_cell[k] = (float)(1u << K.sort.bits[k]) / ext[k];
            }
            K.sort.dir_res = res;
            K.sort.dir_bits = bits;
            K.sort.tx_shift = 15u + bits;
            K.sort.key_bits = 15u + bits + txb;
            if (K.sort.key_bits > 32u) return hrt_fail(HRT_E_INVALID, "re-sort keys do not fit 32 bits");
        }
        K.sort.off_scratch = L.off_sort_scratch;
        K.sort.off_keys = L.off_sort_keys;
        K.sort.off_tmp = L.off_sort_tmp;
        K.sort.tmp_bytes = L.sort_tmp_bytes;
    }

    HRT_HIP(hrt_hip_set_device(p->device), "hipSetDevice");
    const uint32_t nb = s->num_bounces;
    /* the second stream (created with the first trace that can use it; HRT_OVERLAP=0: one stream) */
    hrt_problem *pp = (hrt_problem *)p;
    int aux = 0, forked = 0;
    if (p->kpatch.mask) {
        const char *ov = getenv("HRT_OVERLAP");
        if (!(ov && *ov == '0')) {
            if (!pp->aux_stream) {
                int e2;
                if ((e2 = hrt_hip_stream_create(&pp->aux_stream)) || (e2 = hrt_hip_event_create_sync(&pp->aux_ev[0])) ||
                    (e2 = hrt_hip_event_create_sync(&pp->aux_ev[1])))
                    return hrt_fail_hip(e2, "hipStreamCreate(records stream)");
            }
            aux = 1;
        }
    }
    /* events (only with a timer): [0,1] around LoS; per launch b: [2] start, [4] end of the records kernel (patch
     * tables; else = start), [3] end of the trace / fused kernel, [5] end of the shade kernel */
    void **ev = timer ? timer->ev : NULL;
    if (timer && timer->num_bounces != nb)
        return hrt_fail(HRT_E_INVALID, "timer was created for %u bounces, trace has %u", timer->num_bounces, nb);
    int hip = 0;
#define STEP(call) do { if (!hip) hip = (call); } while (0)
    /* counts, device error flag, and the super-chunk counts right behind them */
    STEP(hrt_hip_memset_async((uint8_t *)d_ws + L.off_counts, 0, L.off_los - L.off_counts, stream));
    /* the LoS pass: its own (tiny) kernel, or -- when launch 0 is one fused kernel -- a few extra
     * workgroups of that kernel (a launch less: it matters on launch sets of a few 10^4 rays) */
    /* (big tables with few pairs: the sliced LoS kernel, hrt_los_big_kernel, not one wave per pair as the tail of
     * launch 0 -- 0.8 ms per 100 k triangles) */
    K.los_blocks = ((K.fuse & HRT_FUSE_LAUNCH0) &&
                    !(p->num_tri >= p->tune.k.los_big_min_tri && (uint64_t)p->num_rx * p->num_tx <= 32u)) ? 1u : 0u;
    if (ev) STEP(hrt_hip_event_record(ev[0], stream));
    if (!K.los_blocks) STEP(hrt_hip_launch_los(&K, stream));
    if (ev) STEP(hrt_hip_event_record(ev[1], stream));
    for (uint32_t b = 0; b <= nb; ++b) {
        if (ev) STEP(hrt_hip_event_record(ev[2 + 4 * b], stream));
        /* Patch tables: the records of bounce b-1 (shadow traces + records, hrt_records_kernel) need only the
         * live list of this launch, and nothing of this launch needs them: they run on the problem's second
         * stream, beside the bounce's own kernels (a VALU-bound kernel beside latency-bound ones), forked and
         * joined by events.  With a timer everything stays on one stream, so that the per-kernel times mean
         * what they say. */
        /* Tables on which whole bounces are fused: the TAIL -- launches 2 .. nb -- as ONE persistent kernel
         * (hrt_chain_kernel: a grid barrier instead of a launch per bounce, and it ends with the first empty
         * list).  Measured on C4 (profiles/HISTORY.md r4): an empty launch costs 3 us + the dispatch of a grid
         * sized for the worst case (7 us for C4's 31 250 workgroups), a launch with a few thousand entries 15-20 us
         * (a workgroup's latency chain); a bounce inside the chain costs about the same as such a launch, its
         * roll call 6 us -- so the chain pays from three or four launches on (C4, 6 bounces: 0.629 -> 0.607 ms;
         * 12 bounces: 0.697 -> 0.613), not before (3 bounces: 0.595 -> 0.602), and launch 1, a LONG list, stays
         * its own kernel (in the chain's static deal of chunks it took 13 % longer).  With a timer every launch
         * keeps its own kernel.  tune: no_chain, chain_from (first launch of the chain; set: no minimum depth) */
        const uint32_t chain_b0 = p->tune.chain_from >= 1 ? (uint32_t)p->tune.chain_from : 2u;
        if (b == chain_b0 && (p->tune.chain_from >= 1 || nb >= 4u) && (K.fuse & HRT_FUSE_BOUNCES) && !ev && !p->sort_rays &&
            !p->tune.no_chain && !g_chain_off && !hip) {
            const int cr = hrt_hip_launch_chain(&K, b, stream);   /* -1: not on this problem */
            if (cr > 0) hip = cr;
            if (cr >= 0) break;
        }
        int own_records = 0;
        if (b >= 1 && !hip) {
            void *rs = stream;
            if (aux && !ev) {
                STEP(hrt_hip_event_record(pp->aux_ev[0], stream));
                STEP(hrt_hip_stream_wait_event(pp->aux_stream, pp->aux_ev[0]));
                rs = pp->aux_stream;
            }
            const int rr = hip ? 0 : hrt_hip_launch_records(&K, b, rs);   /* -1: not this problem */
            if (rr > 0) hip = rr;
            if (rr == 0) own_records = 1;
            if (rr == 0 && rs != stream) forked = 1;
        }
        if (ev) STEP(hrt_hip_event_record(ev[4 + 4 * b], stream));   /* end of the records kernel (or: nothing ran) */
        if (own_records && b == nb) {   /* the last launch: records only */
            if (ev) { STEP(hrt_hip_event_record(ev[3 + 4 * b], stream)); STEP(hrt_hip_event_record(ev[5 + 4 * b], stream)); }
            continue;
        }
        /* one kernel for the whole launch (trace + shading + stable compaction) where that is what the
         * launch wants: launch 0 always, every launch on tables of one culling round; then the "trace" time is
         * the fused kernel's and the "shade" time zero.  (Patch tables, records in their own kernel: the bounce
         * alone as one fused kernel was measured and lost -- C3 1.21 -> 1.40 ms: at 118 registers the packet walk
         * runs at 4 waves per SIMD, and its spinning waves keep the records kernel from overlapping.) */
        const int fused = b == 0 ? (K.fuse & HRT_FUSE_LAUNCH0) != 0 : (K.fuse & HRT_FUSE_BOUNCES) != 0;
        int fused_rc = -1;
        if (fused && !hip) fused_rc = hrt_hip_launch_fused(&K, b, stream);   /* -1: this table / variant is not fusable */
        if (fused_rc < 0 && b == 0 && K.los_blocks) {   /* launch 0 is not fused after all: the LoS pass as its own kernel */
            K.los_blocks = 0u;
            STEP(hrt_hip_launch_los(&K, stream));
            if (ev) {   /* (the timer: LoS ends here, launch 0 begins here) */
                STEP(hrt_hip_event_record(ev[1], stream));
                STEP(hrt_hip_event_record(ev[2], stream));
                STEP(hrt_hip_event_record(ev[4], stream));
            }
        }
        if (fused_rc >= 0) {
            STEP(fused_rc);
            if (ev) STEP(hrt_hip_event_record(ev[3 + 4 * b], stream));
            if (p->sort_rays && b < nb) STEP(hrt_hip_sort_hits(&K, b, stream));
            if (ev) STEP(hrt_hip_event_record(ev[5 + 4 * b], stream));
            continue;
        }
        STEP(hrt_hip_launch_trace(&K, b, stream));
        if (ev) STEP(hrt_hip_event_record(ev[3 + 4 * b], stream));
        STEP(hrt_hip_launch_shade(&K, b, stream));
        if (p->sort_rays && b < nb) STEP(hrt_hip_sort_hits(&K, b, stream));   /* part of the "shade" time */
        if (ev) STEP(hrt_hip_event_record(ev[5 + 4 * b], stream));
    }
    if (forked) {   /* join: the caller's stream continues behind the last records kernel */
        STEP(hrt_hip_event_record(pp->aux_ev[1], pp->aux_stream));
        STEP(hrt_hip_stream_wait_event(stream, pp->aux_ev[1]));
    }
#undef STEP
    if (timer) timer->recorded = !hip;
    if (hip) return hrt_fail_hip(hip, "hrt_trace");
    return HRT_OK;
}

void hrt_work_from_counts(const hrt_problem *p, const hrt_shard *s, const uint32_t *counts,
                          hrt_stats *st)
{
    memset(st, 0, sizeof *st);
    const uint64_t nb = s->num_bounces, T = p->num_tri, nrx = p->num_rx;
    /* (hrt_stats.live has 34 slots: the first 33 launches and their hits; the totals cover every launch) */
    const uint64_t n0 = (uint64_t)p->num_tx * hrt_shard_num_local(s);
    st->live[0] = n0;
    for (uint64_t b = 1; b <= nb && b < 34; ++b) st->live[b] = counts[b];
    uint64_t tests = (uint64_t)p->num_rx * p->num_tx * T;
    for (uint64_t b = 0; b < nb; ++b) {
        const uint64_t in = b == 0 ? n0 : counts[b], hits = counts[b + 1];
        tests += T * (in + nrx * hits);
        st->records += nrx * hits;
    }
    st->tests = tests;
    st->device = p->device;
}

/* ------------------------------------------------------------------ device helpers */

int hrt_device_count(int *out)
{
    int n = 0;
    int rc = hrt_hip_device_count(&n);
    if (rc) { *out = 0; return hrt_fail_hip(rc, "hipGetDeviceCount"); }
    *out = n;
    return HRT_OK;
}
int hrt_device_malloc(int device, void **out, uint64_t bytes)
{
    HRT_HIP(hrt_hip_set_device(device), "hipSetDevice");
    HRT_HIP(hrt_hip_malloc(out, bytes), "hipMalloc");
    return HRT_OK;
}
int hrt_device_free(int device, void *ptr)
{
    HRT_HIP(hrt_hip_set_device(device), "hipSetDevice");
    HRT_HIP(hrt_hip_free(ptr), "hipFree");
    return HRT_OK;
}
int hrt_device_upload(int device, void *dst, const void *src, uint64_t bytes)
{
    HRT_HIP(hrt_hip_set_device(device), "hipSetDevice");
    HRT_HIP(hrt_hip_h2d(dst, src, bytes), "hipMemcpy H2D");
    return HRT_OK;
}
int hrt_device_download(int device, void *dst, const void *src, uint64_t bytes)
{
    HRT_HIP(hrt_hip_set_device(device), "hipSetDevice");
    HRT_HIP(hrt_hip_d2h(dst, src, bytes), "hipMemcpy D2H");
    return HRT_OK;
}
int hrt_device_sync(int device, void *stream)
{
    HRT_HIP(hrt_hip_set_device(device), "hipSetDevice");
    HRT_HIP(hrt_hip_stream_sync(stream), "hipStreamSynchronize");
    return HRT_OK;
}
int hrt_device_mem_info(int device, uint64_t *free_bytes, uint64_t *total_bytes)
{
    HRT_HIP(hrt_hip_set_device(device), "hipSetDevice");
    HRT_HIP(hrt_hip_mem_info(free_bytes, total_bytes), "hipMemGetInfo");
    return HRT_OK;
}

/* ------------------------------------------------------------------ self test */

int hrt_selftest_math(int device, int fn, const float *in, float *out, uint64_t n)
{
    if (!in || !out || fn < 0 || fn > 7) return hrt_fail(HRT_E_INVALID, "hrt_selftest_math: bad argument");
    void *d_in = NULL, *d_out = NULL;
    int rc = hrt_device_malloc(device, &d_in, n * 4);
    if (rc) return rc;
    if ((rc = hrt_device_malloc(device, &d_out, n * 4))) { hrt_device_free(device, d_in); return rc; }
    if (!(rc = hrt_device_upload(device, d_in, in, n * 4))) {
        int e = hrt_hip_selftest_math(fn, (const float *)d_in, (float *)d_out, n, NULL);
        if (e) rc = hrt_fail_hip(e, "hrt_selftest_math");
        else if (!(rc = hrt_device_sync(device, NULL))) rc = hrt_device_download(device, out, d_out, n * 4);
    }
    hrt_device_free(device, d_in);
    hrt_device_free(device, d_out);
    return rc;
}

/* Diagnostic counters of the packet-culling loop (all zero unless built with `make STATS=1`):
 * out[kind][16], kind 0 = primary traces of launch 0, 1 = primary traces of later launches,
 * 2 = shadow traces; columns: see include/hrt_device.h. */
int hrt_debug_kernel_stats(int device, uint64_t *out48, int reset)
{
    HRT_HIP(hrt_hip_set_device(device), "hipSetDevice");
    HRT_HIP(hrt_hip_read_stats((unsigned long long *)out48, reset), "hrt_hip_read_stats");
    return HRT_OK;
}
