/* accel.c -- the acceleration structure over the triangle table (the reference
 * marks the spot "TODO BVH" at src/compute_paths.c:246 and tests every triangle).
 *
 * What is built here, on the host, once per problem (DESIGN.md section 9 has the proofs):
 *
 *   1. ORDER.  The triangle table is put into a balanced k-d order of the triangle centroids, so
 *      that 64 consecutive rows -- one culling round of the trace kernel, a LEAF -- are neighbours
 *      in space, and so are the 64 leaves of every node above.  The reference resolves equal-distance ties by its loop order (lowest (mesh, face)
 *      wins, src/compute_paths.c:253-275); the kernels keep that by tracking the lexicographic
 *      minimum of (distance, original index), for which `orig` (new -> original) is uploaded.
 *   2. LEAVES.  Per leaf a bounding sphere (c, R) of its triangles and Lambda, its longest edge;
 *      per triangle the guard pair (qs, l): l its longest edge and qs = SF * 1e-5 * l^2 / area,
 *      the shape number that scales the reference's rounding noise into a distance.
 *   3. INNER LEVELS (tables of more than HRT_ACCEL_BIG triangles).  64-ary: node j of level k is
 *      the bounding sphere of nodes 64j .. 64j+63 of level k-1 (level 0 = leaves), plus Lambda.
 *   4. PLANE TREE (same tables).  The triangles once more, sorted by the direction of their
 *      normal (sign dropped): 64-entry leaves of triangle ids with a guard record each, and 64-ary
 *      levels of normal cones above them.  A packet asks it for every triangle whose plane is
 *      nearly parallel to the packet's rays -- the only ones the spheres cannot vouch for.
 *
 * Everything is computed in double and rounded to the conservative side.
 */
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "hrt_internal.h"

#define SF_QS (HRT_GUARD_SF * 1e-5)

static double dmax(double a, double b) { return a > b ? a : b; }

static uint32_t spread2(uint32_t v)   /* 15 bits -> every second bit */
{
    v &= 0x7fffu;
    v = (v | (v << 8)) & 0x00ff00ffu;
    v = (v | (v << 4)) & 0x0f0f0f0fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}

typedef struct { uint32_t key, idx; } keyed;
static int keyed_cmp(const void *a, const void *b)
{
    const keyed *x = (const keyed *)a, *y = (const keyed *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx);   /* stable: deterministic tables */
}

static float up(double v)   /* smallest float >= v (v >= 0), NaN/inf kept */
{
    float f = (float)v;
    if (isfinite(v) && (double)f < v) f = nextafterf(f, INFINITY);
    return f;
}

static int finite3(const float *p) { return isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]); }

void hrt_accel_free(hrt_accel *a)
{
    if (!a) return;
    free(a->orig); free(a->newidx); free(a->leaf); free(a->tg);
    for (int k = 0; k < HRT_ACCEL_MAX_LEVELS; ++k) { free(a->node[k]); free(a->pl_node[k]); }
    free(a->pl_index); free(a->pl_rec); free(a->fine);
    memset(a, 0, sizeof *a);
}

/* Spatial order of the rows: a balanced k-d ordering of the triangle centroids -- the index range
 * is cut at the median of its longest axis, recursively, the cut rounded to a multiple of the
 * largest power of 64 below the range (so that leaves of 64 rows, and the 64-leaf groups of every
 * level above, are each one compact cell) -- Morton order proved too loose: 64 consecutive codes
 * often straddle a jump of the curve.  Triangles with a non-finite centroid go last.
 * rows: [T][HRT_TRI_FLOATS] in the reference's (mesh, face) order.  Fills a->orig / a->newidx.
 * `reorder` == 0 keeps the identity. */
typedef struct { float c[3]; uint32_t idx; } cent;
static int g_axis;
static int cent_cmp(const void *a, const void *b)
{
    const cent *x = (const cent *)a, *y = (const cent *)b;
    if (x->c[g_axis] != y->c[g_axis]) return x->c[g_axis] < y->c[g_axis] ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx);
}
static void kd_order(cent *v, uint32_t n)
{
    while (n > 64u) {
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t i = 0; i < n; ++i)
            for (int k = 0; k < 3; ++k) {
                if (v[i].c[k] < lo[k]) lo[k] = v[i].c[k];
                if (v[i].c[k] > hi[k]) hi[k] = v[i].c[k];
            }
        int ax = 0;
        if (hi[1] - lo[1] > hi[ax] - lo[ax]) ax = 1;
        if (hi[2] - lo[2] > hi[ax] - lo[ax]) ax = 2;
        g_axis = ax;                       /* (called under the problem-creation lock: one builder at a time) */
        qsort(v, n, sizeof(cent), cent_cmp);
        uint32_t unit = 64u;
        while ((uint64_t)unit * 64u < n) unit *= 64u;
        uint32_t r = unit * (uint32_t)((n + unit) / (2u * unit));   /* ~ n / 2, a multiple of unit */
        if (r == 0) r = unit;
        if (r >= n) r = n - (n % unit ? n % unit : unit);
        if (r == 0 || r >= n) return;
        kd_order(v, r);                    /* recurse on the left, loop on the right */
        v += r;
        n -= r;
    }
}

static pthread_mutex_t g_kd_lock = PTHREAD_MUTEX_INITIALIZER;

int hrt_accel_order(hrt_accel *a, const float *rows, uint32_t T, int reorder)
{
    memset(a, 0, sizeof *a);
    a->num_tri = T;
    a->orig = (uint32_t *)malloc((size_t)(T ? T : 1) * 4);
    a->newidx = (uint32_t *)malloc((size_t)(T ? T : 1) * 4);
    cent *cs = (cent *)malloc((size_t)(T ? T : 1) * sizeof(cent));
    if (!a->orig || !a->newidx || !cs) { free(cs); return hrt_fail(HRT_E_NOMEM, "out of host memory"); }
    uint32_t nf = 0, nb = T;               /* finite centroids from the front, the others from the back */
    for (uint32_t j = 0; j < T; ++j) {
        const float *r = rows + (size_t)j * HRT_TRI_FLOATS;
        cent c;
        int ok = 1;
        for (int k = 0; k < 3; ++k) {
            const double g = (double)r[k] + ((double)r[3 + k] + (double)r[6 + k]) / 3.0;
            c.c[k] = (float)g;
            if (!isfinite(c.c[k])) ok = 0;
        }
        c.idx = j;
        if (ok || !reorder) cs[nf++] = c;
        else cs[--nb] = c;
    }
    if (reorder) {
        pthread_mutex_lock(&g_kd_lock);
        kd_order(cs, nf);
        pthread_mutex_unlock(&g_kd_lock);
        /* the non-finite ones: in index order behind the rest */
        for (uint32_t i = nb, k = T; i < k && i + 1 < k; ++i, --k) { const cent t = cs[i]; cs[i] = cs[k - 1]; cs[k - 1] = t; }
    }
    for (uint32_t j = 0; j < T; ++j) {
        a->orig[j] = cs[j].idx;
        a->newidx[cs[j].idx] = j;
    }
    free(cs);
    return HRT_OK;
}

/* bounding sphere of a set of spheres / points given as (c, r) pairs via callback-free arrays */
static void sphere_of(const double (*c)[3], const double *r, uint32_t n, float out[4])
{
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    int bad = (n == 0);
    for (uint32_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) {
            if (!isfinite(c[i][k]) || !isfinite(r[i])) bad = 1;
            if (c[i][k] - r[i] < lo[k]) lo[k] = c[i][k] - r[i];
            if (c[i][k] + r[i] > hi[k]) hi[k] = c[i][k] + r[i];
        }
    if (bad) { out[0] = out[1] = out[2] = 0.f; out[3] = NAN; return; }   /* NaN radius: never "far" */
    double m[3] = {0.5 * (lo[0] + hi[0]), 0.5 * (lo[1] + hi[1]), 0.5 * (lo[2] + hi[2])};
    /* the centre is what the float stores; the radius is measured from THAT point */
    float mf[3] = {(float)m[0], (float)m[1], (float)m[2]};
    double R = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const double dx = c[i][0] - mf[0], dy = c[i][1] - mf[1], dz = c[i][2] - mf[2];
        R = dmax(R, sqrt(dx * dx + dy * dy + dz * dz) + r[i]);
    }
    out[0] = mf[0]; out[1] = mf[1]; out[2] = mf[2];
    out[3] = up(R * (1.0 + 1e-6) + 1e-30);
}

/* per-triangle numbers from a table row: longest edge (rounded up), shape number qs */
static void tri_guard(const float *r, double *ell, double *qs)
{
    const double e1[3] = {r[3], r[4], r[5]}, e2[3] = {r[6], r[7], r[8]};
    const double e3[3] = {e2[0] - e1[0], e2[1] - e1[1], e2[2] - e1[2]};
    const double l1 = sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
    const double l2 = sqrt(e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]);
    const double l3 = sqrt(e3[0] * e3[0] + e3[1] * e3[1] + e3[2] * e3[2]);
    const double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    const double area = 0.5 * sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    const double l = dmax(l1, dmax(l2, l3));
    *ell = l * (1.0 + 1e-6);
    double q = (area > 0 && isfinite(area) && isfinite(l)) ? (l * l / area) * (1.0 + 1e-5) : INFINITY;
    /* the float normal of a thin triangle is itself only good to ~1e-7 q: never trust it beyond */
    if (!(q < 1e6) || !finite3(r + 9)) q = INFINITY;
    *qs = SF_QS * q;
}

static int canon_key(const float *n, uint32_t *key, double c[3])
{
    if (!finite3(n)) { *key = 0xffffffffu; c[0] = c[1] = c[2] = NAN; return 0; }
    double v[3] = {n[0], n[1], n[2]};
    int ax = 0;
    if (fabs(v[1]) > fabs(v[ax])) ax = 1;
    if (fabs(v[2]) > fabs(v[ax])) ax = 2;
    if (v[ax] == 0) { *key = 0xffffffffu; c[0] = c[1] = c[2] = NAN; return 0; }
    const double s = v[ax] < 0 ? -1.0 : 1.0;
    for (int k = 0; k < 3; ++k) c[k] = v[k] * s;
    const double u = c[(ax + 1) % 3] / c[ax], w = c[(ax + 2) % 3] / c[ax];   /* in [-1, 1] */
    const uint32_t qu = (uint32_t)((u * 0.5 + 0.5) * 32767.0), qw = (uint32_t)((w * 0.5 + 0.5) * 32767.0);
    *key = ((uint32_t)ax << 30) | spread2(qu) | (spread2(qw) << 1);
    return 1;
}

/* rows: the table ALREADY in its final (Morton) order */
static int cmp_float(const void *a, const void *b)
{
    const float x = *(const float *)a, y = *(const float *)b;
    return (x > y) - (x < y);
}

int hrt_accel_build(hrt_accel *a, const float *rows, const hrt_tune *tune)
{
    const uint32_t T = a->num_tri;
    const uint32_t nl = (T + 63u) / 64u;
    a->num_leaf = nl;
    a->leaf = (float *)calloc((size_t)(nl ? nl : 1) * HRT_NODE_FLOATS, sizeof(float));
    a->tg = (float *)calloc((size_t)(T ? T : 1) * 2, sizeof(float));
    if (!a->leaf || !a->tg) return hrt_fail(HRT_E_NOMEM, "out of host memory");
    double (*cs)[3] = (double (*)[3])malloc(sizeof(double[3]) * 192);
    double *rs = (double *)calloc(192, sizeof(double));
    if (!cs || !rs) { free(cs); free(rs); return hrt_fail(HRT_E_NOMEM, "out of host memory"); }
    for (uint32_t j = 0; j < T; ++j) {
        double l, qs;
        tri_guard(rows + (size_t)j * HRT_TRI_FLOATS, &l, &qs);
        a->tg[2 * (size_t)j] = up(qs);
        a->tg[2 * (size_t)j + 1] = up(l);
    }
    for (uint32_t b = 0; b < nl; ++b) {
        const uint32_t j0 = b * 64u, j1 = (j0 + 64u < T) ? j0 + 64u : T;
        uint32_t n = 0;
        double lam = 0;
        for (uint32_t j = j0; j < j1; ++j) {
            const float *r = rows + (size_t)j * HRT_TRI_FLOATS;
            for (int v = 0; v < 3; ++v, ++n) {
                for (int k = 0; k < 3; ++k)
                    cs[n][k] = (double)r[k] + (v == 1 ? (double)r[3 + k] : (v == 2 ? (double)r[6 + k] : 0.0));
                rs[n] = 0;
            }
            lam = dmax(lam, (double)a->tg[2 * (size_t)j + 1]);
            if (!isfinite((double)a->tg[2 * (size_t)j + 1])) lam = INFINITY;
        }
        float *L = a->leaf + (size_t)b * HRT_NODE_FLOATS;
        sphere_of((const double (*)[3])cs, rs, n, L);
        L[4] = up(lam);
    }
    /* fine leaves: spheres of HRT_FINE_ROWS consecutive rows (the order is a k-d order: any run of rows is
     * a compact cell), for the flat scan of closest_hit_fine -- the default walk of tables of more than
     * HRT_FINE_MIN_TRI triangles that are not "big" (HRT_ACCEL_FINE_MIN=n replaces the threshold;
     * HRT_ACCEL_FINE=0 brings the 64-row leaf walk back). */
    int want_fine = T > HRT_FINE_MIN_TRI;
    if (tune->accel_fine_min != UINT64_MAX) want_fine = (unsigned long long)T > tune->accel_fine_min;
    if (tune->accel_fine == 0) want_fine = 0;
    if (want_fine) {
        const uint32_t nfn = (T + HRT_FINE_ROWS - 1u) / HRT_FINE_ROWS;
        a->fine = (float *)calloc((size_t)nfn * HRT_NODE_FLOATS, sizeof(float));
        if (!a->fine) { free(cs); free(rs); return hrt_fail(HRT_E_NOMEM, "out of host memory"); }
        a->num_fine = nfn;
        for (uint32_t b = 0; b < nfn; ++b) {
            const uint32_t j0 = b * HRT_FINE_ROWS, j1 = (j0 + HRT_FINE_ROWS < T) ? j0 + HRT_FINE_ROWS : T;
            uint32_t n = 0;
            double lam = 0;
            for (uint32_t j = j0; j < j1; ++j) {
                const float *r = rows + (size_t)j * HRT_TRI_FLOATS;
                for (int v = 0; v < 3; ++v, ++n) {
                    for (int k = 0; k < 3; ++k)
                        cs[n][k] = (double)r[k] + (v == 1 ? (double)r[3 + k] : (v == 2 ? (double)r[6 + k] : 0.0));
                    rs[n] = 0;
                }
                lam = dmax(lam, (double)a->tg[2 * (size_t)j + 1]);
                if (!isfinite((double)a->tg[2 * (size_t)j + 1])) lam = INFINITY;
            }
            float *L = a->fine + (size_t)b * HRT_NODE_FLOATS;
            sphere_of((const double (*)[3])cs, rs, n, L);
            L[4] = up(lam);
        }
    }
    free(cs); free(rs);
    {
        /* Inner levels + plane tree pay when a leaf is small against the scene (a packet then passes
         * most leaves at a distance): tables of more than HRT_ACCEL_BIG triangles whose median leaf
         * radius is below HRT_ACCEL_SPARSE of the scene's.  In a dense scene (a room full of clutter,
         * every leaf near every ray) the per-triangle culling of the flat walk is the better tool.
         * HRT_TUNE: accel_big=n replaces the triangle threshold (0: always) and drops the
         * sparseness condition; accel_sparse=x replaces the ratio. */
        const int forced = tune->accel_big != HRT_ACCEL_BIG;
        a->big = (unsigned long long)T > tune->accel_big;
        if (a->big && !forced) {
            const double ratio = tune->accel_sparse;
            float *rad = (float *)malloc((size_t)nl * sizeof(float));
            double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
            uint32_t n = 0;
            if (!rad) return hrt_fail(HRT_E_NOMEM, "out of host memory");
            for (uint32_t b = 0; b < nl; ++b) {
                const float *L = a->leaf + (size_t)b * HRT_NODE_FLOATS;
                if (!isfinite(L[3])) continue;
                rad[n++] = L[3];
                for (int k = 0; k < 3; ++k) {
                    if (L[k] < lo[k]) lo[k] = L[k];
                    if (L[k] > hi[k]) hi[k] = L[k];
                }
            }
            if (n) {
                /* median (nl = T / 64 can be 2.6e5 at 16 M triangles: a sort, not a selection in O(nl^2)) */
                qsort(rad, n, sizeof(float), cmp_float);
                const double ext = 0.5 * sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) +
                                              (hi[2] - lo[2]) * (hi[2] - lo[2]));
                a->big = (double)rad[n / 2] < ratio * ext;
            } else a->big = 0;
            free(rad);
        }
    }
    if (!a->big && !a->fine) return HRT_OK;

    /* ---- inner levels over the leaves (big tables only) ---- */
    if (a->big) {
        const float *below = a->leaf;
        uint32_t nb = nl;
        double (*c64)[3] = (double (*)[3])malloc(sizeof(double[3]) * 64);
        double r64[64];
        if (!c64) return hrt_fail(HRT_E_NOMEM, "out of host memory");
        uint32_t k = 0;
        while (nb > 64u && k < HRT_ACCEL_MAX_LEVELS) {
            const uint32_t nn = (nb + 63u) / 64u;
            float *lev = (float *)calloc((size_t)nn * HRT_NODE_FLOATS, sizeof(float));
            if (!lev) { free(c64); return hrt_fail(HRT_E_NOMEM, "out of host memory"); }
            for (uint32_t j = 0; j < nn; ++j) {
                const uint32_t c0 = j * 64u, c1 = (c0 + 64u < nb) ? c0 + 64u : nb;
                double lam = 0;
                for (uint32_t c = c0; c < c1; ++c) {
                    const float *q = below + (size_t)c * HRT_NODE_FLOATS;
                    c64[c - c0][0] = q[0]; c64[c - c0][1] = q[1]; c64[c - c0][2] = q[2];
                    r64[c - c0] = q[3];
                    lam = dmax(lam, (double)q[4]);
                    if (!isfinite((double)q[4])) lam = INFINITY;
                }
                float *N = lev + (size_t)j * HRT_NODE_FLOATS;
                sphere_of((const double (*)[3])c64, r64, c1 - c0, N);
                N[4] = up(lam);
            }
            a->node[k] = lev;
            a->node_count[k] = nn;
            below = lev;
            nb = nn;
            ++k;
        }
        free(c64);
        if (nb > 64u) {   /* more than 64^4 triangles: no sphere levels */
            for (uint32_t q = 0; q < HRT_ACCEL_MAX_LEVELS; ++q) { free(a->node[q]); a->node[q] = NULL; a->node_count[q] = 0; }
            k = 0;
            a->big = 0;
            if (!a->fine) return HRT_OK;
        }
        a->num_levels = k;
    }

    /* ---- plane tree: triangles sorted by normal direction ---- */
    {
        keyed *ks = (keyed *)malloc((size_t)T * sizeof(keyed));
        double (*cn)[3] = (double (*)[3])malloc(sizeof(double[3]) * (size_t)T);
        if (!ks || !cn) { free(ks); free(cn); return hrt_fail(HRT_E_NOMEM, "out of host memory"); }
        for (uint32_t j = 0; j < T; ++j) {
            ks[j].idx = j;
            canon_key(rows + (size_t)j * HRT_TRI_FLOATS + 9, &ks[j].key, cn[j]);
            if (!isfinite((double)a->tg[2 * (size_t)j])) ks[j].key = 0xffffffffu;   /* no usable guard: "always" */
        }
        qsort(ks, T, sizeof(keyed), keyed_cmp);
        const uint32_t pnl = (T + 63u) / 64u;
        a->pl_num_leaf = pnl;
        a->pl_index = (uint32_t *)malloc((size_t)pnl * 64 * 4);
        a->pl_rec = (float *)calloc((size_t)pnl * 64 * HRT_NODE_FLOATS, sizeof(float));
        if (!a->pl_index || !a->pl_rec) { free(ks); free(cn); return hrt_fail(HRT_E_NOMEM, "out of host memory"); }
        for (uint32_t i = 0; i < pnl * 64u; ++i) {
            float *rec = a->pl_rec + (size_t)i * HRT_NODE_FLOATS;
            if (i >= T) {   /* padding: a record that is never flagged (the lane is masked anyway) */
                a->pl_index[i] = HRT_NO_HIT;
                continue;
            }
            const uint32_t j = ks[i].idx;
            const float *r = rows + (size_t)j * HRT_TRI_FLOATS;
            a->pl_index[i] = j;
            rec[0] = r[0]; rec[1] = r[1]; rec[2] = r[2];
            rec[3] = a->tg[2 * (size_t)j + 1];               /* l: the ball (p1, l) holds the triangle */
            rec[4] = r[9]; rec[5] = r[10]; rec[6] = r[11];
            rec[7] = a->tg[2 * (size_t)j];                   /* qs */
        }
        /* cone levels: level 0 = leaves of 64 entries, level k = 64 nodes of level k-1.  Node =
         * axis nu, then sin and cos of (beta + g): beta the largest angle between nu and a member
         * normal (sign dropped), g = asin(Gamma), Gamma = (12 / mu) max qs the direction margin the
         * members need.  (2, 0) = "always visit". */
        uint32_t span = 64u, k = 0, count = pnl;
        for (;;) {
            float *lev = (float *)calloc((size_t)count * HRT_NODE_FLOATS, sizeof(float));
            if (!lev) { free(ks); free(cn); return hrt_fail(HRT_E_NOMEM, "out of host memory"); }
            for (uint32_t nidx = 0; nidx < count; ++nidx) {
                const uint64_t i0 = (uint64_t)nidx * span, i1 = (i0 + span < T) ? i0 + span : T;
                double s[3] = {0, 0, 0}, first[3] = {0, 0, 0}, gam = 0;
                int always = 0, have = 0;
                for (uint64_t i = i0; i < i1; ++i) {
                    const uint32_t j = ks[i].idx;
                    if (ks[i].key == 0xffffffffu) { always = 1; break; }
                    if (!have) { memcpy(first, cn[j], sizeof first); have = 1; }
                    const double sg = (cn[j][0] * first[0] + cn[j][1] * first[1] + cn[j][2] * first[2]) < 0 ? -1.0 : 1.0;
                    for (int c = 0; c < 3; ++c) s[c] += sg * cn[j][c];
                    gam = dmax(gam, (12.0 / HRT_GUARD_MU) * (double)a->tg[2 * (size_t)j]);
                }
                float *N = lev + (size_t)nidx * HRT_NODE_FLOATS;
                const double sl = sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
                if (always || !have || !(sl > 0) || !(gam < 1.0)) {
                    N[0] = 0.f; N[1] = 0.f; N[2] = 1.f; N[3] = 2.f; N[4] = 0.f;
                    continue;
                }
                const double nu[3] = {s[0] / sl, s[1] / sl, s[2] / sl};
                const float nuf[3] = {(float)nu[0], (float)nu[1], (float)nu[2]};
                double cmin = 1.0;   /* smallest |nu . n| over the members, against the STORED axis */
                for (uint64_t i = i0; i < i1; ++i) {
                    const uint32_t j = ks[i].idx;
                    const double nl2 = sqrt(cn[j][0] * cn[j][0] + cn[j][1] * cn[j][1] + cn[j][2] * cn[j][2]);
                    const double fl = sqrt((double)nuf[0] * nuf[0] + (double)nuf[1] * nuf[1] + (double)nuf[2] * nuf[2]);
                    const double c = fabs(cn[j][0] * nuf[0] + cn[j][1] * nuf[1] + cn[j][2] * nuf[2]) / (nl2 * fl);
                    if (c < cmin) cmin = c;
                }
                double ang = acos(cmin > 1 ? 1 : cmin) + asin(gam) + 1e-5;   /* beta + g, widened */
                if (!(ang < 1.5)) { N[0] = 0.f; N[1] = 0.f; N[2] = 1.f; N[3] = 2.f; N[4] = 0.f; continue; }
                N[0] = nuf[0]; N[1] = nuf[1]; N[2] = nuf[2];
                N[3] = up(sin(ang) * (1.0 + 1e-6));
                N[4] = (float)(cos(ang) * (1.0 + 1e-6));   /* used with a positive weight: round up */
                if ((double)N[4] < cos(ang)) N[4] = nextafterf(N[4], INFINITY);
            }
            a->pl_node[k] = lev;
            a->pl_count[k] = count;
            ++k;
            if (count <= 64u || k >= HRT_ACCEL_MAX_LEVELS) break;
            count = (count + 63u) / 64u;
            span *= 64u;
        }
        a->pl_levels = k;
        free(ks); free(cn);
        if (a->pl_count[k - 1] > 64u) {
            /* more than 64^4 triangles in a sparse scene: the trees do not fit their levels -- drop them
             * and walk the leaves (any table size) instead of failing the problem */
            for (uint32_t q = 0; q < HRT_ACCEL_MAX_LEVELS; ++q) {
                free(a->node[q]); a->node[q] = NULL; a->node_count[q] = 0;
                free(a->pl_node[q]); a->pl_node[q] = NULL; a->pl_count[q] = 0;
            }
            free(a->pl_index); a->pl_index = NULL;
            free(a->pl_rec); a->pl_rec = NULL;
            a->num_levels = 0; a->pl_levels = 0; a->pl_num_leaf = 0;
            a->big = 0;
            free(a->fine); a->fine = NULL; a->num_fine = 0;   /* (the fine leaves need the plane tree as their guard) */
        } else {
            a->planes = 1;
        }
    }
    return HRT_OK;
}
