/* Design study (CPU, not shipped): candidate-set sizes of table designs on real waves.
 * Restates the device's packet_culls / packet_bounds / rxt_cell (hrt_kernels.hip) in host float
 * arithmetic so that table designs can be priced before a kernel exists.
 *   gcc -O2 -fopenmp -shared -fPIC cand.c -o /tmp/hrt_study/libcand.so -lm */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ROWF 20
#ifndef RXT_N
#define RXT_N 48
#endif
#define EPS 1.1920928955078125e-07f

typedef struct { float oc[3], ro, bc[3], br, ax[3], cosa, sina; int usable; } Packet;

static inline float fdot(const float *a, const float *b) { return fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0])); }
static inline void fcross(const float *a, const float *b, float *o)
{
    o[0] = fmaf(a[1], b[2], -(a[2] * b[1])); o[1] = fmaf(a[2], b[0], -(a[0] * b[2])); o[2] = fmaf(a[0], b[1], -(a[1] * b[0]));
}
static inline void cone_bounds(const Packet *P, const float *G, float *hp, float *hm)
{
    const float g2 = fdot(G, G), c1 = fdot(P->ax, G);
    const float s1 = sqrtf(fmaxf(0.f, fmaf(-c1, c1, g2)));
    const float sl = fabsf(c1) + s1, t1 = c1 * P->cosa, t2 = s1 * P->sina;
    *hp = fmaf(1e-4f, sl, t2 + t1);
    *hm = fmaf(1e-4f, sl, t2 - t1);
}
static int packet_culls(const Packet *P, const float *r)
{
    const float Cy = r[12], Cz = r[13], Cw = r[14], Lx = r[15], Ly = r[16], Lz = r[17], Lw = r[18];
    const float kE = 16.f * EPS;
    const float *v1 = r, *e1 = r + 3, *e2 = r + 6, *nh = r + 9;
    float sb[3] = {P->bc[0] - v1[0], P->bc[1] - v1[1], P->bc[2] - v1[2]};
    const float S = (fabsf(sb[0]) + fabsf(sb[1])) + (fabsf(sb[2]) + P->br);
    float sc[3] = {P->oc[0] - v1[0], P->oc[1] - v1[1], P->oc[2] - v1[2]};
    const float dn = fdot(P->ax, nh);
    const int one_sided = Lw * fmaf(fabsf(dn), P->cosa, -P->sina) > fmaf(3.f, Cy, 1e-30f);
    const float k2S = (2.f * kE) * S;
    const float tol_u = fmaf(k2S, Ly, Cz), tol_v = fmaf(k2S, Lx, Cz), tol_w = fmaf(k2S, Lx + Ly, Cw);
    const float h = fdot(sb, nh);
    const float thr = fmaf(-1e-4f * Lw, fabsf(h) + P->br, -(k2S * Lx) * Ly);
    int rej_p = (P->br + h) * Lw < thr, rej_m = (P->br - h) * Lw < thr;
    float Gu[3], Gv[3], Gw[3], hp, hm;
    fcross(e2, sc, Gu);
    fcross(sc, e1, Gv);
    for (int k = 0; k < 3; ++k) Gw[k] = fmaf(nh[k], Lw, Gu[k] + Gv[k]);
    const float ro = P->ro * 1.0001f;
    cone_bounds(P, Gu, &hp, &hm);
    rej_p |= fmaf(Ly, ro, hp) < -tol_u; rej_m |= fmaf(Ly, ro, hm) < -tol_u;
    cone_bounds(P, Gv, &hp, &hm);
    rej_p |= fmaf(Lx, ro, hp) < -tol_v; rej_m |= fmaf(Lx, ro, hm) < -tol_v;
    cone_bounds(P, Gw, &hp, &hm);
    rej_p |= fmaf(Lz, ro, hm) < -tol_w; rej_m |= fmaf(Lz, ro, hp) < -tol_w;
    if (one_sided) return dn < 0.f ? rej_p : rej_m;
    return rej_p & rej_m;
}

static uint32_t rxt_cell(const float *a)
{
    const float ax = fabsf(a[0]), ay = fabsf(a[1]), az = fabsf(a[2]);
    uint32_t m = 0; float major = a[0], c1 = a[1], c2 = a[2];
    if (ay > ax && ay >= az) { m = 1; major = a[1]; c1 = a[2]; c2 = a[0]; }
    else if (az > ax && az > ay) { m = 2; major = a[2]; c1 = a[0]; c2 = a[1]; }
    const float inv = 1.f / major, u = c1 * inv, v = c2 * inv;
    int iu = (int)((u * 0.5f + 0.5f) * (float)RXT_N), iv = (int)((v * 0.5f + 0.5f) * (float)RXT_N);
    iu = iu < 0 ? 0 : (iu > RXT_N - 1 ? RXT_N - 1 : iu);
    iv = iv < 0 ? 0 : (iv > RXT_N - 1 ? RXT_N - 1 : iv);
    const uint32_t f = m + (major < 0.f ? 3u : 0u);
    return (f * RXT_N + (uint32_t)iv) * RXT_N + (uint32_t)iu;
}
static void rxt_dir(uint32_t f, double u, double v, double *out)
{
    const uint32_t m = f % 3u; const double sgn = f >= 3u ? -1.0 : 1.0; double c[3];
    c[m] = sgn; c[(m + 1u) % 3u] = u * sgn; c[(m + 2u) % 3u] = v * sgn;
    const double l = sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    out[0] = c[0] / l; out[1] = c[1] / l; out[2] = c[2] / l;
}
/* axis + (cos, sin) of every cube-map cell, widened by aq (problem.c rxt_bins) */
void cell_cones(float *axis3, float *cs2, double aq)
{
    for (uint32_t f = 0; f < 6; ++f) for (uint32_t iv = 0; iv < RXT_N; ++iv) for (uint32_t iu = 0; iu < RXT_N; ++iu) {
        const uint32_t cell = (f * RXT_N + iv) * RXT_N + iu;
        const double u0 = 2.0 * iu / RXT_N - 1.0, u1 = 2.0 * (iu + 1) / RXT_N - 1.0, v0 = 2.0 * iv / RXT_N - 1.0, v1 = 2.0 * (iv + 1) / RXT_N - 1.0;
        double ctr[3], q[3], cmin = 1.0;
        rxt_dir(f, 0.5 * (u0 + u1), 0.5 * (v0 + v1), ctr);
        const double us[2] = {u0, u1}, vs[2] = {v0, v1};
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) {
            rxt_dir(f, us[a], vs[b], q);
            const double cc = q[0] * ctr[0] + q[1] * ctr[1] + q[2] * ctr[2];
            if (cc < cmin) cmin = cc;
        }
        const double ang = acos(cmin > 1 ? 1 : cmin) + aq + 3e-4;
        axis3[3 * cell] = (float)ctr[0]; axis3[3 * cell + 1] = (float)ctr[1]; axis3[3 * cell + 2] = (float)ctr[2];
        cs2[2 * cell] = (float)cos(ang); cs2[2 * cell + 1] = (float)sin(ang);
    }
}

/* table of an apex: key (cell, rbin) -> W words.  Rays: origin = apex - dir * r (towards == 1: shadow
 * rays, which arrive at the apex) or origin = apex + dir * r (towards == 0: rays that leave an image
 * source), dir in the cell's cone, r in [redge[k], redge[k+1]].  ro_apex: line-point radius. */
void build_table(const float *rows, int T, const float *apex, float ro_apex, int towards, const float *axis3,
                 const float *cs2, const float *redge, int nr, uint64_t *masks /* [cells][nr][W] */)
{
    const int W = (T + 63) / 64, NC = 6 * RXT_N * RXT_N;
#pragma omp parallel for schedule(dynamic, 64)
    for (int cell = 0; cell < NC; ++cell)
        for (int k = 0; k < nr; ++k) {
            Packet P;
            const float *a = axis3 + 3 * cell;
            const double ca = cs2[2 * cell], sa = cs2[2 * cell + 1];
            const double r0 = redge[k], r1 = redge[k + 1], rm = 0.5 * (r0 + r1);
            const double sgn = towards ? -1.0 : 1.0;
            for (int c = 0; c < 3; ++c) { P.oc[c] = apex[c]; P.ax[c] = a[c]; P.bc[c] = (float)(apex[c] + sgn * a[c] * rm); }
            const double q0 = sqrt(fmax(0.0, rm * rm + r0 * r0 - 2.0 * rm * r0 * ca)), q1 = sqrt(fmax(0.0, rm * rm + r1 * r1 - 2.0 * rm * r1 * ca));
            (void)sa;
            P.br = (float)(fmax(q0, q1) * 1.001 + 1e-4 + 1e-5 * (fabs(P.bc[0]) + fabs(P.bc[1]) + fabs(P.bc[2])));
            P.ro = ro_apex; P.cosa = cs2[2 * cell]; P.sina = cs2[2 * cell + 1]; P.usable = 1;
            uint64_t *m = masks + ((size_t)cell * nr + k) * W;
            for (int w = 0; w < W; ++w) m[w] = 0;
            for (int j = 0; j < T; ++j)
                if (!packet_culls(&P, rows + (size_t)j * ROWF)) m[j >> 6] |= 1ull << (j & 63);
        }
}

static int rbin_of(float r, const float *redge, int nr)
{
    int k = 0;
    while (k + 1 < nr && r >= redge[k + 1]) ++k;
    return k;
}

/* the device's packet_bounds (shadow: apex mode) on 64 rays */
static void packet_of(const float *o, const float *d, int n, int shadow, const float *apex, Packet *P)
{
    float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
    for (int i = 0; i < n; ++i) for (int c = 0; c < 3; ++c) { lo[c] = fminf(lo[c], o[3 * i + c]); hi[c] = fmaxf(hi[c], o[3 * i + c]); }
    float ext[3];
    for (int c = 0; c < 3; ++c) { P->bc[c] = 0.5f * (lo[c] + hi[c]); ext[c] = hi[c] - lo[c]; }
    P->br = 0.5f * sqrtf(fdot(ext, ext)) * 1.0001f + 1e-6f * (fabsf(P->bc[0]) + fabsf(P->bc[1]) + fabsf(P->bc[2]));
    float sd[3] = {0, 0, 0};
    if (shadow) {
        for (int c = 0; c < 3; ++c) { P->oc[c] = apex[c]; sd[c] = apex[c] - P->bc[c]; }
        const float lmax = sqrtf(fdot(sd, sd)) * 1.0001f + P->br;
        P->ro = 8.f * (0.5f * EPS) * lmax + 1e-7f;
    } else {
        for (int c = 0; c < 3; ++c) P->oc[c] = P->bc[c];
        P->ro = P->br;
        for (int i = 0; i < n; ++i) for (int c = 0; c < 3; ++c) sd[c] += d[3 * i + c];
    }
    const float n2 = fdot(sd, sd), inv = 1.f / sqrtf(fmaxf(n2, 1e-30f));
    for (int c = 0; c < 3; ++c) P->ax[c] = sd[c] * inv;
    float cm = 1.f;
    for (int i = 0; i < n; ++i) cm = fminf(cm, fdot(d + 3 * i, P->ax));
    cm = cm * (1.f - 1e-5f) - 2e-6f;
    P->cosa = cm;
    P->sina = sqrtf(fmaxf(0.f, fmaf(-cm, cm, 1.f))) * 1.00001f + 1e-7f;
    P->usable = (n2 > 1e-12f) && (cm > 0.5f);
}

static inline int popc(const uint64_t *m, int W) { int c = 0; for (int w = 0; w < W; ++w) c += __builtin_popcountll(m[w]); return c; }

/* Shadow traces of one launch: n entries (origins o), apex rx, table masks[cells][nr][W].
 * out[0] wave-traces, [1] sum over lanes of per-lane candidates, [2] sum over waves of max-over-lanes,
 * [3] sum over waves of union, [4] sum over waves of the PACKET test's candidates (today's kernel),
 * [5] waves whose packet is unusable, [6] sum over waves of the 2nd largest... (unused)
 * hist_max[257], hist_union[257], hist_packet[257]: histograms over waves */
void shadow_eval(const float *rows, int T, const float *o, int n, const float *apex, const uint64_t *masks,
                 const float *redge, int nr, double *out, int64_t *hist_max, int64_t *hist_union, int64_t *hist_packet)
{
    const int W = (T + 63) / 64, nw = (n + 63) / 64;
    double s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : s1, s2, s3, s4, s5)
    for (int w = 0; w < nw; ++w) {
        const int i0 = w * 64, cnt = (n - i0 < 64) ? n - i0 : 64;
        float d[192];
        uint64_t uni[8] = {0};
        int mx = 0;
        for (int l = 0; l < cnt; ++l) {
            const float *oo = o + 3 * (size_t)(i0 + l);
            float wv[3] = {apex[0] - oo[0], apex[1] - oo[1], apex[2] - oo[2]};
            const float r = sqrtf((wv[0] * wv[0] + wv[1] * wv[1]) + wv[2] * wv[2]);
            for (int c = 0; c < 3; ++c) d[3 * l + c] = wv[c] / r;
            const uint32_t cell = rxt_cell(d + 3 * l);
            const int k = rbin_of(r, redge, nr);
            const uint64_t *m = masks + ((size_t)cell * nr + k) * W;
            const int pc = popc(m, W);
            s1 += pc;
            if (pc > mx) mx = pc;
            for (int q = 0; q < W; ++q) uni[q] |= m[q];
        }
        const int un = popc(uni, W);
        Packet P;
        packet_of(o + 3 * (size_t)i0, d, cnt, 1, apex, &P);
        int pk = T;
        if (P.usable) {
            pk = 0;
            for (int j = 0; j < T; ++j) pk += !packet_culls(&P, rows + (size_t)j * ROWF);
        } else s5 += 1;
        s2 += mx; s3 += un; s4 += pk;
#pragma omp atomic
        hist_max[mx > 256 ? 256 : mx]++;
#pragma omp atomic
        hist_union[un > 256 ? 256 : un]++;
#pragma omp atomic
        hist_packet[pk > 256 ? 256 : pk]++;
    }
    out[0] = nw; out[1] = s1; out[2] = s2; out[3] = s3; out[4] = s4; out[5] = s5;
}

/* Bounce traces (rays o, d): today's packet test per wave; and a grid design: key (spatial cell of o on a
 * gx*gy*gz grid over [lo, hi], direction cell on a cube map of dn cells per edge) -> built lazily here by
 * the packet test on (cell box, cell cone).  out as above. */
void bounce_eval_packet(const float *rows, int T, const float *o, const float *d, int n, double *out, int64_t *hist_packet)
{
    const int nw = (n + 63) / 64;
    double s4 = 0, s5 = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : s4, s5)
    for (int w = 0; w < nw; ++w) {
        const int i0 = w * 64, cnt = (n - i0 < 64) ? n - i0 : 64;
        Packet P;
        packet_of(o + 3 * (size_t)i0, d + 3 * (size_t)i0, cnt, 0, NULL, &P);
        int pk = T;
        if (P.usable) {
            pk = 0;
            for (int j = 0; j < T; ++j) pk += !packet_culls(&P, rows + (size_t)j * ROWF);
        } else s5 += 1;
        s4 += pk;
#pragma omp atomic
        hist_packet[pk > 256 ? 256 : pk]++;
    }
    out[0] = nw; out[4] = s4; out[5] = s5;
}

/* per-lane lookups of the bounce rays against an apex table (image sources): every lane has its own apex
 * index (ap[i], -1 = none -> whole table), tables[ap] -> masks.  r = |o - apex|. */
void bounce_eval_apex(int T, const float *o, const float *d, const int32_t *ap, int n, const float *apexes,
                      const uint64_t *const *tables, const float *redge, int nr, double *out, int64_t *hist_max,
                      int64_t *hist_union)
{
    const int W = (T + 63) / 64, nw = (n + 63) / 64;
    double s1 = 0, s2 = 0, s3 = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : s1, s2, s3)
    for (int w = 0; w < nw; ++w) {
        const int i0 = w * 64, cnt = (n - i0 < 64) ? n - i0 : 64;
        uint64_t uni[8] = {0};
        int mx = 0;
        for (int l = 0; l < cnt; ++l) {
            const size_t i = (size_t)(i0 + l);
            int pc = T;
            if (ap[i] >= 0) {
                const float *a = apexes + 3 * ap[i];
                const float wv[3] = {o[3 * i] - a[0], o[3 * i + 1] - a[1], o[3 * i + 2] - a[2]};
                const float r = sqrtf((wv[0] * wv[0] + wv[1] * wv[1]) + wv[2] * wv[2]);
                const uint32_t cell = rxt_cell(d + 3 * i);
                const int k = rbin_of(r, redge, nr);
                const uint64_t *m = tables[ap[i]] + ((size_t)cell * nr + k) * W;
                pc = popc(m, W);
                for (int q = 0; q < W; ++q) uni[q] |= m[q];
            } else {
                for (int q = 0; q < W; ++q) uni[q] = ~0ull;
            }
            s1 += pc;
            if (pc > mx) mx = pc;
        }
        int un = popc(uni, W);
        if (un > T) un = T;
        s2 += mx; s3 += un;
#pragma omp atomic
        hist_max[mx > 256 ? 256 : mx]++;
#pragma omp atomic
        hist_union[un > 256 ? 256 : un]++;
    }
    out[0] = nw; out[1] = s1; out[2] = s2; out[3] = s3;
}

/* ---------------- patch tables: key = (source triangle A, cell of the (u, v) grid on A) ----------------
 * A ray origin o that lies on triangle A (within hmax of its plane, inside its inflated outline) falls into
 * one cell of A's (nu x nv) grid in the (e1, e2) basis; per (cell, apex) a candidate mask built by the packet
 * test on { origins in the cell's ball, lines meeting ball(apex, ro) }.  pdef per triangle: nu, nv, base. */
typedef struct { int nu, nv, base; } PatchDef;

static void patch_packet(const float *r, const PatchDef *pd, int iu, int iv, const float *apex, float ro_apex, int towards,
                         float slack, Packet *P)
{
    /* cell = v1 + [iu, iu+1]/nu e1 + [iv, iv+1]/nv e2: centre and half-diagonal */
    const double u0 = (iu + 0.5) / pd->nu, v0 = (iv + 0.5) / pd->nv;
    double c[3], hd[3], hd2[3];
    for (int k = 0; k < 3; ++k) {
        c[k] = r[k] + u0 * r[3 + k] + v0 * r[6 + k];
        hd[k] = 0.5 * r[3 + k] / pd->nu + 0.5 * r[6 + k] / pd->nv;
        hd2[k] = 0.5 * r[3 + k] / pd->nu - 0.5 * r[6 + k] / pd->nv;
    }
    const double rad = fmax(sqrt(hd[0] * hd[0] + hd[1] * hd[1] + hd[2] * hd[2]), sqrt(hd2[0] * hd2[0] + hd2[1] * hd2[1] + hd2[2] * hd2[2]));
    double w[3] = {apex[0] - c[0], apex[1] - c[1], apex[2] - c[2]};
    const double dist = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    const double br = rad * 1.001 + slack;
    for (int k = 0; k < 3; ++k) { P->bc[k] = (float)c[k]; P->oc[k] = apex[k]; P->ax[k] = (float)((towards ? 1.0 : -1.0) * w[k] / fmax(dist, 1e-30)); }
    P->br = (float)br;
    P->ro = ro_apex;
    double s = (br + ro_apex) / fmax(dist, 1e-30) * 1.001 + 1e-5;
    if (s >= 0.86) { P->usable = 0; s = 0.86; } else P->usable = 1;   /* apex inside / next to the cell: no cone */
    P->sina = (float)s;
    P->cosa = (float)(sqrt(1.0 - s * s) * 0.9999);
}

void build_patch_table(const float *rows, int T, const PatchDef *pd, int npatch, const float *apex, int apex_per_tri, float ro_apex, int towards,
                       float slack, uint64_t *masks /* [npatch][W] */)
{
    const int W = (T + 63) / 64;
#pragma omp parallel for schedule(dynamic, 1)
    for (int a = 0; a < T; ++a)
        for (int iv = 0; iv < pd[a].nv; ++iv)
            for (int iu = 0; iu < pd[a].nu; ++iu) {
                uint64_t *m = masks + (size_t)(pd[a].base + iv * pd[a].nu + iu) * W;
                Packet P;
                patch_packet(rows + (size_t)a * ROWF, &pd[a], iu, iv, apex_per_tri ? apex + 3 * a : apex, ro_apex, towards, slack, &P);
                for (int w = 0; w < W; ++w) m[w] = 0;
                for (int j = 0; j < T; ++j)
                    if (!P.usable || !packet_culls(&P, rows + (size_t)j * ROWF)) m[j >> 6] |= 1ull << (j & 63);
            }
    (void)npatch;
}

/* patch of origin o on triangle a: -1 if o is not (provably) inside the cell's ball */
static int patch_of(const float *rows, const PatchDef *pd, int a, const float *o, float slack)
{
    const float *r = rows + (size_t)a * ROWF;
    const double s[3] = {o[0] - r[0], o[1] - r[1], o[2] - r[2]};
    const double *dummy = s; (void)dummy;
    const double e1[3] = {r[3], r[4], r[5]}, e2[3] = {r[6], r[7], r[8]}, n[3] = {r[9], r[10], r[11]};
    const double h = s[0] * n[0] + s[1] * n[1] + s[2] * n[2];
    if (!(fabs(h) <= 0.5 * slack)) return -1;
    const double a11 = e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2], a12 = e1[0] * e2[0] + e1[1] * e2[1] + e1[2] * e2[2];
    const double a22 = e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2];
    const double b1 = s[0] * e1[0] + s[1] * e1[1] + s[2] * e1[2], b2 = s[0] * e2[0] + s[1] * e2[1] + s[2] * e2[2];
    const double det = a11 * a22 - a12 * a12;
    const double u = (b1 * a22 - b2 * a12) / det, v = (b2 * a11 - b1 * a12) / det;
    int iu = (int)floor(u * pd[a].nu), iv = (int)floor(v * pd[a].nv);
    /* outside the grid by more than the slack: not served */
    const double tu = 0.4 * slack / sqrt(a11), tv = 0.4 * slack / sqrt(a22);
    if (u < -tu || v < -tv || u > 1 + tu || v > 1 + tv) return -1;
    iu = iu < 0 ? 0 : (iu >= pd[a].nu ? pd[a].nu - 1 : iu);
    iv = iv < 0 ? 0 : (iv >= pd[a].nv ? pd[a].nv - 1 : iv);
    return pd[a].base + iv * pd[a].nu + iu;
}

/* per-lane lookups of patch tables: tri[i] = source triangle of entry i; out as shadow_eval (no packet column);
 * out[5] = lanes not served */
void patch_eval(const float *rows, int T, const PatchDef *pd, const float *o, const uint32_t *tri, int n, const uint64_t *masks,
                float slack, double *out, int64_t *hist_max, int64_t *hist_union)
{
    const int W = (T + 63) / 64, nw = (n + 63) / 64;
    double s1 = 0, s2 = 0, s3 = 0, s5 = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : s1, s2, s3, s5)
    for (int w = 0; w < nw; ++w) {
        const int i0 = w * 64, cnt = (n - i0 < 64) ? n - i0 : 64;
        uint64_t uni[8] = {0};
        int mx = 0;
        for (int l = 0; l < cnt; ++l) {
            const size_t i = (size_t)(i0 + l);
            const int p = patch_of(rows, pd, (int)tri[i], o + 3 * i, slack);
            int pc = T;
            if (p >= 0) {
                const uint64_t *m = masks + (size_t)p * W;
                pc = popc(m, W);
                for (int q = 0; q < W; ++q) uni[q] |= m[q];
            } else {
                s5 += 1;
                for (int q = 0; q < W; ++q) uni[q] = ~0ull;
            }
            s1 += pc;
            if (pc > mx) mx = pc;
        }
        int un = popc(uni, W);
        if (un > T) un = T;
        s2 += mx; s3 += un;
#pragma omp atomic
        hist_max[mx > 256 ? 256 : mx]++;
#pragma omp atomic
        hist_union[un > 256 ? 256 : un]++;
    }
    out[0] = nw; out[1] = s1; out[2] = s2; out[3] = s3; out[5] = s5;
}

/* bounce traces with the wave split into groups of equal `group id` (the reflection sequence of each ray):
 * per wave, cost model  sum over groups ( overhead + per_cand * candidates(group packet) ), an unusable
 * group costing per_cand * T.  mode 0: never split; 1: split only waves whose whole packet is wide
 * (cos < cos_min) or unusable; 2: always split.  out: [0] waves, [1] total cost, [2] groups walked,
 * [3] candidates walked, [4] unusable (sub)packets */
void bounce_eval_split(const float *rows, int T, const float *o, const float *d, const int32_t *gid, int n, int mode,
                       float cos_min, double overhead, double per_cand, double *out)
{
    const int nw = (n + 63) / 64;
    double cost = 0, groups = 0, cands = 0, unus = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : cost, groups, cands, unus)
    for (int w = 0; w < nw; ++w) {
        const int i0 = w * 64, cnt = (n - i0 < 64) ? n - i0 : 64;
        Packet P;
        packet_of(o + 3 * (size_t)i0, d + 3 * (size_t)i0, cnt, 0, NULL, &P);
        int split = mode == 2 || (mode == 1 && (!P.usable || P.cosa < cos_min));
        if (!split) {
            int pk = T;
            if (P.usable) { pk = 0; for (int j = 0; j < T; ++j) pk += !packet_culls(&P, rows + (size_t)j * ROWF); } else unus += 1;
            cost += overhead + per_cand * pk; groups += 1; cands += pk;
            continue;
        }
        int done[64] = {0};
        for (int l = 0; l < cnt; ++l) {
            if (done[l]) continue;
            float go[192], gd[192];
            int m = 0;
            for (int k = l; k < cnt; ++k)
                if (!done[k] && gid[i0 + k] == gid[i0 + l]) {
                    done[k] = 1;
                    memcpy(go + 3 * m, o + 3 * (size_t)(i0 + k), 12);
                    memcpy(gd + 3 * m, d + 3 * (size_t)(i0 + k), 12);
                    ++m;
                }
            Packet G;
            packet_of(go, gd, m, 0, NULL, &G);
            int pk = T;
            if (G.usable) { pk = 0; for (int j = 0; j < T; ++j) pk += !packet_culls(&G, rows + (size_t)j * ROWF); } else unus += 1;
            cost += overhead + per_cand * pk; groups += 1; cands += pk;
        }
    }
    out[0] = nw; out[1] = cost; out[2] = groups; out[3] = cands; out[4] = unus;
}
