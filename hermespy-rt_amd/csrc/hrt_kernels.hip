// hrt_kernels.hip -- the compute_paths hot loop as HIP kernels for gfx950 (MI355X, CDNA4),
// plus the thin C-ABI shim the host C code calls (hrt_kparams.h).
//
// What runs here (reference lines relative to the reference repository):
//   closest_hit        src/compute_paths.c:237-287  moeller_trumbore (brute force, all triangles)
//   incidence_angle    src/compute_paths.c:281-283
//   fresnel            src/compute_paths.c:300-344  refl_coefs (ITU-R P.2040-3 eq. 31a/31b)
//   scatter_pattern    src/compute_paths.c:359-415  scat_coefs
//   hrt_bounce_kernel  src/compute_paths.c:460-466 (state init), :596-723 (one bounce:
//                      trace, Fresnel, FSL, reflect, scatter to every RX)
//   hrt_los_kernel     src/compute_paths.c:515-577
//
// Design (MI355X-first, not the reference's loop nest):
//   * One ray per lane, wave64.  Launch b of the bounce kernel takes the COMPACT live list
//     produced by launch b-1 (the rays that hit at bounce b-1, with their post-reflection
//     state), first casts their num_rx shadow rays and writes the scatter records of bounce
//     b-1 (every lane busy: a live ray always owes its records), then traces bounce b and
//     appends the survivors to the next live list with a wave ballot + prefix count and ONE
//     atomic per wave.  Every field of the lists is a separate cap-long array, so a wave
//     reads and writes 256-byte contiguous runs.
//   * The triangle table (v1, e1, e2, n, mesh id: 64 B per triangle) is staged once per
//     workgroup in LDS; the triangle index is wave-uniform, so the inner loop's three
//     ds_read_b128 are broadcasts (no bank conflicts), and the material table and RX
//     positions sit next to it.  Scenes larger than the LDS budget fall back to reading the
//     same table with wave-uniform (scalar-cache) loads.
//   * Geometry is IEEE-exact and contraction-free (built with -ffp-contract=off, correctly
//     rounded division/sqrt, denormals on): hit decisions, hit indices, reflected rays and
//     delays are BIT-IDENTICAL to the C reference.  The float libm calls of the shading code
//     (sinf/cosf/expf/acosf) are bit-exact restatements of glibc 2.35's (hrt_libm.h, pinned
//     exhaustively against the host libm by oracle/libm_probe.c), so amplitudes are
//     bit-identical too; the one double-precision call, acos for the incidence angle, uses the
//     device library and is rounded to float (agrees with glibc except with probability
//     ~2^-29 per evaluation).
//   * No MFMA: this is branchy intersection, not a contraction.
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stdio.h>

#include "hrt_kparams.h"
#include "hrt_libm.h"

#pragma clang fp contract(off)

namespace {

constexpr float kEps = 1.1920928955078125e-07f;             // FLT_EPSILON
constexpr float kOnePlusEps = 1.00000011920928955078125f;   // next float after 1
constexpr float kPi = 3.14159265358979323846f;              // src/compute_paths.c:18 (float)
constexpr float kC = 299792458.0f;                          // src/compute_paths.c:19

struct F3 { float x, y, z; };

__device__ __forceinline__ F3 sub3(F3 a, F3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ F3 add3(F3 a, F3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ F3 mul3(F3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
// inc/vec3.h:29-32: (x*x' + y*y') + z*z'
__device__ __forceinline__ float dot3(F3 a, F3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// inc/vec3.h:20-28
__device__ __forceinline__ F3 cross3(F3 a, F3 b)
{
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

struct Hit { uint32_t tri; float t; };

// Closest hit over the whole triangle table, lowest index wins ties (strict '<').
// `tri` points at LDS (broadcast reads) or at global memory (wave-uniform loads).
template <typename TriPtr>
__device__ __forceinline__ Hit closest_hit(TriPtr tri, uint32_t num_tri, F3 o, F3 d)
{
    float best = 1e9f;
    uint32_t who = HRT_NO_HIT;
    for (uint32_t j = 0; j < num_tri; ++j) {
        const float4 q0 = tri[4 * j], q1 = tri[4 * j + 1], q2 = tri[4 * j + 2];
        const F3 v1 = {q0.x, q0.y, q0.z};
        const F3 e1 = {q0.w, q1.x, q1.y};
        const F3 e2 = {q1.z, q1.w, q2.x};
        const F3 pv = cross3(d, e2);
        const float det = dot3(e1, pv);
        if (det > -kEps && det < kEps) continue;
        const F3 s = sub3(o, v1);
        const float u = dot3(s, pv) / det;
        if (u < -kEps || u > kOnePlusEps) continue;
        const F3 q = cross3(s, e1);
        const float v = dot3(d, q) / det;
        const float w = u + v;
        if (v < -kEps || w > kOnePlusEps) continue;
        const float dist = dot3(e2, q) / det;
        if (dist > kEps && dist < best) { best = dist; who = j; }
    }
    return {who, best};
}

template <typename TriPtr>
__device__ __forceinline__ F3 tri_normal(TriPtr tri, uint32_t j)
{
    const float4 q2 = tri[4 * j + 2];
    return {q2.y, q2.z, q2.w};
}

template <typename TriPtr>
__device__ __forceinline__ uint32_t tri_mesh(TriPtr tri, uint32_t j)
{
    return __float_as_uint(tri[4 * j + 3].x);
}

// acos in double of the float dot product, stored to float, folded to [0, pi/2] with the
// float pi (src/compute_paths.c:281-283).
__device__ __forceinline__ float incidence_angle(F3 n, F3 d)
{
    float th = (float)acos((double)dot3(n, d));
    if (th > kPi * 0.5f) th = kPi - th;   // (double)th > (double)pi_f/2. is the same test
    return th;
}

// float libm calls of the shading code: bit-exact restatements of the host libm (hrt_libm.h)
__device__ __forceinline__ float sin_f(float x) { return hrt_sinf(x); }
__device__ __forceinline__ float cos_f(float x) { return hrt_cosf(x); }
__device__ __forceinline__ float exp_f(float x) { return hrt_expf(x); }
__device__ __forceinline__ float acos_f(float x) { return hrt_acosf(x); }

// src/compute_paths.c:152-164
__device__ __forceinline__ void complex_div(float ar, float ai, float br, float bi, float &cr,
                                            float &ci)
{
    const float den = br * br + bi * bi;
    cr = (ar * br + ai * bi) / den;
    ci = (ai * br - ar * bi) / den;
}

// One row of the material table in LDS: 4 float4
//   m0 = eta_re, eta_sqrt_re, eta_inv_re, eta_inv_sqrt_re
//   m1 = eta_im, eta_sqrt_im, eta_inv_im, eta_inv_sqrt_im
//   m2 = eta_abs, eta_abs_pow2, eta_abs_inv_sqrt, r
//   m3 = s, s1_alpha, -, -
__device__ __forceinline__ void fresnel(const float4 *mat, uint32_t mi, float th, float R[4])
{
    const float4 m0 = mat[4 * mi], m1 = mat[4 * mi + 1], m2 = mat[4 * mi + 2];
    const float s1 = sin_f(th);
    if (m2.z * s1 > 1.f - kEps) {
        R[0] = R[2] = 1.f;
        R[1] = R[3] = 0.f;
        return;
    }
    const float s2 = s1 * s1;
    const float c2r = sqrtf(1.f + m0.z / m2.y * s2);
    const float c2i = sqrtf(1.f - m1.z / m2.y * s2);
    const float pr = m0.y * c2r - m1.y * c2i;
    const float pi = m0.y * c2i + m1.y * c2r;
    const float c1 = cos_f(th);
    complex_div(c1 - pr, -pi, c1 + pr, pi, R[0], R[1]);
    const float qr = m0.y * c1;
    const float qi = m1.y * c1;
    complex_div(qr - c2r, qi - c2i, qr + c2r, qi + c2i, R[2], R[3]);
    R[0] *= m2.w; R[1] *= m2.w; R[2] *= m2.w; R[3] *= m2.w;
}

__device__ __forceinline__ void scatter_pattern(const float4 *mat, uint32_t mi, float th_s,
                                                float th_i, float S[4])
{
    const float4 m3 = mat[4 * mi + 3];
    const float s = m3.x, alpha = m3.y;   // alpha: small integer held as float
    const float cs = cos_f(th_s), ci = cos_f(th_i), si = sin_f(th_i);
    const float dth = fabsf(th_s - th_i);
    const float f = s * exp_f(-alpha * dth);
    const float rough = 1.0f / (1.0f + alpha);
    const float spec = rough * cs;
    const float diff = (1.0f - rough) * cs;
    float te = f * (spec + diff);
    float tm = f * (spec * ci + diff);
    const float ph = alpha * si * 0.1f;
    const float sp = sin_f(ph);
    float tei = te * sp;
    float tmi = tm * sp;
    const float nrm = sqrtf(te * te + tei * tei + tm * tm + tmi * tmi);
    if (nrm > 1e-6f) { te /= nrm; tei /= nrm; tm /= nrm; tmi /= nrm; }
    S[0] = te; S[1] = tei; S[2] = tm; S[3] = tmi;
}

__device__ __forceinline__ uint32_t lane_prefix(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                     __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// ---- workspace addressing (include/hrt_device.h) ----
__device__ __forceinline__ float *hit_field(const hrt_kparams &P, uint32_t b, uint32_t f)
{
    return reinterpret_cast<float *>(P.ws + P.off_hits + (uint64_t)b * P.hit_block_bytes +
                                     (uint64_t)f * P.cap * 4u);
}
__device__ __forceinline__ float *rec_field(const hrt_kparams &P, uint32_t b, uint32_t rx,
                                            uint32_t f)
{
    return reinterpret_cast<float *>(P.ws + P.off_recs + (uint64_t)b * P.rec_block_bytes +
                                     ((uint64_t)rx * 9u + f) * P.cap * 4u);
}
__device__ __forceinline__ unsigned long long *mask_words(const hrt_kparams &P, uint32_t b,
                                                          uint32_t rx)
{
    return reinterpret_cast<unsigned long long *>(
        P.ws + P.off_masks + ((uint64_t)b * P.num_rx + rx) * (P.cap / 64u) * 8u);
}

enum : uint32_t {
    H_RAY = 0, H_TRI, H_THETA, H_FS0, H_OX, H_OY, H_OZ, H_DX, H_DY, H_DZ,
    H_A0, H_A1, H_A2, H_A3, H_TAU
};
enum : uint32_t { R_A0 = 0, R_A1, R_A2, R_A3, R_TAU, R_DX, R_DY, R_DZ, R_DFS };

// LDS image: [num_tri*4 float4 (if staged)] [17*4 float4 materials] [num_rx float4 RX pos]
template <bool TRI_IN_LDS>
__global__ __launch_bounds__(HRT_BLOCK) void hrt_bounce_kernel(const hrt_kparams P,
                                                               const uint32_t b)
{
    extern __shared__ float4 lds[];
    const uint32_t tid = threadIdx.x;
    const bool first = (b == 0);
    const bool do_trace = (b < P.num_bounces);
    uint32_t *counts = reinterpret_cast<uint32_t *>(P.ws + P.off_counts);
    const uint32_t n_in = first ? P.n0 : counts[b];
    if ((uint64_t)blockIdx.x * HRT_BLOCK >= n_in) return;   // whole block: nothing to do

    const uint32_t T = P.num_tri;
    const float4 *g_tri = reinterpret_cast<const float4 *>(P.tri);
    float4 *l_tri = lds;
    float4 *l_mat = lds + (TRI_IN_LDS ? 4u * T : 0u);
    float4 *l_rx = l_mat + 4u * HRT_NUM_MATERIALS;
    if (TRI_IN_LDS)
        for (uint32_t k = tid; k < 4u * T; k += HRT_BLOCK) l_tri[k] = g_tri[k];
    {
        const float4 *g_mat = reinterpret_cast<const float4 *>(P.mat);
        for (uint32_t k = tid; k < 4u * HRT_NUM_MATERIALS; k += HRT_BLOCK) l_mat[k] = g_mat[k];
        for (uint32_t k = tid; k < P.num_rx; k += HRT_BLOCK)
            l_rx[k] = make_float4(P.rx_pos[3 * k], P.rx_pos[3 * k + 1], P.rx_pos[3 * k + 2], 0.f);
    }
    __syncthreads();
    // the table the loops read: LDS image or (wave-uniform index => scalar loads) global
    auto tri = [&]() {
        if constexpr (TRI_IN_LDS) return (const float4 *)l_tri;
        else return g_tri;
    }();

    const uint32_t lane = tid & 63u;

    for (uint64_t base = (uint64_t)blockIdx.x * HRT_BLOCK; base < n_in;
         base += (uint64_t)gridDim.x * HRT_BLOCK) {
        const uint32_t i = (uint32_t)base + tid;
        const bool valid = i < n_in;

        // ---- ray state ----
        uint32_t ray = 0, htri = 0;
        float theta = 0.f, fs0 = 0.f, tau = 0.f;
        F3 o = {0.f, 0.f, 0.f}, d = {0.f, 0.f, 1.f};
        float a0 = 1.f, a1 = 0.f, a2 = 1.f, a3 = 0.f;
        if (valid) {
            if (first) {
                // src/compute_paths.c:452-466 + the launch Doppler term :494-500
                const uint32_t tx = i / P.num_local, il = i - tx * P.num_local;
                ray = i;
                o = {P.tx_pos[3 * tx], P.tx_pos[3 * tx + 1], P.tx_pos[3 * tx + 2]};
                d = {P.dirs[3 * (uint64_t)il], P.dirs[3 * (uint64_t)il + 1],
                     P.dirs[3 * (uint64_t)il + 2]};
                const F3 tv = {P.tx_vel[3 * tx], P.tx_vel[3 * tx + 1], P.tx_vel[3 * tx + 2]};
                fs0 = dot3(tv, d) * P.dop_mult;
            } else {
                const uint32_t pb = b - 1;
                ray = __float_as_uint(hit_field(P, pb, H_RAY)[i]);
                htri = __float_as_uint(hit_field(P, pb, H_TRI)[i]);
                theta = hit_field(P, pb, H_THETA)[i];
                fs0 = hit_field(P, pb, H_FS0)[i];
                o = {hit_field(P, pb, H_OX)[i], hit_field(P, pb, H_OY)[i],
                     hit_field(P, pb, H_OZ)[i]};
                d = {hit_field(P, pb, H_DX)[i], hit_field(P, pb, H_DY)[i],
                     hit_field(P, pb, H_DZ)[i]};
                a0 = hit_field(P, pb, H_A0)[i];
                a1 = hit_field(P, pb, H_A1)[i];
                a2 = hit_field(P, pb, H_A2)[i];
                a3 = hit_field(P, pb, H_A3)[i];
                tau = hit_field(P, pb, H_TAU)[i];
            }
        }

        // ---- scatter the hits of bounce b-1 to every RX, in RX order, carrying theta
        //      (src/compute_paths.c:671-723; quirks Q6, Q7, Q8) ----
        if (!first) {
            const uint32_t pb = b - 1;
            F3 n = {0.f, 0.f, 1.f}, mvel = {0.f, 0.f, 0.f};
            uint32_t mat = 0;
            if (valid) {
                n = tri_normal(g_tri, htri);
                const uint32_t mesh = tri_mesh(g_tri, htri);
                const float4 mm = reinterpret_cast<const float4 *>(P.mesh)[mesh];
                mvel = {mm.x, mm.y, mm.z};
                mat = __float_as_uint(mm.w);
            }
            for (uint32_t rx = 0; rx < P.num_rx; ++rx) {
                bool unblocked = false;
                if (valid) {
                    const float4 rp = l_rx[rx];
                    F3 w = sub3({rp.x, rp.y, rp.z}, o);
                    const float d2rx = sqrtf(dot3(w, w));
                    w = {w.x / d2rx, w.y / d2rx, w.z / d2rx};
                    const Hit sh = closest_hit(tri, T, o, w);
                    if (sh.tri != HRT_NO_HIT) theta = incidence_angle(tri_normal(tri, sh.tri), w);
                    if (sh.tri != HRT_NO_HIT && sh.t <= 1.f) {
                        rec_field(P, pb, rx, R_A0)[i] = 0.f;
                        rec_field(P, pb, rx, R_A1)[i] = 0.f;
                        rec_field(P, pb, rx, R_A2)[i] = 0.f;
                        rec_field(P, pb, rx, R_A3)[i] = 0.f;
                        rec_field(P, pb, rx, R_TAU)[i] = 0.f;
                    } else {
                        unblocked = true;
                        const float th_s = acos_f(dot3(w, n));
                        float S[4];
                        scatter_pattern(l_mat, mat, th_s, theta, S);
                        float o0 = a0 * S[0] - a1 * S[1];
                        float o1 = a0 * S[1] + a1 * S[0];
                        float o2 = a2 * S[2] - a3 * S[3];
                        float o3 = a2 * S[3] + a3 * S[2];
                        float f2 = P.fsl_mult * d2rx;
                        f2 *= f2;
                        if (f2 > 1.f) { o0 /= f2; o1 /= f2; o2 /= f2; o3 /= f2; }
                        rec_field(P, pb, rx, R_A0)[i] = o0;
                        rec_field(P, pb, rx, R_A1)[i] = o1;
                        rec_field(P, pb, rx, R_A2)[i] = o2;
                        rec_field(P, pb, rx, R_A3)[i] = o3;
                        rec_field(P, pb, rx, R_TAU)[i] = tau + d2rx / kC;
                        rec_field(P, pb, rx, R_DX)[i] = -w.x;
                        rec_field(P, pb, rx, R_DY)[i] = -w.y;
                        rec_field(P, pb, rx, R_DZ)[i] = -w.z;
                        rec_field(P, pb, rx, R_DFS)[i] = dot3(sub3(w, d), mvel) * P.dop_mult;
                    }
                }
                const unsigned long long m = __ballot(unblocked);
                if (lane == 0 && valid) mask_words(P, pb, rx)[i >> 6] = m;
            }
        }

        // ---- trace bounce b (src/compute_paths.c:611-659) ----
        if (do_trace) {
            bool hit = false;
            uint32_t ntri = 0;
            float nth = 0.f;
            if (valid) {
                const Hit h = closest_hit(tri, T, o, d);
                if (h.tri != HRT_NO_HIT) {
                    hit = true;
                    ntri = h.tri;
                    const F3 n = tri_normal(tri, h.tri);
                    nth = incidence_angle(n, d);
                    const uint32_t mesh = tri_mesh(tri, h.tri);
                    const uint32_t mat =
                        __float_as_uint(reinterpret_cast<const float4 *>(P.mesh)[mesh].w);
                    float R[4];
                    fresnel(l_mat, mat, nth, R);
                    float fsl = P.fsl_mult * h.t;
                    fsl *= fsl;
                    if (fsl > 1.f) { R[0] /= fsl; R[1] /= fsl; R[2] /= fsl; R[3] /= fsl; }
                    const float b0 = a0 * R[0] - a1 * R[1];
                    const float b1 = a0 * R[1] + a1 * R[0];
                    const float b2 = a2 * R[2] - a3 * R[3];
                    const float b3 = a2 * R[3] + a3 * R[2];
                    a0 = b0; a1 = b1; a2 = b2; a3 = b3;
                    tau += h.t / kC;
                    o = add3(mul3(d, h.t), o);
                    const float dn = dot3(d, n);
                    d = sub3(d, mul3(n, 2.f * dn));
                    o = add3(o, mul3(d, 1e-4f));
                }
            }
            // stream compaction of the survivors: ballot + prefix count, one atomic per wave
            const unsigned long long m = __ballot(hit);
            if (m) {
                uint32_t wbase = 0;
                if (lane == 0) wbase = atomicAdd(&counts[b + 1], (uint32_t)__popcll(m));
                wbase = __shfl(wbase, 0);
                if (hit) {
                    const uint32_t k = wbase + lane_prefix(m);
                    hit_field(P, b, H_RAY)[k] = __uint_as_float(ray);
                    hit_field(P, b, H_TRI)[k] = __uint_as_float(ntri);
                    hit_field(P, b, H_THETA)[k] = nth;
                    hit_field(P, b, H_FS0)[k] = fs0;
                    hit_field(P, b, H_OX)[k] = o.x;
                    hit_field(P, b, H_OY)[k] = o.y;
                    hit_field(P, b, H_OZ)[k] = o.z;
                    hit_field(P, b, H_DX)[k] = d.x;
                    hit_field(P, b, H_DY)[k] = d.y;
                    hit_field(P, b, H_DZ)[k] = d.z;
                    hit_field(P, b, H_A0)[k] = a0;
                    hit_field(P, b, H_A1)[k] = a1;
                    hit_field(P, b, H_A2)[k] = a2;
                    hit_field(P, b, H_A3)[k] = a3;
                    hit_field(P, b, H_TAU)[k] = tau;
                }
            }
        }
    }
}

// LoS pass, one lane per (rx, tx) pair (src/compute_paths.c:515-577).  Tiny: one workgroup.
// Output per pair: HRT_LOS_FLOATS floats {status, a, tau, dir_tx xyz, freq_shift, -}.
__global__ __launch_bounds__(HRT_BLOCK) void hrt_los_kernel(const hrt_kparams P)
{
    const float4 *tri = reinterpret_cast<const float4 *>(P.tri);
    float *out = reinterpret_cast<float *>(P.ws + P.off_los);
    const uint32_t n = P.num_rx * P.num_tx;
    for (uint32_t off = threadIdx.x; off < n; off += HRT_BLOCK) {
        const uint32_t rx = off / P.num_tx, tx = off - rx * P.num_tx;
        const F3 o = {P.tx_pos[3 * tx], P.tx_pos[3 * tx + 1], P.tx_pos[3 * tx + 2]};
        const F3 r = {P.rx_pos[3 * rx], P.rx_pos[3 * rx + 1], P.rx_pos[3 * rx + 2]};
        const F3 d = sub3(r, o);
        float *q = out + 8u * off;
        uint32_t status;
        float a = 0.f, tau = 0.f, fs = 0.f;
        F3 u = {0.f, 0.f, 0.f};
        if (dot3(d, d) < kEps) {
            status = 0u;   // coincident: unit gain, zero delay (:531-544)
            a = 1.f;
        } else {
            // per-lane triangle loop over the global table (different rays per lane, same
            // triangle index: still wave-uniform addresses)
            const Hit h = closest_hit(tri, P.num_tri, o, d);
            if (h.tri != HRT_NO_HIT && h.t <= 1.f) {
                status = 1u;   // blocked (:548-554)
            } else {
                status = 2u;
                const float dist = sqrtf(dot3(d, d));
                u = {d.x / dist, d.y / dist, d.z / dist};
                const float fsl = P.fsl_mult * dist;   // linear, not squared (quirk Q4)
                a = (fsl > 1.f) ? 1.f / fsl : 1.f;
                tau = dist / kC;
                // quirk Q5: always the FIRST tx / rx velocity
                const F3 tv = {P.tx_vel[0], P.tx_vel[1], P.tx_vel[2]};
                const F3 rv = {P.rx_vel[0], P.rx_vel[1], P.rx_vel[2]};
                fs = (dot3(tv, u) - dot3(rv, u)) * P.dop_mult;
            }
        }
        q[0] = __uint_as_float(status);
        q[1] = a; q[2] = tau; q[3] = u.x; q[4] = u.y; q[5] = u.z; q[6] = fs; q[7] = 0.f;
    }
}

// evaluates one of the hrt_libm.h functions (or the incidence-angle acos) over an array: the
// GPU side of tests/test_gpu_libm.py
__global__ void hrt_selftest_math_kernel(int fn, const float *in, float *out, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = in[i];
    float y;
    switch (fn) {
    case 0: y = hrt_sinf(x); break;
    case 1: y = hrt_cosf(x); break;
    case 2: y = hrt_expf(x); break;
    case 3: y = hrt_acosf(x); break;
    default: {   // src/compute_paths.c:281-283 with dot(n, d) = x
        float th = (float)acos((double)x);
        if (th > kPi * 0.5f) th = kPi - th;
        y = th;
    }
    }
    out[i] = y;
}

thread_local char g_err[256];

}  // namespace

// =====================================================================================
// The shim: plain C entry points over the HIP runtime (hrt_kparams.h).
// =====================================================================================
extern "C" {

int hrt_hip_device_count(int *n) { return (int)hipGetDeviceCount(n); }
int hrt_hip_set_device(int dev) { return (int)hipSetDevice(dev); }
int hrt_hip_malloc(void **p, uint64_t bytes) { return (int)hipMalloc(p, bytes ? bytes : 1); }
int hrt_hip_free(void *p) { return (int)hipFree(p); }
int hrt_hip_h2d(void *dst, const void *src, uint64_t bytes)
{
    return (int)hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
}
int hrt_hip_d2h(void *dst, const void *src, uint64_t bytes)
{
    return (int)hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost);
}
int hrt_hip_memset_async(void *dst, int value, uint64_t bytes, void *stream)
{
    return (int)hipMemsetAsync(dst, value, bytes, (hipStream_t)stream);
}
int hrt_hip_stream_sync(void *stream) { return (int)hipStreamSynchronize((hipStream_t)stream); }
int hrt_hip_mem_info(uint64_t *free_b, uint64_t *total_b)
{
    size_t f = 0, t = 0;
    const int rc = (int)hipMemGetInfo(&f, &t);
    *free_b = f;
    *total_b = t;
    return rc;
}

int hrt_hip_launch_los(const hrt_kparams *P, void *stream)
{
    hipLaunchKernelGGL(hrt_los_kernel, dim3(1), dim3(HRT_BLOCK), 0, (hipStream_t)stream, *P);
    return (int)hipGetLastError();
}

int hrt_hip_launch_bounce(const hrt_kparams *P, uint32_t bounce, void *stream)
{
    // Shapes are validated by the host (hrt_trace); here only the launch geometry.
    const uint64_t n_max = (bounce == 0) ? P->n0 : P->cap;
    uint64_t blocks = (n_max + HRT_BLOCK - 1) / HRT_BLOCK;
    if (blocks > HRT_MAX_GRID) blocks = HRT_MAX_GRID;
    if (blocks == 0) blocks = 1;
    const uint64_t tri_bytes = (uint64_t)P->num_tri * HRT_TRI_FLOATS * 4u;
    const bool in_lds = tri_bytes <= HRT_LDS_TRI_BYTES_MAX;
    const size_t small = (size_t)(HRT_NUM_MATERIALS * HRT_MAT_FLOATS * 4u) +
                         (size_t)P->num_rx * 16u;
    if (in_lds) {
        const size_t lds = (size_t)tri_bytes + small;
        if (lds > 64u * 1024u) {
            const hipError_t e = hipFuncSetAttribute(
                reinterpret_cast<const void *>(&hrt_bounce_kernel<true>),
                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL(hrt_bounce_kernel<true>, dim3((uint32_t)blocks), dim3(HRT_BLOCK), lds,
                           (hipStream_t)stream, *P, bounce);
    } else {
        hipLaunchKernelGGL(hrt_bounce_kernel<false>, dim3((uint32_t)blocks), dim3(HRT_BLOCK),
                           small, (hipStream_t)stream, *P, bounce);
    }
    return (int)hipGetLastError();
}

int hrt_hip_selftest_math(int fn, const float *d_in, float *d_out, uint64_t n, void *stream)
{
    if (n == 0) return 0;
    hipLaunchKernelGGL(hrt_selftest_math_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, fn, d_in, d_out, n);
    return (int)hipGetLastError();
}

int hrt_hip_event_create(void **ev)
{
    hipEvent_t e;
    const int rc = (int)hipEventCreate(&e);
    *ev = (void *)e;
    return rc;
}
int hrt_hip_event_destroy(void *ev) { return (int)hipEventDestroy((hipEvent_t)ev); }
int hrt_hip_event_record(void *ev, void *stream)
{
    return (int)hipEventRecord((hipEvent_t)ev, (hipStream_t)stream);
}
int hrt_hip_event_elapsed_ms(void *start, void *stop, float *ms)
{
    return (int)hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop);
}
const char *hrt_hip_error_string(int err)
{
    snprintf(g_err, sizeof g_err, "HIP error %d: %s", err, hipGetErrorString((hipError_t)err));
    return g_err;
}

}  // extern "C"
