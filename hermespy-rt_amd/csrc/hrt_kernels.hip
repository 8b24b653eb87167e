// hrt_kernels.hip -- the compute_paths hot loop as HIP kernels for gfx950 (MI355X, CDNA4),
// plus the thin C-ABI shim the host C code calls (hrt_kparams.h).
//
// What runs here (reference lines relative to the reference repository):
//   closest_hit        src/compute_paths.c:237-287  moeller_trumbore (brute force, all triangles)
//   incidence_angle    src/compute_paths.c:281-283
//   fresnel            src/compute_paths.c:300-344  refl_coefs (ITU-R P.2040-3 eq. 31a/31b)
//   scatter_pattern    src/compute_paths.c:359-415  scat_coefs
//   hrt_trace_kernel   the moeller_trumbore calls of :615 and :682 (all traces of a launch)
//   hrt_shade_kernel   src/compute_paths.c:460-466 (state init), :616-664 (Fresnel, FSL,
//                      reflect), :671-723 (scatter records)
//   hrt_fused_kernel   both of the above for one launch in ONE kernel (launch 0 always; every launch on
//                      tables of <= 64 triangles; body: hrt_fused_body.inc), hrt_chain_kernel: the tail of
//                      such launches as one persistent kernel with grid barriers
//   hrt_records_kernel :671-723 on tables of 65-256 triangles (patch tables): the shadow traces and the
//                      scatter records of a launch; hrt_image_kernel: the primary rays of launch 1 there
//   hrt_los_kernel     src/compute_paths.c:515-577
//
// Design (MI355X-first, not the reference's loop nest):
//   * The general form (any table; the small tables use the kernels above):
//     per launch b two kernels over the COMPACT live list of launch b-1 (the rays that hit at
//     bounce b-1, with their post-reflection state; at b = 0 the launch set in a coherent order).
//     hrt_trace_kernel does all intersection work -- per entry the num_rx shadow rays of bounce
//     b-1 and the ray of bounce b -- and hrt_shade_kernel everything per ray that is not
//     intersection (scatter records in RX order with the reference's theta carry, Fresnel, free-
//     space loss, reflection), writing the survivors in order into the next live list (stable
//     compaction from per-chunk counts: no scan pass, no staging copy).  Every field of the
//     lists is a separate cap-long array, addressed through buffer descriptors, so a wave reads
//     and writes 256-byte contiguous runs.
//   * In the trace kernel a wavefront is a RAY PACKET: wave reductions (DPP, inline asm) bound its
//     64 origins and directions; the lanes then test 64 TRIANGLES at a time against the packet
//     (a provably conservative test in numerator space, FMAs allowed: it only bounds) and only
//     the surviving triangles are walked, every lane testing its own ray with the reference's
//     exact float sequence behind division-free certain-reject stages.
//   * The triangle table (80 B rows, conflict-free for per-lane gathers) is staged once per
//     workgroup in LDS when it is at most 40 KiB, gathered from global memory / L2 otherwise
//     (occupancy is worth more than LDS latency); the material table and the RX positions sit in
//     LDS.
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
#include <string.h>

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "hrt_kparams.h"
#include "hrt_libm.h"

#pragma clang fp contract(off)

namespace {

constexpr float kEps = 1.1920928955078125e-07f;             // FLT_EPSILON
constexpr float kOnePlusEps = 1.00000011920928955078125f;   // next float after 1
constexpr float kPi = 3.14159265358979323846f;              // src/compute_paths.c:18 (float)
constexpr float kC = 299792458.0f;                          // src/compute_paths.c:19
constexpr uint32_t kErrFuseTimeout = HRT_ERR_FUSE_TIMEOUT;   // bit of the trace's error word (counts[nb + 1]): see lb_exclusive
constexpr uint32_t kErrChainTimeout = HRT_ERR_CHAIN_TIMEOUT;   // ... the same from hrt_chain_kernel (its grid was not resident)
constexpr uint32_t kErrVoid = kErrFuseTimeout | kErrChainTimeout;   // either: the step is void, every kernel of it returns at once

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

// One triangle = HRT_ROW float4 (80 bytes): v1.xyz e1.x | e1.yz e2.xy | e2.z n.xyz |
// E_d c_uv c_w |e1| | |e2| |e2-e1| |e1 x e2| mesh.  80 B is deliberate: when the 64 lanes of a wave each read the
// row of a DIFFERENT triangle (packet culling), the 16-lane groups of a ds_read_b128 start at
// banks 20*l mod 64 -- 16 distinct multiples of 4 -- so the gather is conflict-free (a 64-byte
// row gives 4-way conflicts: 76 % of all LDS cycles before the change).
constexpr uint32_t HRT_ROW = HRT_TRI_FLOATS / 4;

// Closest hit over the whole triangle table, lowest index wins ties (strict '<').
// `tri` points at LDS (broadcast reads) or at global memory (wave-uniform loads).
//
// Variant 0 ("plain"): the reference's test sequence as written -- three IEEE divisions per
// triangle that survives the early-outs, per-lane divergent early-outs.
// The table is in a spatial order (csrc/host/accel.c), so "lowest index wins ties" of the
// reference's scan becomes: track the lexicographic minimum of (distance, ORIGINAL index) --
// `orig[j]` is the position of row j in the reference's loop order.  who_o starts at 0, so the
// tie rule can never fire before a first hit (dist == 1e9f is not a hit in the reference either).
template <typename TriPtr>
__device__ __forceinline__ Hit closest_hit_plain(TriPtr tri, const uint32_t *__restrict__ orig,
                                                 uint32_t num_tri, F3 o, F3 d)
{
    float best = 1e9f;
    uint32_t who = HRT_NO_HIT, who_o = 0u;
    for (uint32_t j = 0; j < num_tri; ++j) {
        const float4 q0 = tri[HRT_ROW * j], q1 = tri[HRT_ROW * j + 1], q2 = tri[HRT_ROW * j + 2];
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
        if (dist > kEps && (dist < best || (dist == best && orig[j] < who_o))) {
            best = dist; who = j; who_o = orig[j];
        }
    }
    return {who, best};
}

// Variant 1 ("staged"): the same decisions with the divisions moved behind division-free
// CERTAIN-REJECT tests on the numerators, and every skip wave-uniform.
//
// With a = |det| and N' = N * sign(det) (sign-bit xor; IEEE division is sign-symmetric, so
// fl(N/det) == fl(N'/a) bit for bit), each reference test "fl(N'/a) < c" is implied by
// "N' < fl(k*a)" for a constant k a few 2^-20 beyond c: the product's rounding (2^-24
// relative) cannot bridge the margin and fl() is monotone.  A triangle is dropped only when
// one of these certain-reject conditions holds for EVERY lane of the wave (wave-uniform
// branch: no divergence, and an instruction costs the same for 1 or 64 active lanes anyway);
// whatever survives goes through the reference's exact sequence, so accepted hits, their
// distances and the lowest-index tie-break are bit-identical to variant 0.  Rejections are
// written in "reject if <condition>" form so NaNs (never rejected by the reference's
// comparisons) are not rejected here either.  Derivations: DESIGN.md, "staged intersection".
//   R1  u < -eps        <=  Nu' < -fl(k1*a)            k1 = eps*(1+2^-18)
//   R2  u > 1+eps       <=  Nu' >  fl(k2*a)            k2 = 1+2^-20
//   R3  v < -eps        <=  Nv' < -fl(k1*a)
//   R4  u+v > 1+eps     <=  fl(Nu'+Nv') > fl(k3*a)     k3 = 1+2^-19   (given R1..R3 not certain)
//   R5  dist <= eps     <=  Nt' <  fl(k5*a)            k5 = eps*(1-2^-18)  (all triangles BEHIND
//                                                       the ray origin fall here)
//   R6  dist >  best    <=  Nt' >  fl(fl(best*a)*k2)   (strictly: a tie is never rejected here)
constexpr float kK1 = 1.1920928955078125e-07f * (1.f + 0x1p-18f);
constexpr float kK2 = 1.f + 0x1p-20f;
constexpr float kK3 = 1.f + 0x1p-19f;
constexpr float kK5 = 1.1920928955078125e-07f * (1.f - 0x1p-18f);

__device__ __forceinline__ uint32_t lane_prefix(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                     __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// true iff p holds in every ACTIVE lane: one v_cmp into an SGPR pair + scalar compare (hipcc's
// __all() goes through a v_cndmask/v_cmp_ne pair)
__device__ __forceinline__ bool wave_all(bool p) { return __builtin_amdgcn_ballot_w64(!p) == 0ull; }

__device__ __forceinline__ float xor_sign(float x, uint32_t sign_bit)
{
    return __uint_as_float(__float_as_uint(x) ^ sign_bit);
}

template <typename TriPtr>
__device__ __forceinline__ Hit closest_hit_staged(TriPtr tri, const uint32_t *__restrict__ orig,
                                                  uint32_t num_tri, F3 o, F3 d)
{
    float best = 1e9f;
    uint32_t who = HRT_NO_HIT, who_o = 0u;
    for (uint32_t j = 0; j < num_tri; ++j) {
        const float4 q0 = tri[HRT_ROW * j], q1 = tri[HRT_ROW * j + 1], q2 = tri[HRT_ROW * j + 2];
        const F3 v1 = {q0.x, q0.y, q0.z};
        const F3 e1 = {q0.w, q1.x, q1.y};
        const F3 e2 = {q1.z, q1.w, q2.x};
        const F3 pv = cross3(d, e2);
        const float det = dot3(e1, pv);
        const F3 s = sub3(o, v1);
        const float nu = dot3(s, pv);
        const float a = fabsf(det);
        const uint32_t sg = __float_as_uint(det) & 0x80000000u;
        const float nu_s = xor_sign(nu, sg);
        const float k1a = kK1 * a, k2a = kK2 * a;
        bool rej = (a < kEps) | (nu_s < -k1a) | (nu_s > k2a);
        if (wave_all(rej)) continue;
        const F3 q = cross3(s, e1);
        const float nv = dot3(d, q);
        const float nv_s = xor_sign(nv, sg);
        rej |= (nv_s < -k1a) | ((nu_s + nv_s) > kK3 * a);
        if (wave_all(rej)) continue;
        const float nt = dot3(e2, q);
        const float nt_s = xor_sign(nt, sg);
        rej |= (nt_s < kK5 * a) | (nt_s > (best * a) * kK2);
        if (wave_all(rej)) continue;
        // the reference's exact sequence (src/compute_paths.c:263-275) for the survivors
        const float u = nu / det;
        const float v = nv / det;
        const float w = u + v;
        const float dist = nt / det;
        const bool miss = (det > -kEps && det < kEps) | (u < -kEps) | (u > kOnePlusEps) |
                          (v < -kEps) | (w > kOnePlusEps);
        const uint32_t oj = orig[j];
        const bool take = !rej & !miss & (dist > kEps) & ((dist < best) | ((dist == best) & (oj < who_o)));
        best = take ? dist : best;
        who = take ? j : who;
        who_o = take ? oj : who_o;
    }
    return {who, best};
}

// Variant 2 ("packet"): the wave is treated as a RAY PACKET.  Per trace, wave reductions give a
// bounding ball of the 64 origins (centre oc, radius ro) and a bounding cone of the 64
// directions (axis ax, half-angle alpha); then the lanes swap roles -- lane l tests TRIANGLE
// 64k+l against the packet -- and a ballot yields the candidate triangles, which are walked in
// ascending index order through the staged test above.  A triangle is culled only when the
// reference's float test is PROVABLY rejecting for every ray of the packet.
//
// The proof works in numerator space, where the float error is an absolute bound with no
// conditioning blow-up (u = 2^-24, eps = 2^-23; E_* bound |float numerator - exact numerator|
// for any origin with |o - v1| <= S; derivation in DESIGN.md "packet culling"):
//     E_d = 16 eps |e1||e2|        E_u = 16 eps S |e2|       E_v = 16 eps S |e1|
//     E_t = 16 eps S |e1||e2|      (the worst-case analysis gives 10u, i.e. 3x below these)
// With sigma = sign(det_f) and a_f = |det_f|, acceptance by the reference requires
//     sigma*Nu >= -(2 eps (|N| + E_d) + E_u)                         (u >= -eps)
//     sigma*Nv >= -(2 eps (|N| + E_d) + E_v)                         (v >= -eps)
//     sigma*(Nu + Nv - det) <= 4 eps (|N| + E_d) + E_u + E_v + E_d   (u + v <= 1 + eps)
//     sigma*Nt > -E_t                                                (dist > eps)
// on the EXACT numerators Nu = d.(e2 x s), Nv = d.(s x e1), Nu+Nv-det = d.((e2-e1) x (s-e1)),
// Nt = s.N, det = -d.N (s = o - v1, N = e1 x e2).  Each is a linear form d.G(o) with G affine
// in o, so over the packet it is bounded by |G(oc)| * cos(angle(ax, G(oc)) -/+ alpha) plus
// |edge| * ro.  sigma is the same for all rays iff the cone does not straddle the plane
// (|N| (|ax.n| cos(alpha) - sin(alpha)) > 2 E_d); otherwise the triangle is kept.  Tolerances
// are doubled and a 1e-4 relative slack covers |d| != 1 and the roundoff of this test itself.
// Diagnostic counters (built only with -DHRT_KERNEL_STATS, `make STATS=1`): per kind of trace
// (0 primary of launch 0, 1 primary of later launches, 2 shadow): wave-traces, usable packets,
// candidate triangles, staged bodies that reached stage 2 / 3 / the exact divisions.
#if defined(HRT_KERNEL_STATS) || defined(HRT_PHASE_STATS)
__device__ unsigned long long g_stats[3][HRT_STATS_COLS];
#endif
#ifdef HRT_PHASE_STATS
__device__ unsigned long long g_phase[65536][8];
#endif
#ifdef HRT_UNIT_CLOCKS   // (make EXTRA=-DHRT_UNIT_CLOCKS) one record per wave-trace: start, duration | kind << 56 | usable << 60
__device__ unsigned long long g_unit[1u << 21][2];
__device__ unsigned int g_unit_n;
#endif
#ifdef HRT_KERNEL_STATS
#define HRT_STAT(kind, idx, val)                                                      \
    do {                                                                              \
        if (lane == 0) atomicAdd(&g_stats[kind][idx], (unsigned long long)(val));     \
    } while (0)
#else
#define HRT_STAT(kind, idx, val) do { } while (0)
#endif

// Wave-wide reductions on the VALU only (DPP row shifts + row broadcasts, the GCN/CDNA
// reduction idiom), written as inline assembly: ONE in-place instruction per step
// (`v_min_f32_dpp v, v, v row_shr:1`: lanes whose DPP source is out of range are not written and
// keep their own value, which is the identity of min/max/add-of-nothing), where the builtin
// route costs four (identity v_mov, v_mov_dpp, a NaN-canonicalising v_max, the operation) plus
// a hazard s_nop.  Several independent chains are interleaved step by step so that the two wait
// states a DPP read needs after the VALU write of its source are filled with work, not s_nops.
// The total ends in lane 63 and is read into an SGPR, i.e. wave-uniform.  All 64 lanes must be
// active (callers run in uniform control flow); inputs must not be NaN.  (The leading s_nop 4
// covers the worst hazard in front of a DPP instruction -- 5 wait states after a VALU write of
// EXEC -- since the compiler's hazard recogniser does not look inside inline assembly.)
__device__ __forceinline__ float lane63f(float v, uint32_t l)   // the value of lane l (wave-uniform l), in every lane
{
    return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), (int)l));
}
constexpr uint32_t kMaxSourceGroups = 6u;   // virtual sources of a wave traced as packets of their own (then: the rest as one)
__device__ __forceinline__ float lane63(float v)
{
    return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 63));
}

// component-wise minimum of lo and maximum of hi over the wave
__device__ __forceinline__ void wave_min3_max3(F3 &lo, F3 &hi)
{
    asm volatile(
        "s_nop 4\n"
        "v_min_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n"
        "v_min_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n"
        "v_min_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n"
        "v_max_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n"
        "v_max_f32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n"
        "v_max_f32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n"
        "v_min_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n"
        "v_min_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n"
        "v_min_f32_dpp %2, %2, %2 row_shr:2 row_mask:0xf bank_mask:0xf\n"
        "v_max_f32_dpp %3, %3, %3 row_shr:2 row_mask:0xf bank_mask:0xf\n"
        "v_max_f32_dpp %4, %4, %4 row_shr:2 row_mask:0xf bank_mask:0xf\n"
        "v_max_f32_dpp %5, %5, %5 row_shr:2 row_mask:0xf bank_mask:0xf\n"
        "v_min_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n"
        "v_min_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n"
        "v_min_f32_dpp %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf\n"
        "v_max_f32_dpp %3, %3, %3 row_shr:4 row_mask:0xf bank_mask:0xf\n"
        "v_max_f32_dpp %4, %4, %4 row_shr:4 row_mask:0xf bank_mask:0xf\n"
        "v_max_f32_dpp %5, %5, %5 row_shr:4 row_mask:0xf bank_mask:0xf\n"
        "v_min_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n"
        "v_min_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n"
        "v_min_f32_dpp %2, %2, %2 row_shr:8 row_mask:0xf bank_mask:0xf\n"
        "v_max_f32_dpp %3, %3, %3 row_shr:8 row_mask:0xf bank_mask:0xf\n"
        "v_max_f32_dpp %4, %4, %4 row_shr:8 row_mask:0xf bank_mask:0xf\n"
        "v_max_f32_dpp %5, %5, %5 row_shr:8 row_mask:0xf bank_mask:0xf\n"
        "v_min_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
        "v_min_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
        "v_min_f32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
        "v_max_f32_dpp %3, %3, %3 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
        "v_max_f32_dpp %4, %4, %4 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
        "v_max_f32_dpp %5, %5, %5 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
        "v_min_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
        "v_min_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
        "v_min_f32_dpp %2, %2, %2 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
        "v_max_f32_dpp %3, %3, %3 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
        "v_max_f32_dpp %4, %4, %4 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
        "v_max_f32_dpp %5, %5, %5 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
        : "+v"(lo.x), "+v"(lo.y), "+v"(lo.z), "+v"(hi.x), "+v"(hi.y), "+v"(hi.z));
    lo = {lane63(lo.x), lane63(lo.y), lane63(lo.z)};
    hi = {lane63(hi.x), lane63(hi.y), lane63(hi.z)};
}

__device__ __forceinline__ F3 wave_sum3(F3 v)
{
    asm volatile(
        "s_nop 4\n"
        "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %2, %2, %2 row_shr:2 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %2, %2, %2 row_shr:8 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
        "v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
        "v_add_f32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
        "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
        "v_add_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
        "v_add_f32_dpp %2, %2, %2 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
        : "+v"(v.x), "+v"(v.y), "+v"(v.z));
    return {lane63(v.x), lane63(v.y), lane63(v.z)};
}

// sum of a u32 over the wave, in every lane's SGPR-uniform result
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
    asm volatile(
        "s_nop 4\n"
        "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n"
        "v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n"
        "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n"
        "v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n"
        "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
        "s_nop 1\n"
        "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
        "s_nop 1\n"
        : "+v"(v));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ float wave_min_f(float v)
{
    asm volatile(
        "s_nop 4\n"
        "v_min_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n"
        "v_min_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n"
        "v_min_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n"
        "v_min_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n"
        "v_min_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
        "s_nop 1\n"
        "v_min_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
        "s_nop 1\n"
        : "+v"(v));
    return lane63(v);
}

// bitwise OR of a 64-bit value over the wave (two interleaved 32-bit chains), wave-uniform result
__device__ __forceinline__ unsigned long long wave_or64(unsigned long long v)
{
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    asm volatile(
        "s_nop 4\n"
        "v_or_b32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n"
        "v_or_b32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n"
        "s_nop 0\n"
        "v_or_b32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n"
        "v_or_b32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n"
        "s_nop 0\n"
        "v_or_b32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n"
        "v_or_b32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n"
        "s_nop 0\n"
        "v_or_b32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n"
        "v_or_b32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n"
        "s_nop 0\n"
        "v_or_b32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
        "v_or_b32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
        "s_nop 0\n"
        "v_or_b32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
        "v_or_b32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
        "s_nop 1\n"
        : "+v"(lo), "+v"(hi));
    const uint32_t ulo = (uint32_t)__builtin_amdgcn_readlane((int)lo, 63);
    const uint32_t uhi = (uint32_t)__builtin_amdgcn_readlane((int)hi, 63);
    return ((unsigned long long)uhi << 32) | (unsigned long long)ulo;
}

struct Ball { F3 c; float r; bool ok; };   // bounding ball of the packet's ray origins

struct Packet {
    F3 oc;      // a point every ray LINE of the packet passes within `ro` of
    float ro;
    F3 bc;      // bounding ball of the actual ray origins (behind test, |s| bound)
    float br;
    F3 ax;      // direction cone
    float cosa, sina;
    bool usable;
};

// The culling test needs conservative bounds, not exact values: it uses the single-instruction
// approximate v_sqrt_f32 / v_rsq_f32 (1 ulp) instead of the IEEE-correct expansions (~15
// instructions each); their error is far inside the 1e-4 relative slack added below.
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }

// The culling test is free to round differently from the reference (it only needs conservative
// bounds; everything that survives runs the reference's exact sequence), so unlike the rest of
// this file it uses fused multiply-adds: a dot product is 3 instructions instead of 5, a cross
// product 6 instead of 9 -- the kernel is VALU-issue-bound, so that is time.
__device__ __forceinline__ float fdot3(F3 a, F3 b)
{
    return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x));
}
__device__ __forceinline__ F3 fcross3(F3 a, F3 b)
{
    return {__builtin_fmaf(a.y, b.z, -(a.z * b.y)), __builtin_fmaf(a.z, b.x, -(a.x * b.z)),
            __builtin_fmaf(a.x, b.y, -(a.y * b.x))};
}

// must be called by ALL lanes of the wave (valid = lane carries a real ray)
__device__ __forceinline__ Ball origin_ball(F3 o, bool valid)
{
    const float big = 3.0e38f;
    F3 lo = {valid ? o.x : big, valid ? o.y : big, valid ? o.z : big};
    F3 hi = {valid ? o.x : -big, valid ? o.y : -big, valid ? o.z : -big};
    wave_min3_max3(lo, hi);
    const float lox = lo.x, loy = lo.y, loz = lo.z, hix = hi.x, hiy = hi.y, hiz = hi.z;
    Ball B;
    B.c = {0.5f * (lox + hix), 0.5f * (loy + hiy), 0.5f * (loz + hiz)};
    const F3 ext = {hix - lox, hiy - loy, hiz - loz};
    // half diagonal of the bounding box, rounded up
    B.r = 0.5f * fast_sqrt(fdot3(ext, ext)) * 1.0001f +
          1e-6f * (fabsf(B.c.x) + fabsf(B.c.y) + fabsf(B.c.z));
    B.ok = lox <= hix;
    return B;
}

// Direction cone of the packet.  `shadow`: every ray was aimed at the point `apex` (shadow
// rays towards an RX): numerators Nu, Nv, Nu+Nv-det and det are invariants of the ray LINE,
// and the line of a ray built as normalise(apex - o) passes within 4u*L of apex (componentwise
// rounding of the subtraction and the division), so the edge tests use the apex as the common
// line point with that tiny radius instead of the origins' bounding ball.
__device__ __forceinline__ Packet packet_bounds(const Ball &B, F3 d, bool valid, const bool shadow,
                                                F3 apex)
{
    Packet P;
    P.bc = B.c;
    P.br = B.r;
    // (bounds, not reference arithmetic: approximate rsq/sqrt and FMAs, like the culling test)
    F3 sd;
    if (shadow) {
        // every ray points at the apex: the direction from the ball's centre is as good an axis as
        // the mean direction, and costs no reduction (any unit axis is sound: the half-angle below is
        // measured against whichever is used)
        P.oc = apex;
        sd = sub3(apex, B.c);
        const float lmax = fast_sqrt(fdot3(sd, sd)) * 1.0001f + B.r;
        P.ro = 8.f * (0.5f * kEps) * lmax + 1e-7f;
    } else {
        P.oc = B.c;
        P.ro = B.r;
        sd = wave_sum3({valid ? d.x : 0.f, valid ? d.y : 0.f, valid ? d.z : 0.f});
    }
    const float n2 = fdot3(sd, sd);
    const float inv = fast_rsq(fmaxf(n2, 1e-30f));
    P.ax = {sd.x * inv, sd.y * inv, sd.z * inv};
    float c = wave_min_f(valid ? fdot3(d, P.ax) : 1.f);
    // |d| and |ax| are 1 within ~1e-6: widen the cone instead
    c = c * (1.f - 1e-5f) - 2e-6f;
    P.cosa = c;
    P.sina = fast_sqrt(fmaxf(0.f, __builtin_fmaf(-c, c, 1.f))) * 1.00001f + 1e-7f;
    // wide packets (half-angle > ~60 deg) or degenerate axis: culling cannot pay
    P.usable = (n2 > 1e-12f) && (c > 0.5f) && B.ok;
    return P;
}

// Bounds of d.G (hp) and of -d.G (hm) over the cone (|d| = 1), valid as UPPER bounds whenever they
// are negative -- which is all the culling test asks of them (it compares them with a negative
// threshold).  With beta the angle between the axis and G, the maximum of d.G over the cone is
// g cos(beta - alpha) for beta > alpha and g otherwise; in the second case the first expression
// is still positive (|beta - alpha| < 90 deg), so using it throughout never turns a "not culled"
// into a "culled" and saves the select and the square root for g.  The 1e-4 relative slack (for
// the roundoff of c1, s1 and the approximate v_sqrt) is taken on |c1| + s1 >= g.
__device__ __forceinline__ void cone_bounds(const Packet &P, F3 G, float &hp, float &hm)
{
    const float g2 = fdot3(G, G);
    const float c1 = fdot3(P.ax, G);                                       // g cos(beta)
    const float s1 = fast_sqrt(fmaxf(0.f, __builtin_fmaf(-c1, c1, g2)));   // g sin(beta)
    const float sl = fabsf(c1) + s1;
    const float t1 = c1 * P.cosa, t2 = s1 * P.sina;
    hp = __builtin_fmaf(1e-4f, sl, t2 + t1);
    hm = __builtin_fmaf(1e-4f, sl, t2 - t1);
}

// true iff triangle row (q0,q1,q2) with constants C = (-, E_d, 4 eps a_N, c_w) and lengths
// L = (|e1|, |e2|, |e2-e1|, |N|) is provably rejected by the reference's test for every ray of the
// packet.  The acceptance conditions are stated for sigma = sign(det_f), whatever it is: they are
// tested under BOTH hypotheses sigma = +1 and sigma = -1 with the bounds of the whole cone, and
// the triangle is culled when every hypothesis that is possible for some ray of the packet is
// rejected -- only one when the cone does not straddle the triangle's plane, both otherwise (so
// triangles seen edge-on by a wide packet are culled too, as long as they lie off to its side).
// Tolerances (doubled): tol_u = 2 (2 eps a_N + E_u) = C.z + 2 E_u,  a_N = 1.0001 |N| + E_d,
// tol_w = 2 (4 eps a_N + E_u + E_v + E_d) + 4e-6 |N| = C.w + 2 (E_u + E_v); the triangle-only
// parts come from the table (problem.c), the parts proportional to S are formed here.
__device__ __forceinline__ bool packet_culls(const Packet &P, float4 q0, float4 q1, float4 q2,
                                             float4 q3, float4 q4)
{
    // the row's last two float4, every word of the first and three of the second are used, so
    // that the per-lane gathers stay 128-/96-bit LDS reads (split into 32-bit reads at this
    // row stride they run into 4-way bank conflicts): E_d c_uv c_w |e1| , |e2| |e2-e1| |N| (mesh)
    const float4 C = make_float4(0.f, q3.x, q3.y, q3.z);
    const float4 L = make_float4(q3.w, q4.x, q4.y, q4.z);
    constexpr float kE = 16.f * kEps;
    const F3 v1 = {q0.x, q0.y, q0.z};
    const F3 e1 = {q0.w, q1.x, q1.y};
    const F3 e2 = {q1.z, q1.w, q2.x};
    const F3 nh = {q2.y, q2.z, q2.w};
    const F3 sb = sub3(P.bc, v1);
    // S >= |o - v1| for every origin: the 1-norm of sb bounds its length (two adds instead of a
    // dot product and a quarter-rate square root; S only scales tolerances of ~1e-6 relative)
    const float S = (fabsf(sb.x) + fabsf(sb.y)) + (fabsf(sb.z) + P.br);
    const F3 sc = sub3(P.oc, v1);
    const float dn = fdot3(P.ax, nh);
    // all rays on one side of the plane's direction field: sigma = sign(det) = sign(-d.N) is known
    const bool one_sided =
        L.w * __builtin_fmaf(fabsf(dn), P.cosa, -P.sina) > __builtin_fmaf(3.f, C.y, 1e-30f);
    const float k2S = (2.f * kE) * S;                 // 2 E_u = k2S |e2|, 2 E_v = k2S |e1|
    const float tol_u = __builtin_fmaf(k2S, L.y, C.z);
    const float tol_v = __builtin_fmaf(k2S, L.x, C.z);
    const float tol_w = __builtin_fmaf(k2S, L.x + L.y, C.w);
    // behind: sigma*Nt = sigma * (o - v1).N  with (o - v1).n in [h - br, h + br] (divided by |N|)
    const float h = fdot3(sb, nh);
    const float thr = __builtin_fmaf(-1e-4f * L.w, fabsf(h) + P.br, -(k2S * L.x) * L.y);   // -2 E_t
    bool rej_p = (P.br + h) * L.w < thr;
    bool rej_m = (P.br - h) * L.w < thr;
    // Nu = d.Gu, Nv = d.Gv, Nu + Nv - det = d.Gw (+- |edge| ro for the spread of the line points)
    const F3 Gu = fcross3(e2, sc);
    const F3 Gv = fcross3(sc, e1);
    // N = e1 x e2 is taken as n * |N| (table values, ~2e-7 relative): covered by 4e-6 |N| in tol_w
    const F3 Gw = {__builtin_fmaf(nh.x, L.w, Gu.x + Gv.x), __builtin_fmaf(nh.y, L.w, Gu.y + Gv.y),
                   __builtin_fmaf(nh.z, L.w, Gu.z + Gv.z)};
    const float ro = P.ro * 1.0001f;
    float hp, hm;
    // acceptance needs sigma*Nu >= -tol_u, sigma*Nv >= -tol_v, sigma*(Nu+Nv-det) <= tol_w
    cone_bounds(P, Gu, hp, hm);
    rej_p |= __builtin_fmaf(L.y, ro, hp) < -tol_u;
    rej_m |= __builtin_fmaf(L.y, ro, hm) < -tol_u;
    cone_bounds(P, Gv, hp, hm);
    rej_p |= __builtin_fmaf(L.x, ro, hp) < -tol_v;
    rej_m |= __builtin_fmaf(L.x, ro, hm) < -tol_v;
    cone_bounds(P, Gw, hp, hm);
    rej_p |= __builtin_fmaf(L.z, ro, hm) < -tol_w;
    rej_m |= __builtin_fmaf(L.z, ro, hp) < -tol_w;
    if (one_sided) return (dn < 0.f) ? rej_p : rej_m;
    return rej_p & rej_m;
}

// One staged test of triangle J for every lane of the packet walk.  The certain-reject state is
// kept as a 64-bit LANE MASK in SGPRs: each ballot of a single comparison is one v_cmp writing an
// SGPR pair, the ORs and the "every lane rejected?" test are scalar instructions (a ballot of an
// OR of comparisons costs two extra VALU instructions per stage).  All 64 lanes are active here;
// `inval` has the bits of the lanes that carry no ray.  Ties: lexicographic (distance, orig[J]).
#define HRT_BALLOT(c) __builtin_amdgcn_ballot_w64(c)
#define HRT_STAGED_BODY(J)                                                                      \
    {                                                                                           \
        const float4 q0 = tri[HRT_ROW * (J)], q1 = tri[HRT_ROW * (J) + 1], q2 = tri[HRT_ROW * (J) + 2];           \
        HRT_STAGED_TEST(J, q0, q1, q2)                                                          \
    }
#define HRT_STAGED_TEST(J, q0, q1, q2)                                                          \
    {                                                                                           \
        const F3 v1 = {q0.x, q0.y, q0.z};                                                       \
        const F3 e1 = {q0.w, q1.x, q1.y};                                                       \
        const F3 e2 = {q1.z, q1.w, q2.x};                                                       \
        const F3 pv = cross3(d, e2);                                                            \
        const float det = dot3(e1, pv);                                                         \
        const F3 s = sub3(o, v1);                                                               \
        const float nu = dot3(s, pv);                                                           \
        const float a = fabsf(det);                                                             \
        const uint32_t sg = __float_as_uint(det) & 0x80000000u;                                 \
        const float nu_s = xor_sign(nu, sg);                                                    \
        const float k1a = kK1 * a, k2a = kK2 * a;                                               \
        unsigned long long rm = inval | HRT_BALLOT(a < kEps) | HRT_BALLOT(nu_s < -k1a) |        \
                                HRT_BALLOT(nu_s > k2a);                                         \
        if (rm != ~0ull) {                                                                      \
            HRT_STAT(kind, 3, 1);                                                               \
            const F3 q = cross3(s, e1);                                                         \
            const float nv = dot3(d, q);                                                        \
            const float nv_s = xor_sign(nv, sg);                                                \
            rm |= HRT_BALLOT(nv_s < -k1a) | HRT_BALLOT((nu_s + nv_s) > kK3 * a);                \
            if (rm != ~0ull) {                                                                  \
                HRT_STAT(kind, 4, 1);                                                           \
                const float nt = dot3(e2, q);                                                   \
                const float nt_s = xor_sign(nt, sg);                                            \
                rm |= HRT_BALLOT(nt_s < kK5 * a) | HRT_BALLOT(nt_s > (best * a) * kK2);         \
                if (rm != ~0ull) {                                                              \
                    HRT_STAT(kind, 5, 1);                                                       \
                    const float u = nu / det;                                                   \
                    const float v = nv / det;                                                   \
                    const float w = u + v;                                                      \
                    const float dist = nt / det;                                                \
                    const bool rej = (rm >> lane) & 1ull;                                       \
                    const bool miss = (det > -kEps && det < kEps) | (u < -kEps) |               \
                                      (u > kOnePlusEps) | (v < -kEps) | (w > kOnePlusEps);      \
                    const uint32_t oj = orig[(J)];   /* J is wave-uniform: a scalar load */      \
                    const bool take = !rej & !miss & (dist > kEps) &                            \
                                      ((dist < best) | ((dist == best) & (oj < who_o)));        \
                    best = take ? dist : best;                                                  \
                    who = take ? (J) : who;                                                     \
                    who_o = take ? oj : who_o;                                                  \
                }                                                                               \
            }                                                                                   \
        }                                                                                       \
    }

// ALL lanes of the wave must call this (uniform control flow); invalid lanes carry dummies.
// Two phases per block of up to 16 rounds (1024 triangles): first every round is culled and
// its candidate mask parked in this wave's LDS slots, THEN the candidates are walked -- so the
// packet description and the culling temporaries are dead while the intersection tests run
// (and vice versa), which is what keeps the kernel's register allocation low.
constexpr uint32_t kMaskRounds = 16;
// Per-wave LDS scratch of the trace / fused kernels, in float4: 2 * kMaskRounds words-of-four for the
// walks (leaf constants, stacks, queues) and -- for the fine-leaves walk (variant 9), whose table is
// read from GLOBAL memory -- a buffer of kCandBuf candidate rows (3 float4 of the row + its index):
// the lane that finds a candidate in a culling round has the row in its registers and parks it
// here, so that the candidate walk reads LDS instead of paying a dependent L2 round trip per
// candidate.  (The 64-row leaf walk gains nothing from it: it is bound by the instructions of its
// ~350 culling rounds per trace, not by its candidates.)
constexpr uint32_t kCandBuf = 64u;
__host__ __device__ constexpr uint32_t wave_scratch4(bool cand_buf) { return 2u * kMaskRounds + (cand_buf ? 4u * kCandBuf : 0u); }

// cube-map cell of a direction (shared with the host builder, problem.c: rxt_cell_dir): face =
// major axis + 3 * (major < 0), (u, v) = the two other components over the (signed) major one
__device__ __forceinline__ uint32_t rxt_cell(F3 a)
{
    const float ax = fabsf(a.x), ay = fabsf(a.y), az = fabsf(a.z);
    uint32_t m = 0u;
    float major = a.x, c1 = a.y, c2 = a.z;
    if (ay > ax && ay >= az) { m = 1u; major = a.y; c1 = a.z; c2 = a.x; }
    else if (az > ax && az > ay) { m = 2u; major = a.z; c1 = a.x; c2 = a.y; }
    // (1 ulp: the host widens every cell's cone by 3e-4 rad for the rounding of this lookup)
    const float inv = __builtin_amdgcn_rcpf(major);
    const float u = c1 * inv, v = c2 * inv;
    int iu = (int)((u * 0.5f + 0.5f) * (float)HRT_RXT_N), iv = (int)((v * 0.5f + 0.5f) * (float)HRT_RXT_N);
    iu = iu < 0 ? 0 : (iu > HRT_RXT_N - 1 ? HRT_RXT_N - 1 : iu);
    iv = iv < 0 ? 0 : (iv > HRT_RXT_N - 1 ? HRT_RXT_N - 1 : iv);
    const uint32_t f = m + (major < 0.f ? 3u : 0u);
    return (f * HRT_RXT_N + (uint32_t)iv) * HRT_RXT_N + (uint32_t)iu;
}

// Tables of at most 64 triangles, traces bound to an APEX (launch rays leave a TX, shadow rays
// arrive at an RX): every lane looks up the candidate mask of the cube-map cell of ITS OWN direction
// in the apex's table (hrt_krxt.cell_mask: what the packet test cannot reject for any line through
// the apex with a direction in the cell, origins anywhere in the scene's ball -- built by the very
// same packet_culls, hrt_rxt_build_kernel, so its soundness is that of the packet test on a bigger
// packet), the wave ORs the masks, and the union is walked through the staged test.  No origin
// ball, no cone, no culling round: ~60 instead of ~330 instructions in front of the walk.
// `apex_k` is per lane (a wave of the launch set may straddle two TXs).  A lane whose origin is not
// inside the ball the tables were built for (cannot happen for hit points and TXs; NaNs) asks for
// every triangle.
template <typename TriPtr>
__device__ __forceinline__ Hit closest_hit_masked(TriPtr tri, const uint32_t *__restrict__ orig, const hrt_krxt &X,
                                                  uint32_t apex_k, uint32_t num_tri, F3 o, F3 d, bool valid,
                                                  uint32_t lane, [[maybe_unused]] int kind)
{
    float best = 1e9f;
    uint32_t who = HRT_NO_HIT, who_o = 0u;
    const unsigned long long inval = HRT_BALLOT(!valid);
    if (inval == ~0ull) return {who, best};
    unsigned long long mine = 0ull;
    if (valid) {
        const F3 dc = sub3(o, {X.cx, X.cy, X.cz});
        const bool inside = fast_sqrt(fdot3(dc, dc)) * 1.0001f <= X.region_r;   // NaN: false
        mine = inside ? X.cell_mask[(uint64_t)apex_k * HRT_RXT_BINS + rxt_cell(d)]
                      : (num_tri >= 64u ? ~0ull : ((1ull << num_tri) - 1ull));
    }
    unsigned long long m = wave_or64(mine);
    HRT_STAT(kind, 0, 1);
    HRT_STAT(kind, 1, 1);
    HRT_STAT(kind, 2, __popcll(m));
    while (m) {   // any order: ties go by (distance, original index)
        const uint32_t j = (uint32_t)__builtin_ctzll(m);
        m &= m - 1ull;
        HRT_STAGED_BODY(j)
    }
    return {who, best};
}

using Rsrc = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ Rsrc make_rsrc(const uint8_t *base)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base), 0, 0xffffffff, 0x00020000);
}

// bitwise OR of eight 32-bit words over the wave (eight interleaved in-place DPP chains: the chains fill
// each other's read-after-write wait states), wave-uniform results
__device__ __forceinline__ void wave_or256(uint32_t (&w)[8])
{
#define HRT_OR8(CTRL)                                                   \
        "v_or_b32_dpp %0, %0, %0 " CTRL "\n" "v_or_b32_dpp %1, %1, %1 " CTRL "\n"  \
        "v_or_b32_dpp %2, %2, %2 " CTRL "\n" "v_or_b32_dpp %3, %3, %3 " CTRL "\n"  \
        "v_or_b32_dpp %4, %4, %4 " CTRL "\n" "v_or_b32_dpp %5, %5, %5 " CTRL "\n"  \
        "v_or_b32_dpp %6, %6, %6 " CTRL "\n" "v_or_b32_dpp %7, %7, %7 " CTRL "\n"
    asm volatile(
        "s_nop 4\n"
        HRT_OR8("row_shr:1 row_mask:0xf bank_mask:0xf")
        HRT_OR8("row_shr:2 row_mask:0xf bank_mask:0xf")
        HRT_OR8("row_shr:4 row_mask:0xf bank_mask:0xf")
        HRT_OR8("row_shr:8 row_mask:0xf bank_mask:0xf")
        HRT_OR8("row_bcast:15 row_mask:0xa bank_mask:0xf")
        HRT_OR8("row_bcast:31 row_mask:0xc bank_mask:0xf")
        "s_nop 1\n"
        : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]));
#undef HRT_OR8
#pragma unroll
    for (int k = 0; k < 8; ++k) w[k] = (uint32_t)__builtin_amdgcn_readlane((int)w[k], 63);
}

// image of the point t in the plane of the triangle (v1 = q0.xyz, unit normal n = q2.yzw) -- the kernel
// that builds the patch tables and the lanes that look them up share this sequence bit for bit
__device__ __forceinline__ F3 image_of(F3 t, float4 q0, float4 q2)
{
    const F3 n = {q2.y, q2.z, q2.w};
    const float dn = fdot3(sub3(t, {q0.x, q0.y, q0.z}), n);
    return {__builtin_fmaf(-2.f * dn, n.x, t.x), __builtin_fmaf(-2.f * dn, n.y, t.y), __builtin_fmaf(-2.f * dn, n.z, t.z)};
}

// Patch tables (hrt_kpatch, hrt_kparams.h): every ray of a launch b >= 1 starts on the triangle it just
// hit (row `htri`).  The lane finds the cell of that triangle's grid its origin lies in -- and is SERVED
// only if the origin provably lies in the cell's ball: within hmax of the plane, at most HRT_PATCH_ACCEPT
// of a cell outside the grid (whatever history put it there: the test is on the origin itself) -- and for
// an image apex only if its line passes the apex ball, leaving it.  A served lane contributes the mask of
// (apex, patch), any other lane of the list the whole table; the wave walks the union through the staged
// test.  ALL lanes must call (uniform control flow).  apex_k: k (shadow rays to RX k) or num_rx + tx.
struct PatchRef { bool served; uint32_t off; };   // off: byte offset of the patch's masks inside one apex's table
template <typename TriPtr>
__device__ __forceinline__ PatchRef patch_locate(TriPtr tri, const hrt_kpatch &X, uint32_t num_tri, uint32_t htri, F3 o,
                                                 const bool image, F3 apex, F3 d)
{
    PatchRef R = {false, 0u};
    if (htri < num_tri) {
        const float4 q0 = tri[HRT_ROW * htri], q2 = tri[HRT_ROW * htri + 2];
        const float4 p0 = reinterpret_cast<const float4 *>(X.pdef)[2u * htri];
        const float4 p1 = reinterpret_cast<const float4 *>(X.pdef)[2u * htri + 1u];
        const F3 sv = sub3(o, {q0.x, q0.y, q0.z});
        const float fu = fdot3(sv, {p0.x, p0.y, p0.z}), fv = fdot3(sv, {p1.x, p1.y, p1.z});
        const float hh = fdot3(sv, {q2.y, q2.z, q2.w});
        const uint32_t bits = __float_as_uint(p1.w), nu = bits & 0xffffu, nv = bits >> 16;
        // (comparisons written so that a NaN is not served)
        R.served = (nu != 0u) & (fabsf(hh) <= X.hmax) & (fu >= -HRT_PATCH_ACCEPT) & (fv >= -HRT_PATCH_ACCEPT) &
                   (fu <= (float)nu + HRT_PATCH_ACCEPT) & (fv <= (float)nv + HRT_PATCH_ACCEPT);
        if (image) {
            // the apex of this lane: the image of its TX in ITS triangle's plane; the line must pass
            // within the radius the tables were built for, leaving the apex (|d| = 1 within 1e-6)
            const F3 im = image_of(apex, q0, q2);
            const F3 wv = sub3(o, im);
            const F3 cx = fcross3(wv, d);
            R.served &= (fdot3(cx, cx) <= 0.98f * X.ro_img * X.ro_img) && (fdot3(wv, d) > 0.f);
        }
        const uint32_t iu = min((uint32_t)max((int)floorf(fu), 0), nu - 1u);
        const uint32_t iv = min((uint32_t)max((int)floorf(fv), 0), nv - 1u);
        R.off = (__float_as_uint(p0.w) + iv * nu + iu) * (HRT_PATCH_WORDS * 8u);
    }
    return R;
}

// the mask words of a located lane for apex `apex_k` (a lane that is not served: the whole table; a lane past the
// end of the list: nothing) -- a request only: nothing waits for the loads here
__device__ __forceinline__ void patch_load(const hrt_kpatch &X, const PatchRef R, uint32_t apex_k, uint32_t num_tri, bool valid,
                                           uint32_t (&w)[8])
{
#pragma unroll
    for (int k = 0; k < 8; ++k) w[k] = 0u;
    if (valid) {
        if (R.served) {
            const Rsrc mr = make_rsrc(reinterpret_cast<const uint8_t *>(X.mask));
            const uint32_t off = R.off + apex_k * X.num_patch * (HRT_PATCH_WORDS * 8u);
            const auto a = __builtin_amdgcn_raw_buffer_load_b128(mr, (int)off, 0, 0);
            const auto b = __builtin_amdgcn_raw_buffer_load_b128(mr, (int)(off + 16u), 0, 0);
            w[0] = (uint32_t)a[0]; w[1] = (uint32_t)a[1]; w[2] = (uint32_t)a[2]; w[3] = (uint32_t)a[3];
            w[4] = (uint32_t)b[0]; w[5] = (uint32_t)b[1]; w[6] = (uint32_t)b[2]; w[7] = (uint32_t)b[3];
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                w[k] = num_tri >= 32u * (uint32_t)(k + 1) ? ~0u : (num_tri > 32u * (uint32_t)k ? (1u << (num_tri - 32u * (uint32_t)k)) - 1u : 0u);
        }
    }
}

// the trace of a wave from its lanes' mask words: the union over the wave, walked through the staged test
template <typename TriPtr, typename OrigPtr>
__device__ __forceinline__ Hit closest_hit_words(TriPtr tri, OrigPtr orig, uint32_t (&w)[8], uint32_t num_tri,
                                                 F3 o, F3 d, bool valid, uint32_t lane, [[maybe_unused]] int kind)
{
    float best = 1e9f;
    uint32_t who = HRT_NO_HIT, who_o = 0u;
    const unsigned long long inval = HRT_BALLOT(!valid);
    if (inval == ~0ull) return {who, best};
    wave_or256(w);
    HRT_STAT(kind, 0, 1);
    HRT_STAT(kind, 1, 1);
#pragma unroll
    for (uint32_t r = 0; r < HRT_PATCH_WORDS; ++r) {
        unsigned long long m = ((unsigned long long)w[2u * r + 1u] << 32) | (unsigned long long)w[2u * r];
        HRT_STAT(kind, 2, __popcll(m));
        while (m) {   // any order: ties go by (distance, original index)
            const uint32_t j = r * 64u + (uint32_t)__builtin_ctzll(m);
            m &= m - 1ull;
            HRT_STAGED_BODY(j)
        }
    }
    (void)num_tri;
    return {who, best};
}

// the trace of a wave whose lanes have located their patches (R; lanes past the end of the list: !valid)
template <typename TriPtr, typename OrigPtr>
__device__ __forceinline__ Hit closest_hit_patch(TriPtr tri, OrigPtr orig, const hrt_kpatch &X,
                                                 const PatchRef R, uint32_t apex_k, uint32_t num_tri, F3 o, F3 d,
                                                 bool valid, uint32_t lane, [[maybe_unused]] int kind)
{
    uint32_t w[8];
    patch_load(X, R, apex_k, num_tri, valid, w);
    HRT_STAT(kind, 6, (valid && R.served) ? 1 : 0);
    return closest_hit_words(tri, orig, w, num_tri, o, d, valid, lane, kind);
}

// launch 0 on patch-table problems: the ray leaves TX `tx` exactly (o == its position), so its candidates are the
// mask of the cube-map cell of its own direction (hrt_kpatch.txcell); the wave ORs and walks the union
template <typename TriPtr, typename OrigPtr>
__device__ __forceinline__ Hit closest_hit_txcell(TriPtr tri, OrigPtr orig, const hrt_kpatch &X, uint32_t tx, uint32_t num_tri,
                                                  F3 o, F3 d, bool valid, uint32_t lane)
{
    uint32_t w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) w[k] = 0u;
    if (valid) {
        const Rsrc mr = make_rsrc(reinterpret_cast<const uint8_t *>(X.txcell));
        const uint32_t off = (tx * HRT_RXT_BINS + rxt_cell(d)) * (HRT_PATCH_WORDS * 8u);
        const auto a = __builtin_amdgcn_raw_buffer_load_b128(mr, (int)off, 0, 0);
        const auto b = __builtin_amdgcn_raw_buffer_load_b128(mr, (int)(off + 16u), 0, 0);
        w[0] = (uint32_t)a[0]; w[1] = (uint32_t)a[1]; w[2] = (uint32_t)a[2]; w[3] = (uint32_t)a[3];
        w[4] = (uint32_t)b[0]; w[5] = (uint32_t)b[1]; w[6] = (uint32_t)b[2]; w[7] = (uint32_t)b[3];
    }
    return closest_hit_words(tri, orig, w, num_tri, o, d, valid, lane, 0);
}

template <bool MULTI, typename TriPtr>
__device__ __forceinline__ Hit closest_hit_packet(TriPtr tri, const uint32_t *__restrict__ orig,
                                                  const hrt_krxt &X, uint32_t rxk,
                                                  uint32_t num_tri, F3 o, F3 d,
                                                  bool valid, uint32_t lane, const Ball &B,
                                                  const bool shadow, F3 apex,
                                                  unsigned long long *wmask, int kind)
{
    float best = 1e9f;
    uint32_t who = HRT_NO_HIT, who_o = 0u;
    const unsigned long long inval = HRT_BALLOT(!valid);
    if (inval == ~0ull) return {who, best};   // a wave past the end of the live list
    // ITEMS are what the rounds walk: the rows 0 .. num_tri-1 of the table, or -- for a shadow packet
    // narrow enough for the RX's direction table -- the entries of its cell's list (row = list[item]).
    uint32_t n_items = num_tri;
    const uint16_t *list = nullptr;
    Packet P0 = packet_bounds(B, d, valid, shadow, apex);
    HRT_STAT(kind, 0, 1);
    HRT_STAT(kind, 1, P0.usable ? 1 : 0);
    // (a launch packet -- kind 0, rxk = num_rx + tx, apex = the TX -- leaves its apex as a shadow packet
    // arrives at its own: same tables; its lines pass within ro of the ball's centre, which is the TX
    // itself unless the wave straddles two TXs)
    if ((shadow || (kind == 0 && X.num_txt != 0u)) && X.enabled && P0.usable) {
        const F3 dc = sub3(B.c, {X.cx, X.cy, X.cz});
        const bool inside = fast_sqrt(fdot3(dc, dc)) * 1.0001f + B.r <= X.region_r;
        float ro_apex = P0.ro;
        if (!shadow) ro_apex += (fabsf(B.c.x - apex.x) + fabsf(B.c.y - apex.y)) + fabsf(B.c.z - apex.z);
        if (inside && P0.sina <= HRT_RXT_SIN_AQ && ro_apex <= X.ro_bin[rxk]) {
            const uint32_t cell = (uint32_t)__builtin_amdgcn_readfirstlane((int)rxt_cell(P0.ax));
            const uint32_t e0 = X.off[rxk * HRT_RXT_BINS + cell], e1 = X.off[rxk * HRT_RXT_BINS + cell + 1u];
            list = X.idx + e0;
            n_items = e1 - e0;
            HRT_STAT(kind, 6, 1);
            HRT_STAT(kind, 7, n_items);
        }
    }
    // The items are walked in blocks of kMaskRounds * 64 = 1024.  MULTI = false is the build for
    // scenes of at most one block (the host dispatches on num_tri): the loop below costs it nothing,
    // while the general build needs more VGPRs.  In the general build the packet description is
    // rebuilt per block (~100 instructions against the ~2000 of a block's culling) rather than kept
    // alive across the candidate walk.
    for (uint32_t blk0 = 0; blk0 < (MULTI ? n_items : 1u); blk0 += kMaskRounds * 64u) {
        const uint32_t blk1 = MULTI ? min(n_items, blk0 + kMaskRounds * 64u) : n_items;
        {
            const Packet P = (MULTI && blk0 != 0u) ? packet_bounds(B, d, valid, shadow, apex) : P0;
            if (!P.usable) {
                HRT_STAT(kind, 2, blk1 - blk0);
                for (uint32_t j = blk0; j < blk1; ++j) HRT_STAGED_BODY(j)
                continue;
            }
            for (uint32_t base = blk0, r = 0; base < blk1; base += 64u, ++r) {
                const uint32_t jl = base + lane;
                bool cand = false;
                if (jl < blk1) {
                    const uint32_t row = list ? (uint32_t)list[jl] : jl;
                    cand = !packet_culls(P, tri[HRT_ROW * row], tri[HRT_ROW * row + 1],
                                         tri[HRT_ROW * row + 2], tri[HRT_ROW * row + 3],
                                         tri[HRT_ROW * row + 4]);
                }
                const unsigned long long m = __builtin_amdgcn_ballot_w64(cand);
                HRT_STAT(kind, 2, __popcll(m));
                if (lane == 0) wmask[r] = m;
            }
        }
        for (uint32_t base = blk0, r = 0; base < blk1; base += 64u, ++r) {
            // written by this wave's lane 0 above, read back by all its lanes: same wave, in order
            unsigned long long m = wmask[r];
            // the rows of this round's items, one per lane: a candidate's row is then a v_readlane
            // away instead of a dependent load of list[item] in front of every staged test
            const uint32_t jl = base + lane;
            const uint32_t rowv = (list && jl < blk1) ? (uint32_t)list[jl] : jl;
            // (the builtin returns int: widen through uint32_t or the low word sign-extends)
            const uint32_t m_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)m);
            const uint32_t m_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(m >> 32));
            m = ((unsigned long long)m_hi << 32) | (unsigned long long)m_lo;
            while (m) {   // any order: ties go by (distance, original index)
                const uint32_t bit = (uint32_t)__builtin_ctzll(m);
                m &= m - 1ull;
                const uint32_t j = (uint32_t)__builtin_amdgcn_readlane((int)rowv, (int)bit);
                HRT_STAGED_BODY(j)
            }
        }
    }
    return {who, best};
}

// =====================================================================================
// Acceleration structure, leaf level (host side csrc/host/accel.c; the proofs
// are DESIGN.md section 9).  The table is in Morton order, so the 64 rows of a culling round -- a
// LEAF -- are neighbours, and a round can be dropped WHOLE when the packet passes the leaf's
// bounding sphere at a distance -- if, and that is the point of the guard below, none of its
// triangles is "doubly grazing" for the packet: a ray that lies (nearly) IN the plane of a triangle
// makes all numerators of the reference's test pure rounding noise, and the reference then reports
// hits on that triangle wherever it is in that plane, however far from the ray.  The per-triangle
// numerator-space test (packet_culls) keeps those; a distance test alone would drop them.
//
// Lemma.  Triangle with unit normal n, area A, longest edge l, p1 its first vertex;
// ray (o, d), |d| = 1, whose LINE misses the triangle by at least m > 0; S >= |o - p| for every
// point p of the triangle.  Then two of the three exact edge numerators Nu, Nv, det - Nu - Nv have
// opposite signs and magnitude at least
//     Dec = (A / l) max( |d.n| phi,  |n.(o - p1)| min(1/8, phi / (2 S)) ),   phi = m^2 / (2 m + l),
// and the reference rejects the triangle under both signs of det as soon as Dec exceeds the
// largest of packet_culls' (doubled) tolerances, which is below 1e-5 l (S + l).  With the safety
// factor SF = 4 the guard is:  max(...) > qs (S + l),  qs = SF 1e-5 l^2 / A (host, rounded up).
// Over a packet: |d.n| >= |ax.n| cos(alpha) - sin(alpha); |n.(o - p1)| >= |n.(bc - p1)| - br; the
// distance from any of the packet's lines to a point c is at least
// r_perp cos(alpha) - |t| sin(alpha) - ro  (t, r_perp: the coordinates of c - oc along / across ax).
// A triangle that fails the guard is not lost, it just goes through packet_culls like today.
// =====================================================================================
struct LeafFar { float m, phi, kappa, S; };

// lane-parallel over LEAVES: is sphere (c, R) with longest edge lam missed by every line of the
// packet, and with which constants.  NaN / infinite inputs give m <= 0 or phi = 0: never "safe".
__device__ __forceinline__ LeafFar leaf_far(const Packet &P, float4 n0, float4 n1)
{
    const F3 c = {n0.x, n0.y, n0.z};
    const float R = n0.w, lam = n1.x;
    const F3 w = sub3(c, P.oc);
    const float t = fdot3(w, P.ax);
    const F3 x = fcross3(w, P.ax);                    // |w x ax| = r_perp, without cancellation
    const float rp = fast_sqrt(fdot3(x, x));
    const float w1 = (fabsf(w.x) + fabsf(w.y)) + fabsf(w.z);
    LeafFar f;
    // (slack: the packet's cone and radii are already widened; 2e-5 (|w| + R) covers this line)
    f.m = __builtin_fmaf(rp, P.cosa, -fabsf(t) * P.sina) - P.ro * 1.0001f - R - 2e-5f * (w1 + R);
    const F3 sb = sub3(c, P.bc);
    f.S = ((fabsf(sb.x) + fabsf(sb.y)) + fabsf(sb.z)) + R + P.br;          // >= |o - p|, 1-norm
    const float mm = fmaxf(f.m, 0.f);
    f.phi = mm * mm * __builtin_amdgcn_rcpf(__builtin_fmaf(2.f, mm, lam)) * 0.9999f;
    f.kappa = fminf(0.125f, 0.5f * f.phi * __builtin_amdgcn_rcpf(f.S) * 0.9999f);
    return f;
}

// lane-parallel over the TRIANGLES of a far leaf: true = provably rejected for every ray of the
// packet (the guard of the lemma); false = must go through packet_culls.
__device__ __forceinline__ bool guard_safe(const Packet &P, float4 q0, float4 q2, float2 g, float phi,
                                           float kappa, float S)
{
    const F3 p1 = {q0.x, q0.y, q0.z};
    const F3 n = {q2.y, q2.z, q2.w};
    const float dn = fdot3(P.ax, n);
    const float amin = __builtin_fmaf(fabsf(dn), P.cosa, -P.sina);
    const F3 sb = sub3(P.bc, p1);
    const float hc = fdot3(n, sb);
    const float hmin = fabsf(hc) - P.br * 1.0001f - 2e-6f * ((fabsf(sb.x) + fabsf(sb.y)) + fabsf(sb.z));
    const float Ti = g.x * (S + g.y);
    // written so that NaNs (degenerate triangles: qs = inf, n = NaN) compare false: not safe
    return (amin * phi > Ti) | (hmin * kappa > Ti);
}

// ALL lanes of the wave must call this (uniform control flow); invalid lanes carry dummies.
// The table is walked in blocks of kMaskRounds leaves: packet bounds, one round over the block's
// leaves (far or not), then per leaf either the guard (far) or packet_culls (near, and whatever
// the guard could not clear), masks parked in LDS, then the candidate walk -- as closest_hit_packet.
template <bool MULTI, typename TriPtr, typename TgPtr, typename LeafPtr>
__device__ __forceinline__ Hit closest_hit_tree(TriPtr tri, TgPtr tg, LeafPtr leaf,
                                                const uint32_t *__restrict__ orig, uint32_t num_tri,
                                                F3 o, F3 d, bool valid, uint32_t lane, const Ball &B,
                                                const bool shadow, F3 apex,
                                                unsigned long long *wmask, float4 *wleaf,
                                                [[maybe_unused]] int kind)
{
    float best = 1e9f;
    uint32_t who = HRT_NO_HIT, who_o = 0u;
    const unsigned long long inval = HRT_BALLOT(!valid);
    if (inval == ~0ull) return {who, best};   // a wave past the end of the live list
    for (uint32_t blk0 = 0; blk0 < (MULTI ? num_tri : 1u); blk0 += kMaskRounds * 64u) {
        const uint32_t blk1 = MULTI ? min(num_tri, blk0 + kMaskRounds * 64u) : num_tri;
        {
            const Packet P = packet_bounds(B, d, valid, shadow, apex);
            if (blk0 == 0) {
                HRT_STAT(kind, 0, 1);
                HRT_STAT(kind, 1, P.usable ? 1 : 0);
            }
            if (!P.usable) {
                HRT_STAT(kind, 2, blk1 - blk0);
                for (uint32_t j = blk0; j < blk1; ++j) HRT_STAGED_BODY(j)
                continue;
            }
            // ---- one round over the leaves of this block: lane l looks at leaf blk0/64 + l ----
            unsigned long long farm;
            {
                const uint32_t lf = (blk0 >> 6) + lane;
                const bool has = lane < kMaskRounds && lf * 64u < blk1;
                const uint32_t lfc = has ? lf : (blk0 >> 6);
                const LeafFar f = leaf_far(P, leaf[2u * lfc], leaf[2u * lfc + 1u]);
                farm = HRT_BALLOT(has && f.m > 0.f);
                if (lane < kMaskRounds) wleaf[lane] = make_float4(f.phi, f.kappa, f.S, 0.f);
            }
#ifdef HRT_KERNEL_STATS
            uint32_t ctot = 0;
#endif
            for (uint32_t base = blk0, r = 0; base < blk1; base += 64u, ++r) {
                const uint32_t jl = base + lane;
                bool cand = jl < blk1;
                if ((farm >> r) & 1ull) {   // wave-uniform
                    const float4 lc = wleaf[r];   // same address in every lane: a broadcast read
                    HRT_STAT(kind, 6, 1);
                    if (cand)
                        cand = !guard_safe(P, tri[HRT_ROW * jl], tri[HRT_ROW * jl + 2], tg[jl], lc.x, lc.y, lc.z);
                    if (HRT_BALLOT(cand) == 0ull) {
                        HRT_STAT(kind, 7, 1);
                        if (lane == 0) wmask[r] = 0ull;
                        continue;
                    }
                }
                if (cand)
                    cand = !packet_culls(P, tri[HRT_ROW * jl], tri[HRT_ROW * jl + 1],
                                         tri[HRT_ROW * jl + 2], tri[HRT_ROW * jl + 3],
                                         tri[HRT_ROW * jl + 4]);
                const unsigned long long m = __builtin_amdgcn_ballot_w64(cand);
                HRT_STAT(kind, 2, __popcll(m));
                if (lane == 0) wmask[r] = m;
            }
        }
        for (uint32_t base = blk0, r = 0; base < blk1; base += 64u, ++r) {
            unsigned long long m = wmask[r];
            const uint32_t m_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)m);
            const uint32_t m_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(m >> 32));
            m = ((unsigned long long)m_hi << 32) | (unsigned long long)m_lo;
            while (m) {
                const uint32_t j = base + (uint32_t)__builtin_ctzll(m);
                m &= m - 1ull;
                HRT_STAGED_BODY(j)
            }
        }
    }
    return {who, best};
}

// =====================================================================================
// Acceleration structure, big tables (more than HRT_ACCEL_BIG triangles): 64-ary levels of
// bounding spheres over the leaves, the plane tree for the guard, and splitting of packets that are
// too wide to cull.
//
//   * A sphere node (any level) is dropped when every line of the packet misses it by
//     m >= max(Lambda / 2, mu S): with that margin a triangle below it whose plane is NOT nearly
//     parallel to the rays -- |d.n| > Gamma_i = (12 / mu) qs_i for every ray -- is rejected by the
//     reference whatever else holds (lemma above, first term: l <= 2 m gives phi >= m / 4, and
//     m <= S gives S + l <= 3 S).
//   * The triangles with |d.n| <= Gamma_i for some ray of the packet are found through the PLANE
//     TREE (triangles sorted by normal direction; a node is a cone of normals and is skipped when
//     the packet's directions stay clear of all its planes), and each is then judged on its own:
//     its ball (p1, l) against the packet's lines, and the lemma with that margin.  Whatever is not
//     cleared goes through packet_culls and the staged test like the triangles of near leaves (a
//     triangle reached both ways is tested twice: same result).
//   * A packet too wide to cull (half-angle > 60 deg: the rays of the wave bounced off different
//     surfaces) takes one staged pass over the whole table, as in the flat walk.
// All control flow is wave-uniform; per-wave scratch (node masks, queue of candidate masks, range
// stack) lives in LDS.
// =====================================================================================
constexpr float kMu = (float)HRT_GUARD_MU;
constexpr float kGammaPerQs = (float)(12.0 / HRT_GUARD_MU);

template <typename TriPtr, typename LeafPtr>
__device__ __forceinline__ Hit closest_hit_big(TriPtr tri, LeafPtr leaf, const hrt_kaccel &A,
                                               uint32_t num_tri, F3 o, F3 d, bool valid, uint32_t lane,
                                               const bool shadow, F3 apex, unsigned long long *wmask,
                                               uint32_t *ws, [[maybe_unused]] int kind)
{
    const uint32_t *__restrict__ orig = A.orig;
    float best = 1e9f;
    uint32_t who = HRT_NO_HIT, who_o = 0u;
    if (HRT_BALLOT(valid) == 0ull) return {who, best};
    // per-wave scratch words: [0..15] queue tags, [16..23] node masks of up to 4 levels (u64),
    // [24..27] their bases, [32..47] lane-range stack
    unsigned long long *nmask = reinterpret_cast<unsigned long long *>(ws + 16);
    uint32_t *nbase = ws + 24, *rstack = ws + 32;
    auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    auto uni64 = [&](unsigned long long v) {
        return ((unsigned long long)uni((uint32_t)(v >> 32)) << 32) | (unsigned long long)uni((uint32_t)v);
    };
    // A packet too wide to cull (the rays of the wave scattered off different surfaces) takes ONE
    // staged pass over the whole table at the end -- like the flat walk.  Cutting such a wave into
    // lane ranges, down to single rays, and walking the tree for each was measured and lost at every
    // size tried (city, T = 25 002: 192 ms with ranges down to single rays, 62 ms down to 8 lanes,
    // 31 ms with the one pass): a tree walk is a chain of dependent loads, the staged pass streams.
    // HRT_ACCEL_DEBUG bit 3 (8) re-enables the cutting for experiments.
    const uint32_t min_range = 64u;
    unsigned long long unresolved = 0ull;
    uint32_t rs = 0;
    if (lane == 0) rstack[0] = 0u | (64u << 8);
    rs = 1;
    while (rs > 0) {
        --rs;
        const uint32_t rg = uni(rstack[rs]);
        const uint32_t lo = rg & 0xffu, hi = rg >> 8;
        const bool sub = valid && lane >= lo && lane < hi;
        const unsigned long long inval = HRT_BALLOT(!sub);
        if (inval == ~0ull) continue;
        const Ball B = origin_ball(o, sub);
        const Packet P = packet_bounds(B, d, sub, shadow, apex);
        if (lo == 0 && hi == 64) {
            HRT_STAT(kind, 0, 1);
            HRT_STAT(kind, 1, P.usable ? 1 : 0);
        }
        if (!P.usable) {
            if (hi - lo > min_range) {
                const uint32_t mid = (lo + hi) >> 1;
                if (lane == 0) {
                    rstack[rs] = mid | (hi << 8);
                    rstack[rs + 1] = lo | (mid << 8);
                }
                rs += 2;
            } else {
                unresolved |= ~inval;
            }
            continue;
        }
        HRT_STAT(kind, 6, 1);
        uint32_t qn = 0;
        auto flush = [&]() {
            for (uint32_t e = 0; e < qn; ++e) {
                const uint32_t tag = uni(ws[e]);
                unsigned long long m = uni64(wmask[e]);
                while (m) {
                    const uint32_t b = (uint32_t)__builtin_ctzll(m);
                    m &= m - 1ull;
                    const uint32_t j = (tag & 0x80000000u) ? A.pl_index[(tag & 0x7fffffffu) * 64u + b]
                                                           : tag * 64u + b;
                    HRT_STAGED_BODY(j)
                }
            }
            qn = 0;
        };
        auto enqueue = [&](uint32_t tag, unsigned long long m) {
            if (m == 0ull) return;
            if (lane == 0) { ws[qn] = tag; wmask[qn] = m; }
            if (++qn == kMaskRounds) flush();
        };
        // ---- sphere levels: level L = top (<= 64 nodes) ... level 0 = leaves ----
        {
            const uint32_t L = A.num_levels;
            auto visit = [&](uint32_t k, uint32_t base) -> unsigned long long {
                const uint32_t count = (k == 0u) ? A.num_leaf : (k == 1u ? A.node_count[0] : (k == 2u ? A.node_count[1] : A.node_count[2]));
                const uint32_t idx = base + lane;
                const bool has = idx < count;
                const uint32_t ic = has ? idx : base;
                float4 n0, n1;
                if (k == 0u) { n0 = leaf[2u * ic]; n1 = leaf[2u * ic + 1u]; }
                else {
                    const float4 *arr = reinterpret_cast<const float4 *>(k == 1u ? A.node[0] : (k == 2u ? A.node[1] : A.node[2]));
                    n0 = arr[2u * ic]; n1 = arr[2u * ic + 1u];
                }
                const LeafFar f = leaf_far(P, n0, n1);
                const bool far = f.m > fmaxf(0.5f * n1.x, kMu * f.S);   // NaN / inf: not far
                HRT_STAT(kind, 8, 1);
                return HRT_BALLOT(has && !far);
            };
            uint32_t k = L;
            {
                const unsigned long long m0 = visit(k, 0u);
                if (lane == 0) { nmask[k] = m0; nbase[k] = 0u; }
            }
            for (;;) {
                const unsigned long long m = uni64(nmask[k]);
                if (m == 0ull) {
                    if (k == L) break;
                    ++k;
                    continue;
                }
                const uint32_t b = (uint32_t)__builtin_ctzll(m);
                const uint32_t idx = uni(nbase[k]) + b;
                if (lane == 0) nmask[k] = m & (m - 1ull);
                if (k == 0u) {   // a near leaf: packet culling over its 64 rows
                    const uint32_t jl = idx * 64u + lane;
                    bool cand = jl < num_tri;
                    if (cand)
                        cand = !packet_culls(P, tri[HRT_ROW * jl], tri[HRT_ROW * jl + 1],
                                             tri[HRT_ROW * jl + 2], tri[HRT_ROW * jl + 3],
                                             tri[HRT_ROW * jl + 4]);
                    const unsigned long long cm = HRT_BALLOT(cand);
                    HRT_STAT(kind, 2, __popcll(cm));
                    HRT_STAT(kind, 7, 1);
                    enqueue(idx, cm);
                } else {
                    --k;
                    const unsigned long long mk = visit(k, idx * 64u);
                    if (lane == 0) { nmask[k] = mk; nbase[k] = idx * 64u; }
                }
            }
        }
        // ---- plane tree: level pl_levels-1 = top ... level 0 = cones of the 64-entry leaves ----
        {
            const uint32_t L = A.pl_levels - 1u;
            auto visit = [&](uint32_t k, uint32_t base) -> unsigned long long {
                const uint32_t count = k == 0u ? A.pl_count[0] : (k == 1u ? A.pl_count[1] : A.pl_count[2]);
                const float4 *arr = reinterpret_cast<const float4 *>(k == 0u ? A.pl_node[0] : (k == 1u ? A.pl_node[1] : A.pl_node[2]));
                const uint32_t idx = base + lane;
                const bool has = idx < count;
                const uint32_t ic = has ? idx : base;
                const float4 n0 = arr[2u * ic], n1 = arr[2u * ic + 1u];
                const float sdot = fabsf(fdot3(P.ax, {n0.x, n0.y, n0.z}));
                // every member plane is clear of the packet's directions by more than its Gamma:
                // elevation of ax over the planes > alpha + beta + g
                const float thr = __builtin_fmaf(P.sina, n1.x, P.cosa * n0.w) * 1.0001f + 1e-6f;
                const bool skip = sdot > thr;                    // NaN: visit
                HRT_STAT(kind, 9, 1);
                return HRT_BALLOT(has && !skip);
            };
            uint32_t k = L;
            {
                const unsigned long long m0 = visit(k, 0u);
                if (lane == 0) { nmask[k] = m0; nbase[k] = 0u; }
            }
            for (;;) {
                const unsigned long long m = uni64(nmask[k]);
                if (m == 0ull) {
                    if (k == L) break;
                    ++k;
                    continue;
                }
                const uint32_t b = (uint32_t)__builtin_ctzll(m);
                const uint32_t idx = uni(nbase[k]) + b;
                if (lane == 0) nmask[k] = m & (m - 1ull);
                if (k == 0u) {   // a leaf of 64 triangle ids: each judged on its own
                    const uint32_t e = idx * 64u + lane;
                    const uint32_t j = A.pl_index[e];
                    const float4 *rec = reinterpret_cast<const float4 *>(A.pl_rec) + 2u * e;
                    const float4 r0 = rec[0], r1 = rec[1];
                    bool cand = j != HRT_NO_HIT;
                    HRT_STAT(kind, 10, 1);
                    if (cand) {
                        const F3 n = {r1.x, r1.y, r1.z};
                        const float amin = __builtin_fmaf(fabsf(fdot3(P.ax, n)), P.cosa, -P.sina);
                        bool safe = amin > kGammaPerQs * r1.w;           // the spheres vouch for it
                        if (!safe) {
                            const LeafFar f = leaf_far(P, r0, make_float4(r0.w, 0.f, 0.f, 0.f));   // ball (p1, l)
                            const F3 sb = sub3(P.bc, {r0.x, r0.y, r0.z});
                            const float hmin = fabsf(fdot3(n, sb)) - P.br * 1.0001f -
                                               2e-6f * ((fabsf(sb.x) + fabsf(sb.y)) + fabsf(sb.z));
                            const float Ti = r1.w * (f.S + r0.w);
                            safe = (f.m > 0.f) & ((amin * f.phi > Ti) | (hmin * f.kappa > Ti));
                        }
                        cand = !safe;
                    }
                    if (HRT_BALLOT(cand) != 0ull) {
                        HRT_STAT(kind, 11, 1);
                        HRT_STAT(kind, 12, __popcll(HRT_BALLOT(cand)));
                        if (cand) {
                            const uint32_t jc = j;
                            cand = !packet_culls(P, tri[HRT_ROW * jc], tri[HRT_ROW * jc + 1],
                                                 tri[HRT_ROW * jc + 2], tri[HRT_ROW * jc + 3],
                                                 tri[HRT_ROW * jc + 4]);
                        }
                        enqueue(idx | 0x80000000u, HRT_BALLOT(cand));
                    }
                } else {
                    --k;
                    const unsigned long long mk = visit(k, idx * 64u);
                    if (lane == 0) { nmask[k] = mk; nbase[k] = idx * 64u; }
                }
            }
        }
        flush();
    }
    if (unresolved != 0ull) {   // every triangle, exactly (also: single rays with a NaN or zero direction)
        const unsigned long long inval = ~unresolved;
        for (uint32_t j = 0; j < num_tri; ++j) HRT_STAGED_BODY(j)
    }
    return {who, best};
}

// The queue of packets that are too wide to cull (closest_hit_fine pushes, hrt_wide_kernel /
// hrt_wide_finish_kernel consume): an entry is (chunk << 32 | trace kind << 2 | wave of the chunk), its 64
// keys are the per-ray minima of (float bits of the distance) << 32 | original triangle index.
constexpr uint32_t HRT_DEFERRED = 0xfffffffeu;   // Hit.tri of a queued packet (wave-uniform)
struct WideQ {
    uint32_t *cnt = nullptr;              // entries pushed in this launch (may run past cap: those ran inline)
    unsigned long long *q = nullptr;
    unsigned long long *keys = nullptr;
    uint32_t cap = 0u;
    float cos_min = 0.f;                  // packets with a wider cone are queued
    unsigned long long entry = 0ull;
};

// =====================================================================================
// Acceleration structure, FINE LEAVES (variant 9: the default walk of tables of more than
// HRT_FINE_MIN_TRI triangles without the big-table trees).  Measured on a city of 25 000 triangles:
// after the re-sort 99 % of the packets are usable, but 354 of the 391 leaves of 64 rows are NEAR a
// packet's lines (a leaf sphere there has a radius of 60 m) and go through a full packet_culls round
// each -- 51 000 of the 70 000 instructions of a trace.  Here the spheres are those of 16 consecutive
// rows (the order is a k-d order: any run of rows is a compact cell), scanned FLAT, 64 spheres per
// round -- no tree, no dependent loads -- with the big tables' criterion: a sphere is dropped when every
// line of the packet misses it by m >= max(Lambda / 2, mu S); the triangles of dropped spheres that
// are nearly parallel to the rays (|d.n| <= Gamma_i for some ray: the only ones the lemma's first term
// does not reject) are found through the PLANE TREE and judged one by one, exactly as in
// closest_hit_big.  The rows of four near spheres fill one packet_culls round (28-43 rounds per trace
// instead of 354), and a candidate's row is parked in LDS by the lane that culled it (wave_scratch4):
// with the candidates read back from the table the walk was bound by one L2 round trip per candidate
// and SLOWER than the 64-row leaves (27.8 against 18.7 ms per step); with the buffer 15.6 (100 002
// triangles: 88.6 -> 67.5).  Soundness: DESIGN_ACCEL.md B.2 / B.4 (the lemma does not care how many
// triangles a sphere holds).
// Per-wave LDS scratch `ws`: words [16..23] node masks of the plane levels, [24..27] their bases; behind
// the 128 words the buffer of kCandBuf candidate rows (wave_scratch4).
// =====================================================================================
template <typename TriPtr>
__device__ __forceinline__ Hit closest_hit_fine(TriPtr tri, const hrt_kaccel &A, uint32_t num_tri, F3 o, F3 d,
                                                bool valid, uint32_t lane, const Ball &B, const bool shadow, F3 apex,
                                                unsigned long long *wmask, uint32_t *ws, [[maybe_unused]] int kind,
                                                const WideQ &wq)
{
    const uint32_t *__restrict__ orig = A.orig;
    float best = 1e9f;
    uint32_t who = HRT_NO_HIT, who_o = 0u;
    const unsigned long long inval = HRT_BALLOT(!valid);
    if (inval == ~0ull) return {who, best};
    const Packet P0 = packet_bounds(B, d, valid, shadow, apex);
    HRT_STAT(kind, 0, 1);
    HRT_STAT(kind, 1, P0.usable ? 1 : 0);
    unsigned long long *nmask = reinterpret_cast<unsigned long long *>(ws + 16);
    uint32_t *nbase = ws + 24;
    auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    auto uni64 = [&](unsigned long long v) {
        return ((unsigned long long)uni((uint32_t)(v >> 32)) << 32) | (unsigned long long)uni((uint32_t)v);
    };
    // candidates: their rows are parked in the wave's LDS buffer by the lane that just culled them (it has
    // the row in registers) and walked from there -- no load from the table in the candidate walk
    float4 *cbuf = reinterpret_cast<float4 *>(ws) + 2u * kMaskRounds;
    uint32_t nbuf = 0u;
    auto flush = [&]() {
        for (uint32_t e = 0; e < nbuf; ++e) {
            const float4 *slot = cbuf + 4u * e;
            const float4 c0 = slot[0], c1 = slot[1], c2 = slot[2];
            const uint32_t j = uni(__float_as_uint(slot[3].x));
            HRT_STAGED_TEST(j, c0, c1, c2)
        }
        nbuf = 0u;
    };
    auto park = [&](bool cand, uint32_t j, float4 c0, float4 c1, float4 c2) {   // all lanes call (uniform)
        const unsigned long long cm = HRT_BALLOT(cand);
        const uint32_t n = (uint32_t)__popcll(cm);
        if (n == 0u) return;
        if (nbuf + n > kCandBuf) flush();
        if (cand) {
            float4 *slot = cbuf + 4u * (nbuf + lane_prefix(cm));
            slot[0] = c0; slot[1] = c1; slot[2] = c2;
            slot[3] = make_float4(__uint_as_float(j), 0.f, 0.f, 0.f);
        }
        nbuf += n;
    };
    // A packet too wide to cull (the rays of the wave scattered off different surfaces; 0.1-0.3 % of the
    // packets after the re-sort) owes every triangle an exact test: a serial chain of T staged tests in
    // ONE wave (4 ms for 25 000 triangles, 16 ms for 100 000) -- few as they are, those passes were the
    // TAIL that set the kernel's time (0.9 resident waves per SIMD on average, profiles/unit_clocks.py).
    // The wave therefore only QUEUES the packet (hrt_wide): hrt_wide_kernel spreads (packet, slice of the
    // table) items over the whole chip and merges the per-ray minima of (distance, original index) with
    // 64-bit atomic minima, hrt_wide_finish_kernel writes the results.  Only when the queue is full does the
    // wave run the pass itself.
    const Packet &P = P0;
#ifdef HRT_UNIT_CLOCKS
    if (lane == 0) { ws[15] = P0.usable ? 1u : 0u; ws[14] = 0u; ws[13] = (uint32_t)(fmaxf(P0.cosa, 0.f) * 255.f); ws[12] = 0u; }
#endif
    const bool wide = !P0.usable || !(P0.cosa >= wq.cos_min);
    if (wide) {
        uint32_t slot = 0xffffffffu;
        if (wq.cnt != nullptr) {
            if (lane == 0) slot = atomicAdd(wq.cnt, 1u);
            slot = uni(slot);
        }
        if (slot < wq.cap) {
            if (lane == 0) wq.q[slot] = wq.entry;
            wq.keys[(uint64_t)slot * 64u + lane] = ~0ull;
            return {HRT_DEFERRED, best};
        }
    }
    if (!P0.usable) {
        // (queue full) every triangle, exactly: 64 rows at a time are fetched by the 64 lanes into the wave's
        // LDS buffer and tested from there (one L2 round trip per 64 triangles)
        HRT_STAT(kind, 2, num_tri);
        for (uint32_t base = 0; base < num_tri; base += 64u) {
            const uint32_t jl = base + lane;
            if (jl < num_tri) {
                float4 *slot4 = cbuf + 4u * lane;
                slot4[0] = tri[HRT_ROW * jl]; slot4[1] = tri[HRT_ROW * jl + 1]; slot4[2] = tri[HRT_ROW * jl + 2];
            }
            const uint32_t n = min(64u, num_tri - base);
            for (uint32_t e = 0; e < n; ++e) {
                const float4 *slot4 = cbuf + 4u * e;
                const float4 c0 = slot4[0], c1 = slot4[1], c2 = slot4[2];
                const uint32_t j = base + e;
                HRT_STAGED_TEST(j, c0, c1, c2)
            }
        }
        return {who, best};
    }
    // ---- fine leaves, flat: lane l looks at sphere base + l ----
    {
        const float4 *fine = reinterpret_cast<const float4 *>(A.fine);
        const uint32_t nfine = A.num_fine;
        for (uint32_t base = 0; base < nfine; base += 64u) {
            const uint32_t idx = base + lane;
            const bool has = idx < nfine;
            const uint32_t ic = has ? idx : base;
            const float4 n0 = fine[2u * ic], n1 = fine[2u * ic + 1u];
            const LeafFar f = leaf_far(P, n0, n1);
            const bool far = f.m > fmaxf(0.5f * n1.x, kMu * f.S);   // NaN / inf: not far
            unsigned long long near = HRT_BALLOT(has && !far);
            HRT_STAT(kind, 8, 1);
#ifdef HRT_UNIT_CLOCKS
            if (lane == 0) ws[14] += (uint32_t)__popcll(near);
#endif
            while (near) {   // the rows of four near spheres make one culling round
                uint32_t s4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    s4[q] = HRT_NO_HIT;
                    if (near) {
                        s4[q] = base + (uint32_t)__builtin_ctzll(near);
                        near &= near - 1ull;
                    }
                }
                const uint32_t g = lane >> 4;
                const uint32_t mine = g == 0u ? s4[0] : (g == 1u ? s4[1] : (g == 2u ? s4[2] : s4[3]));
                const uint32_t jl = mine * HRT_FINE_ROWS + (lane & 15u);
                bool cand = mine != HRT_NO_HIT && jl < num_tri;
                float4 c0 = make_float4(0.f, 0.f, 0.f, 0.f), c1 = c0, c2 = c0;
                if (cand) {
                    c0 = tri[HRT_ROW * jl]; c1 = tri[HRT_ROW * jl + 1]; c2 = tri[HRT_ROW * jl + 2];
                    cand = !packet_culls(P, c0, c1, c2, tri[HRT_ROW * jl + 3], tri[HRT_ROW * jl + 4]);
                }
                HRT_STAT(kind, 2, __popcll(HRT_BALLOT(cand)));
                HRT_STAT(kind, 7, 1);
                park(cand, jl, c0, c1, c2);
            }
        }
    }
    // ---- plane tree: level pl_levels-1 = top ... level 0 = cones of the 64-entry leaves (as closest_hit_big) ----
    {
        const uint32_t L = A.pl_levels - 1u;
        auto visit = [&](uint32_t k, uint32_t base) -> unsigned long long {
            const uint32_t count = k == 0u ? A.pl_count[0] : (k == 1u ? A.pl_count[1] : A.pl_count[2]);
            const float4 *arr = reinterpret_cast<const float4 *>(k == 0u ? A.pl_node[0] : (k == 1u ? A.pl_node[1] : A.pl_node[2]));
            const uint32_t idx = base + lane;
            const bool has = idx < count;
            const uint32_t ic = has ? idx : base;
            const float4 n0 = arr[2u * ic], n1 = arr[2u * ic + 1u];
            const float sdot = fabsf(fdot3(P.ax, {n0.x, n0.y, n0.z}));
            const float thr = __builtin_fmaf(P.sina, n1.x, P.cosa * n0.w) * 1.0001f + 1e-6f;
            const bool skip = sdot > thr;                    // NaN: visit
            HRT_STAT(kind, 9, 1);
            return HRT_BALLOT(has && !skip);
        };
        uint32_t k = L;
        {
            const unsigned long long m0 = visit(k, 0u);
            if (lane == 0) { nmask[k] = m0; nbase[k] = 0u; }
        }
        for (;;) {
            const unsigned long long m = uni64(nmask[k]);
            if (m == 0ull) {
                if (k == L) break;
                ++k;
                continue;
            }
            const uint32_t b = (uint32_t)__builtin_ctzll(m);
            const uint32_t idx = uni(nbase[k]) + b;
            if (lane == 0) nmask[k] = m & (m - 1ull);
            if (k == 0u) {   // a leaf of 64 triangle ids: each judged on its own
                const uint32_t e = idx * 64u + lane;
                const uint32_t j = A.pl_index[e];
                const float4 *rec = reinterpret_cast<const float4 *>(A.pl_rec) + 2u * e;
                const float4 r0 = rec[0], r1 = rec[1];
                bool cand = j != HRT_NO_HIT;
                HRT_STAT(kind, 10, 1);
#ifdef HRT_UNIT_CLOCKS
                if (lane == 0) ws[12] += 1u;
#endif
                if (cand) {
                    const F3 n = {r1.x, r1.y, r1.z};
                    const float amin = __builtin_fmaf(fabsf(fdot3(P.ax, n)), P.cosa, -P.sina);
                    bool safe = amin > kGammaPerQs * r1.w;           // the spheres vouch for it
                    if (!safe) {
                        const LeafFar f = leaf_far(P, r0, make_float4(r0.w, 0.f, 0.f, 0.f));   // ball (p1, l)
                        const F3 sb = sub3(P.bc, {r0.x, r0.y, r0.z});
                        const float hmin = fabsf(fdot3(n, sb)) - P.br * 1.0001f -
                                           2e-6f * ((fabsf(sb.x) + fabsf(sb.y)) + fabsf(sb.z));
                        const float Ti = r1.w * (f.S + r0.w);
                        safe = (f.m > 0.f) & ((amin * f.phi > Ti) | (hmin * f.kappa > Ti));
                    }
                    cand = !safe;
                }
                if (HRT_BALLOT(cand) != 0ull) {
                    HRT_STAT(kind, 11, 1);
                    float4 c0 = make_float4(0.f, 0.f, 0.f, 0.f), c1 = c0, c2 = c0;
                    if (cand) {
                        c0 = tri[HRT_ROW * j]; c1 = tri[HRT_ROW * j + 1]; c2 = tri[HRT_ROW * j + 2];
                        cand = !packet_culls(P, c0, c1, c2, tri[HRT_ROW * j + 3], tri[HRT_ROW * j + 4]);
                    }
                    park(cand, j, c0, c1, c2);
                }
            } else {
                --k;
                const unsigned long long mk = visit(k, idx * 64u);
                if (lane == 0) { nmask[k] = mk; nbase[k] = idx * 64u; }
            }
        }
    }
    flush();
    return {who, best};
}


// Called by ALL lanes of a wave (uniform control flow); lanes with valid == false carry a dummy
// ray and their result is meaningless.  VARIANT: 0 plain, 1 staged, 2 / 3 flat packet culling
// (one block / many), 4 / 5 packet culling behind the leaf spheres + guard (one block / many).
template <int VARIANT, typename TriPtr, typename TgPtr, typename LeafPtr>
__device__ __forceinline__ Hit closest_hit(TriPtr tri, TgPtr tg, LeafPtr leaf, const hrt_kaccel &A,
                                           const hrt_krxt &X, uint32_t rxk,
                                           const uint32_t *__restrict__ orig, uint32_t num_tri, F3 o,
                                           F3 d, bool valid, uint32_t lane, const Ball &B,
                                           const bool shadow, F3 apex, unsigned long long *wmask,
                                           float4 *wleaf, int kind, const WideQ &wq = WideQ{})
{
    if constexpr (VARIANT == 0) {
        Hit h = {HRT_NO_HIT, 1e9f};
        if (valid) h = closest_hit_plain(tri, orig, num_tri, o, d);
        return h;
    } else if constexpr (VARIANT == 1) {
        Hit h = {HRT_NO_HIT, 1e9f};
        if (valid) h = closest_hit_staged(tri, orig, num_tri, o, d);
        return h;
    } else if constexpr (VARIANT == 2 || VARIANT == 3) {
        return closest_hit_packet<(VARIANT == 3)>(tri, orig, X, rxk, num_tri, o, d, valid, lane, B, shadow,
                                                  apex, wmask, kind);
    } else if constexpr (VARIANT == 6) {
        return closest_hit_big(tri, leaf, A, num_tri, o, d, valid, lane, shadow, apex, wmask,
                               reinterpret_cast<uint32_t *>(wleaf), kind);
    } else if constexpr (VARIANT == 9) {
        return closest_hit_fine(tri, A, num_tri, o, d, valid, lane, B, shadow, apex, wmask,
                                reinterpret_cast<uint32_t *>(wleaf), kind, wq);
    } else {
        return closest_hit_tree<(VARIANT == 5)>(tri, tg, leaf, orig, num_tri, o, d, valid, lane, B, shadow,
                                                apex, wmask, wleaf, kind);
    }
}

// acos in double of the float dot product, stored to float, folded to [0, pi/2] with the
// float pi (src/compute_paths.c:281-283).
__device__ __forceinline__ float incidence_angle_inl(F3 n, F3 d)
{
    float th = (float)acos((double)dot3(n, d));
    if (th > kPi * 0.5f) th = kPi - th;   // (double)th > (double)pi_f/2. is the same test
    return th;
}
__device__ __noinline__ float incidence_angle(F3 n, F3 d) { return incidence_angle_inl(n, d); }

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
// Shading is per hit, not per triangle test: it is kept OUT OF LINE so that its double-
// precision polynomial constants and temporaries do not inflate the register allocation of
// the intersection loops (174 -> ~80 VGPRs: 2 -> 5+ waves per SIMD).
__device__ __forceinline__ float4 fresnel_inl(float4 m0, float4 m1, float4 m2, float th)
{
    float s1, c1;
    hrt_sincosf(th, &s1, &c1);   // sinf(th) and cosf(th) (:310, :325) from one reduction
    if (m2.z * s1 > 1.f - kEps) return make_float4(1.f, 0.f, 1.f, 0.f);
    const float s2 = s1 * s1;
    const float c2r = sqrtf(1.f + m0.z / m2.y * s2);
    const float c2i = sqrtf(1.f - m1.z / m2.y * s2);
    const float pr = m0.y * c2r - m1.y * c2i;
    const float pi = m0.y * c2i + m1.y * c2r;
    float4 R;
    complex_div(c1 - pr, -pi, c1 + pr, pi, R.x, R.y);
    const float qr = m0.y * c1;
    const float qi = m1.y * c1;
    complex_div(qr - c2r, qi - c2i, qr + c2r, qi + c2i, R.z, R.w);
    R.x *= m2.w; R.y *= m2.w; R.z *= m2.w; R.w *= m2.w;
    return R;
}
__device__ __noinline__ float4 fresnel(float4 m0, float4 m1, float4 m2, float th) { return fresnel_inl(m0, m1, m2, th); }

// src/compute_paths.c:359-415; s = scattering coefficient, alpha = s1_alpha (small integer)
// (kept out of line, like incidence_angle: inlined into the records kernel they spill -- C3 1.125 -> 1.39 ms with the
// angle inlined, 1.78 with both, at 6 waves; 1.33 at 4)
__device__ __noinline__ float4 scatter_pattern(float s, float alpha, float th_s, float th_i)
{
    const float cs = hrt_cosf_nb(th_s);
    float si, ci;
    hrt_sincosf(th_i, &si, &ci);
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
    return make_float4(te, tei, tm, tmi);
}

// (inlined: the call sequence costs as much as the half of the function a lane runs)
__device__ __forceinline__ float acos_f_ool(float x) { return hrt_acosf(x); }


// Field arrays are addressed through BUFFER RESOURCES: one 128-bit descriptor (SGPRs) per block of
// field arrays -- the hit list of a bounce, the records of a (bounce, rx), the results of a trace
// kind --, the field's offset inside the block as the instruction's scalar offset, and the
// entry's byte offset i*4 as ONE VGPR shared by all fields (`buffer_load_dword v, v_off, s[rsrc],
// s_field offen`).  Plain `field[i]` indexing makes the compiler form a 64-bit address per field
// in VGPRs (it cannot know that i*4 fits 32 bits): two VGPRs and a v_lshl_add_u64 per access,
// ~120 VALU instructions and ~40 VGPRs in the shade kernel (102 -> 7x VGPRs).  One descriptor per
// FIELD would do too, but the compiler hoists them all out of the loops and spills SGPRs.
// The host guarantees HRT_HIT_FIELDS * cap * 4 < 2^32 (hrt_layout_query), so every scalar offset
// fits; the descriptors' range is the whole 32-bit offset space (bounds are the host's business:
// hrt_trace checks the workspace size).
// (Rsrc / make_rsrc: defined above, in front of closest_hit_patch)
__device__ __forceinline__ float ldf(Rsrc r, uint32_t field_off, uint32_t byte_off)
{
    return __uint_as_float((uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, (int)field_off, 0));
}
__device__ __forceinline__ uint32_t ldu(Rsrc r, uint32_t field_off, uint32_t byte_off)
{
    return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, (int)field_off, 0);
}
__device__ __forceinline__ void stf(Rsrc r, uint32_t field_off, uint32_t byte_off, float v)
{
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, (int)byte_off, (int)field_off, 0);
}
__device__ __forceinline__ void stu(Rsrc r, uint32_t field_off, uint32_t byte_off, uint32_t v)
{
    __builtin_amdgcn_raw_buffer_store_b32(v, r, (int)byte_off, (int)field_off, 0);
}
// The same with a cache policy (AUX 16 = sc1: agent scope -- the access goes through the XCD's L2 to memory.
// hrt_chain_kernel hands a bounce's survivors to workgroups on other XCDs INSIDE one kernel this way: no fence,
// no L2 write-back)
template <int AUX>
__device__ __forceinline__ float ldf_x(Rsrc r, uint32_t field_off, uint32_t byte_off)
{
    return __uint_as_float((uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, (int)field_off, AUX));
}
template <int AUX>
__device__ __forceinline__ uint32_t ldu_x(Rsrc r, uint32_t field_off, uint32_t byte_off)
{
    return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, (int)field_off, AUX);
}
template <int AUX>
__device__ __forceinline__ void stf_x(Rsrc r, uint32_t field_off, uint32_t byte_off, float v)
{
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, (int)byte_off, (int)field_off, AUX);
}
// per-lane gathers from the triangle / mesh tables (shade kernel): row j, byte `at` inside the row
__device__ __forceinline__ F3 gather3(Rsrc r, uint32_t row_bytes, uint32_t j, uint32_t at)
{
    const auto v = __builtin_amdgcn_raw_buffer_load_b96(r, (int)(j * row_bytes + at), 0, 0);
    return {__uint_as_float((uint32_t)v[0]), __uint_as_float((uint32_t)v[1]), __uint_as_float((uint32_t)v[2])};
}
__device__ __forceinline__ float4 gather4(Rsrc r, uint32_t row_bytes, uint32_t j, uint32_t at)
{
    const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(j * row_bytes + at), 0, 0);
    return make_float4(__uint_as_float((uint32_t)v[0]), __uint_as_float((uint32_t)v[1]),
                       __uint_as_float((uint32_t)v[2]), __uint_as_float((uint32_t)v[3]));
}
// blocks (include/hrt_device.h): hit list of bounce b; records of (bounce b, rx); results of kind k
__device__ __forceinline__ Rsrc hit_blk(const hrt_kparams &P, uint32_t b)
{
    return make_rsrc(P.ws + P.off_hits + (uint64_t)b * P.hit_block_bytes);
}
__device__ __forceinline__ Rsrc hit_out(const hrt_kparams &P, uint32_t b)
{
    return P.sort.enabled ? make_rsrc(P.ws + P.sort.off_scratch) : hit_blk(P, b);
}
__device__ __forceinline__ Rsrc rec_blk(const hrt_kparams &P, uint32_t b, uint32_t rx)
{
    return make_rsrc(P.ws + P.off_recs + (uint64_t)b * P.rec_block_bytes + (uint64_t)rx * 9u * P.cap * 4u);
}
__device__ __forceinline__ Rsrc res_blk(const hrt_kparams &P, uint32_t k)
{
    return make_rsrc(P.ws + P.off_res + (uint64_t)(2u * k) * P.cap * 4u);
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

// result of trace k of live-list entry i: closest triangle (or HRT_NO_HIT) and its distance
__device__ __forceinline__ uint32_t *res_tri(const hrt_kparams &P, uint32_t k)
{
    return reinterpret_cast<uint32_t *>(P.ws + P.off_res + (uint64_t)(2u * k) * P.cap * 4u);
}
__device__ __forceinline__ float *res_t(const hrt_kparams &P, uint32_t k)
{
    return reinterpret_cast<float *>(P.ws + P.off_res + (uint64_t)(2u * k + 1u) * P.cap * 4u);
}

// src/compute_paths.c:452-455 -- ray i of the launch set.  Lane i takes the i-th ray of the
// COHERENT launch order (P.order), so that a wave is a narrow ray packet.
__device__ __forceinline__ void launch_ray(const hrt_kparams &P, uint32_t i, uint32_t &ray, F3 &o,
                                           F3 &d, uint32_t &tx)
{
    tx = i / P.num_local;
    const uint32_t pos = i - tx * P.num_local;
    const uint32_t il = P.order ? P.order[pos] : pos;
    ray = tx * P.num_local + il;
    o = {P.tx_pos[3 * tx], P.tx_pos[3 * tx + 1], P.tx_pos[3 * tx + 2]};
    // the direction table is indexed by ray id, or (HRT_DIRS_IN_LAUNCH_ORDER) already permuted into
    // launch order: then a wave reads one contiguous 768-byte run instead of 64 scattered rows
    const uint64_t row = P.dirs_in_launch_order ? pos : il;
    d = {P.dirs[3 * row], P.dirs[3 * row + 1], P.dirs[3 * row + 2]};
}

// shadow ray from o towards rx (src/compute_paths.c:676-678): direction and distance
__device__ __forceinline__ F3 shadow_dir(F3 o, F3 rx, float &d2rx)
{
    F3 w = sub3(rx, o);
    d2rx = sqrtf(dot3(w, w));
    return {w.x / d2rx, w.y / d2rx, w.z / d2rx};
}

// ===================================================================================
// TRACE kernel: pure geometry.  Launch b owes, for every entry i of the live list (the rays
// that hit at bounce b-1; at b = 0 the launch set), num_rx shadow traces (b >= 1) and one
// primary trace (b < num_bounces).  A work unit is (chunk of 256 entries, trace kind k); a
// workgroup walks units, a wave does ONE trace per unit for 64 rays: it needs only the ray
// origin/direction, so the register footprint is small and the occupancy high.  Results
// (closest triangle, distance) go to res_tri/res_t[k][i]; all shading is the SHADE kernel's.
// LDS image: [num_tri*5 float4 rows (if staged)][num_rx RX pos][4 waves x 16 u64 masks][4 u32]
// ===================================================================================
#ifndef HRT_TRACE_WAVES_PER_SIMD
#define HRT_TRACE_WAVES_PER_SIMD 1
#endif
#ifndef HRT_HALF_RESULTS_MAX
#define HRT_HALF_RESULTS_MAX 0x7fffu   /* tables of fewer triangles: shadow results are half words (0 = never) */
#endif
#ifndef HRT_TRACE_WAVES_V2
#define HRT_TRACE_WAVES_V2 7   /* the small-table packet kernel: at most 72 VGPRs (measured C3: 1.704 ms at 7, 1.706 at 8 with 28 B of scratch, 1.743 at 6, 1.757 unconstrained) */
#endif
#ifndef HRT_TRACE_WAVES_FINE
#define HRT_TRACE_WAVES_FINE 8   /* the fine walk is bound by dependent loads: 8 waves per SIMD (64 registers, some scratch) beat
                                  * the 5 it takes unconstrained (95 registers): city 100 k 19.4 -> 18.4 ms, room 24 k 10.0 -> 9.4 */
#endif
template <bool TRI_IN_LDS, int VARIANT>
__global__ __launch_bounds__(HRT_BLOCK, (VARIANT == 2 ? HRT_TRACE_WAVES_V2 : (VARIANT == 9 ? HRT_TRACE_WAVES_FINE : HRT_TRACE_WAVES_PER_SIMD))) void hrt_trace_kernel(
    const hrt_kparams P, const uint32_t b)
{
    extern __shared__ float4 lds[];
    const uint32_t tid = threadIdx.x;
    const bool first = (b == 0);
    const uint32_t *counts = reinterpret_cast<const uint32_t *>(P.ws + P.off_counts);
    if (counts[P.num_bounces + 1] & kErrVoid) return;   // a fused launch of this trace timed out: the host redoes the step
    const uint32_t n_in = first ? P.n0 : counts[b];
    const uint32_t n_chunks = (n_in + HRT_BLOCK - 1) / HRT_BLOCK;
    // unit types of this launch: the shadow rays (b >= 1), one type per RX (k = rx) -- with patch tables
    // they are hrt_records_kernel's, not this kernel's -- and the primary rays (b < num_bounces; k = num_rx)
    const bool psa = VARIANT == 2 && P.patch.mask != nullptr && !first;   // (shadow rays: hrt_records_kernel's)
    const uint32_t sh_units = (first || psa) ? 0u : P.num_rx;
    const uint32_t kinds = sh_units + ((b < P.num_bounces) ? 1u : 0u);
    // (static deal: chunks padded to a multiple of 8, see the XCD-aware numbering below)
    constexpr bool kPull = (VARIANT == 6);   // units pulled from a counter (uneven unit costs; on the fine walk
                                             // pulling was measured and lost: 15.7 -> 18.5 ms, the XCD-aware deal matters more)
    const uint32_t n_units = (kPull ? n_chunks : ((n_chunks + 7u) & ~7u)) * kinds;   // < 2^32
    if (blockIdx.x >= n_units) return;

    const uint32_t T = P.num_tri;
    const uint32_t cap4 = (uint32_t)P.cap * 4u;   // bytes per field array
    const float4 *g_tri = reinterpret_cast<const float4 *>(P.tri);
    const float2 *g_tg = reinterpret_cast<const float2 *>(P.acc.tg);
    const float4 *g_leaf = reinterpret_cast<const float4 *>(P.acc.leaf);
    const uint32_t n_leaf = P.acc.num_leaf;
    // LDS image (hrt_hip_launch_trace sizes it): [T x 5 float4 rows | T float2 guard pairs, padded to
    // 16 B | 2 float4 per leaf] (only if staged) [num_rx RX pos][4 waves x 16 u64 masks]
    // [4 waves x 16 float4 leaf constants][4 u32]
    float4 *l_tri = lds;
    float2 *l_tg = reinterpret_cast<float2 *>(lds + HRT_ROW * T);
    float4 *l_leaf = lds + HRT_ROW * T + (T + 1u) / 2u;
    float4 *l_rx = TRI_IN_LDS ? l_leaf + 2u * n_leaf : lds;
    unsigned long long *l_mask = reinterpret_cast<unsigned long long *>(l_rx + P.num_rx) +
                                 (tid >> 6) * kMaskRounds;
    // (per wave 2 * kMaskRounds float4 = 128 words of scratch: leaf constants / the tree walks' stacks and queues)
    float4 *l_wleaf = reinterpret_cast<float4 *>(reinterpret_cast<unsigned long long *>(l_rx + P.num_rx) +
                                                 (HRT_BLOCK / 64u) * kMaskRounds) + (tid >> 6) * wave_scratch4(VARIANT == 9);
    uint32_t *l_wcnt = reinterpret_cast<uint32_t *>(
        reinterpret_cast<float4 *>(reinterpret_cast<unsigned long long *>(l_rx + P.num_rx) +
                                   (HRT_BLOCK / 64u) * kMaskRounds) + (HRT_BLOCK / 64u) * wave_scratch4(VARIANT == 9));
    if (TRI_IN_LDS) {
        for (uint32_t k = tid; k < HRT_ROW * T; k += HRT_BLOCK) l_tri[k] = g_tri[k];
        if constexpr (VARIANT >= 4) {   // guard pairs and leaf records: only the tree variants read them
            for (uint32_t k = tid; k < T; k += HRT_BLOCK) l_tg[k] = g_tg[k];
            for (uint32_t k = tid; k < 2u * n_leaf; k += HRT_BLOCK) l_leaf[k] = g_leaf[k];
        }
    }
    for (uint32_t k = tid; k < P.num_rx; k += HRT_BLOCK)
        l_rx[k] = make_float4(P.rx_pos[3 * k], P.rx_pos[3 * k + 1], P.rx_pos[3 * k + 2], 0.f);
    __syncthreads();
    auto tri = [&]() {
        if constexpr (TRI_IN_LDS) return (const float4 *)l_tri;
        else return g_tri;
    }();
    auto tg = [&]() {
        if constexpr (TRI_IN_LDS) return (const float2 *)l_tg;
        else return g_tg;
    }();
    auto leaf = [&]() {
        if constexpr (TRI_IN_LDS) return (const float4 *)l_leaf;
        else return g_leaf;
    }();
    const uint32_t lane = tid & 63u;

    // Units are dealt statically (unit = blockIdx + k * gridDim) -- except on big tables, where one
    // trace can cost a hundred times another (a wave whose rays scattered is walked ray by ray):
    // there the workgroups pull units from a counter (one returning atomic per unit: ~1 us against
    // units of tens of us; on small tables the static deal is cheaper).
    uint32_t *unit_ctr = reinterpret_cast<uint32_t *>(P.ws + P.off_counts + P.cnt_stride) + b;
    for (uint32_t unit = blockIdx.x;; unit += gridDim.x) {
        if constexpr (kPull) {
            if (tid == 0) l_wcnt[0] = atomicAdd(unit_ctr, 1u);
            __syncthreads();
            unit = l_wcnt[0];
            __syncthreads();
        }
        if (unit >= n_units) break;
        // XCD-aware numbering (static deal): workgroups b and b + 8 share an XCD and its L2, so the
        // `kinds` traces of one chunk -- which re-read the same origins -- go to workgroups 8 apart
        // instead of to `kinds` neighbouring workgroups on different XCDs.  With q = unit / 8 and
        // x = unit % 8: chunk = (q / kinds) * 8 + x, kind = q % kinds (a bijection on the padded range;
        // chunks past the end are skipped).
        uint32_t chunk, kq;   // kq: unit type, shadow types first
        if constexpr (kPull) {
            chunk = unit / kinds;
            kq = unit - chunk * kinds;
        } else {
            const uint32_t q = unit >> 3, x = unit & 7u;
            chunk = (q / kinds) * 8u + x;
            kq = q % kinds;
            if (chunk >= n_chunks) continue;   // padding of the last group of 8 chunks
        }
        const uint32_t i = chunk * HRT_BLOCK + tid;
        const uint32_t i4 = i * 4u;
        const bool valid = i < n_in;
        const bool shadow = kq < sh_units;
        const uint32_t k = shadow ? kq : P.num_rx;   // (without patch tables: the RX of a shadow unit)
        F3 o = {0.f, 0.f, 0.f}, d = {0.f, 0.f, 1.f};
        uint32_t tx_lane = 0u;
        if (valid) {
            if (first) {
                uint32_t ray;
                launch_ray(P, i, ray, o, d, tx_lane);
            } else {
                const uint32_t pb = b - 1;
                o = {ldf(hit_blk(P, pb), H_OX * cap4, i4), ldf(hit_blk(P, pb), H_OY * cap4, i4),
                     ldf(hit_blk(P, pb), H_OZ * cap4, i4)};
                if (!shadow)
                    d = {ldf(hit_blk(P, pb), H_DX * cap4, i4), ldf(hit_blk(P, pb), H_DY * cap4, i4),
                         ldf(hit_blk(P, pb), H_DZ * cap4, i4)};
            }
        }
        F3 apex = {0.f, 0.f, 0.f};
        uint32_t apex_k = k;   // which direction table serves this trace: the RX's, or (launch) the TX's
        if (shadow) {
            const float4 rp = l_rx[k];
            apex = {rp.x, rp.y, rp.z};
            float d2rx;
            const F3 w = shadow_dir(o, apex, d2rx);
            if (valid) d = w;
        } else if (first && P.rxt.num_txt != 0u) {
            uint32_t tx = 0u;
            if (P.num_tx != 1u) {
                tx = (uint32_t)__builtin_amdgcn_readfirstlane((int)((chunk * HRT_BLOCK + (tid & ~63u)) / P.num_local));
                tx = min(tx, P.num_tx - 1u);
            }
            apex = {P.tx_pos[3 * tx], P.tx_pos[3 * tx + 1], P.tx_pos[3 * tx + 2]};
            apex_k = P.num_rx + tx;
        }
        // (tables of <= 64 triangles: apex-bound traces go through the per-cell candidate masks)
        const bool masked = VARIANT == 2 && P.rxt.cell_mask != nullptr && (shadow || first);
        Ball ball = {{0.f, 0.f, 0.f}, 0.f, false};
        if constexpr (VARIANT >= 2 && VARIANT != 6)
            if (!masked && (VARIANT != 2 || shadow || first)) ball = origin_ball(o, valid);
        WideQ wq;
        if constexpr (VARIANT == 9) {
            if (P.wide_cap != 0u) {
                wq.cnt = reinterpret_cast<uint32_t *>(P.ws + P.off_counts + 2u * P.cnt_stride) + b;
                wq.q = reinterpret_cast<unsigned long long *>(P.ws + P.off_wide_q);
                wq.keys = reinterpret_cast<unsigned long long *>(P.ws + P.off_wide_key);
                wq.cap = P.wide_cap;
                wq.cos_min = P.wide_cos;
                wq.entry = ((unsigned long long)chunk << 32) | ((unsigned long long)k << 2) | (tid >> 6);
            }
        }
#ifdef HRT_KERNEL_STATS
        const long long t_unit0 = clock64();
#endif
#ifdef HRT_UNIT_CLOCKS
        const unsigned long long t_w0 = wall_clock64();
#endif
        Hit h = {HRT_NO_HIT, 1e9f};
        bool done = false;
        if constexpr (VARIANT == 2) {
            // The primary rays of a later launch, split by VIRTUAL SOURCE.  Rays that took the same sequence of
            // reflections leave one image of their TX (o - d * path length), and neighbours in launch order that
            // did so form a narrow packet; a wave that mixes sequences is a wide packet that culls little or
            // nothing.  Measured on C3 (profiles/study/split_study.py): 1.13 / 1.24 groups per wave at launches 2 /
            // 3, and the candidates drop from 14.2 / 20.8 per wave to 3.8 / 4.1 -- the mixed waves carried them.
            // The lanes of one virtual source go through packet culling as ONE packet, group after group; any
            // partition is sound (each group's packet is tested for itself), this one is just the coherent one.
            if (!done && !shadow && !first) {
                const float lp = valid ? ldf(hit_blk(P, b - 1), H_TAU * cap4, i4) * kC : 0.f;
                const F3 vs = {o.x - d.x * lp, o.y - d.y * lp, o.z - d.z * lp};
                unsigned long long rem = HRT_BALLOT(valid);
                uint32_t groups = 0u;
                while (rem) {
                    const uint32_t l0 = (uint32_t)__builtin_ctzll(rem);
                    const F3 v0 = {lane63f(vs.x, l0), lane63f(vs.y, l0), lane63f(vs.z, l0)};
                    const float tol = 1e-2f + 1e-5f * lane63f(lp, l0);
                    bool in = ((rem >> lane) & 1ull) != 0ull;
                    // (the last group takes whatever is left; lane l0 is in its own group whatever its numbers are)
                    if (++groups < kMaxSourceGroups)
                        in = in && ((fabsf(vs.x - v0.x) + fabsf(vs.y - v0.y)) + fabsf(vs.z - v0.z) <= tol || lane == l0);
                    const unsigned long long gm = HRT_BALLOT(in);
                    const Ball gb = origin_ball(o, in);
                    const Hit hg = closest_hit<VARIANT>(tri, tg, leaf, P.acc, P.rxt, apex_k, P.acc.orig, T, o, d, in, lane, gb,
                                                        false, apex, l_mask, l_wleaf, 1, wq);
                    if (in) h = hg;
                    rem &= ~gm;
                }
                done = true;
            }
        }
        if (!done)
        h = masked ? closest_hit_masked(tri, P.acc.orig, P.rxt, shadow ? k : P.num_rx + tx_lane, T, o, d, valid,
                                                  lane, shadow ? 2 : 0)
                             : closest_hit<VARIANT>(tri, tg, leaf, P.acc, P.rxt, apex_k, P.acc.orig, T, o, d, valid, lane,
                                                    ball, shadow, apex, l_mask, l_wleaf, shadow ? 2 : (first ? 0 : 1), wq);
#ifdef HRT_UNIT_CLOCKS
        if (lane == 0) {
            const unsigned long long dt = wall_clock64() - t_w0;
            // (no counter: half a million same-address atomics per step would be the measurement)
            const unsigned int slot = (b * 600011u + unit * 4u + (tid >> 6)) & ((1u << 21) - 1u);
            if (unit == 0u && tid == 0u) atomicMax(&g_unit_n, 1u << 21);
            g_unit[slot][0] = t_w0;
            g_unit[slot][1] = dt | ((unsigned long long)(shadow ? 2 : (first ? 0 : 1)) << 56) |
                              ((unsigned long long)(reinterpret_cast<uint32_t *>(l_wleaf)[15] & 1u) << 60) |
                              ((unsigned long long)min(reinterpret_cast<uint32_t *>(l_wleaf)[14], 0xfffu) << 32) |
                              ((unsigned long long)min(reinterpret_cast<uint32_t *>(l_wleaf)[12], 0xfffu) << 44) |
                              ((unsigned long long)(reinterpret_cast<uint32_t *>(l_wleaf)[13] & 0xffu) << 24);
        }
#endif
#ifdef HRT_KERNEL_STATS
        if (lane == 0) {   // per wave-trace: longest and total duration in shader clocks
            const unsigned long long dt = (unsigned long long)(clock64() - t_unit0);
            atomicMax(&g_stats[shadow ? 2 : (first ? 0 : 1)][13], dt);
            atomicAdd(&g_stats[shadow ? 2 : (first ? 0 : 1)][14], dt);
        }
#endif
        // (a packet queued for the wide kernels: results and survivor count are hrt_wide_finish_kernel's)
        const bool deferred = VARIANT == 9 && h.tri == HRT_DEFERRED;
        if (valid && !deferred) {
            // a shadow trace is consumed as "which triangle (for the incidence angle), and is it
            // within 1 m (the reference's blocking test, quirk Q6)": one word, bit 31 = blocked,
            // 0x7fffffff = nothing hit; the bounce itself needs triangle and distance
            if (shadow) {
                // (tables of fewer than 2^15 - 1 triangles: a HALF word, bit 15 = blocked, 0x7fff = nothing hit)
                if (T < HRT_HALF_RESULTS_MAX) {
                    const uint32_t code16 = (h.tri == HRT_NO_HIT) ? 0x7fffu : (h.tri | ((h.t <= 1.f) ? 0x8000u : 0u));
                    __builtin_amdgcn_raw_buffer_store_b16((short)code16, res_blk(P, k), (int)(i * 2u), 0, 0);
                } else {
                    const uint32_t code = (h.tri == HRT_NO_HIT) ? 0x7fffffffu : (h.tri | ((h.t <= 1.f) ? 0x80000000u : 0u));
                    stu(res_blk(P, k), 0u, i4, code);
                }
            } else {
                stu(res_blk(P, k), 0u, i4, h.tri);
                stf(res_blk(P, k), cap4, i4, h.t);
            }
        }
        // the bounce itself: how many rays of this chunk survive.  With the counts of all chunks
        // known BEFORE the shade kernel runs, that kernel can write every survivor straight to
        // its final place in the next live list (stable compaction without a staging copy).
        // (one word per WAVE of the chunk: the sum is the shade kernel's)
        if constexpr (VARIANT == 9) {
            // big tables: one trace can cost a hundred times another, so the four waves of a workgroup do
            // not wait for each other between units -- each publishes its own count
            if (!shadow && !deferred) {
                const unsigned long long hm = __ballot(valid && h.tri != HRT_NO_HIT);
                if (lane == 0) {
                    const uint32_t c = (uint32_t)__popcll(hm);
                    reinterpret_cast<uint32_t *>(P.ws + P.off_chunk_cnt)[chunk * 4u + (tid >> 6)] = c;
                    atomicAdd(reinterpret_cast<uint32_t *>(P.ws + P.off_super_cnt) +
                                  (uint64_t)b * P.num_super + (chunk >> HRT_SUPER_SHIFT), c);
                }
            }
        } else if (!shadow) {
            const unsigned long long hm = __ballot(valid && h.tri != HRT_NO_HIT);
            if (lane == 0) l_wcnt[tid >> 6] = (uint32_t)__popcll(hm);
            __syncthreads();
            if (tid == 0) {
                const uint32_t c = l_wcnt[0] + l_wcnt[1] + l_wcnt[2] + l_wcnt[3];
                reinterpret_cast<uint32_t *>(P.ws + P.off_chunk_cnt)[chunk] = c;
                // ... and the sum over every HRT_SUPER_CHUNKS chunks ("super-chunk"): two short sums in the
                // shade kernel then replace a scan pass over all chunks
                atomicAdd(reinterpret_cast<uint32_t *>(P.ws + P.off_super_cnt) +
                              (uint64_t)b * P.num_super + (chunk >> HRT_SUPER_SHIFT), c);
            }
            __syncthreads();
        }
    }
}

// ===================================================================================
// RECORDS kernel (patch tables, hrt_kpatch): for launch b >= 1, everything the scatter records of bounce
// b-1 need -- the shadow traces AND the records (src/compute_paths.c:671-723, quirks Q6-Q8).  A thread
// takes one entry of the live list: it loads its state, locates its patch ONCE, and walks the RXs IN ORDER:
// shadow direction, the wave's trace (closest_hit_patch: the masks of the lanes' patches ORed, the union
// walked through the staged test), theta carried through the shadow hits, the record.  Nothing of a packet
// is formed -- no origin ball, no cone, no culling round -- so the cost does not depend on how coherent the
// wave is; and the shadow direction is computed once, not once per kernel, with no result words in between.
// Defined behind the shading functions (below).
// ===================================================================================

// ===================================================================================
// WIDE kernels (big tables, behind the trace kernel of the fine walk): the packets that were too wide to
// cull -- 0.1-0.3 % of a launch after the re-sort, each owing EVERY triangle the exact staged test.
// hrt_wide_kernel: item = (queue entry, slice of HRT_WIDE_SLICE table rows), dealt statically to the waves
// of a fixed grid, slice-major (the waves working at one time share a slice: its rows stay in L2); a wave
// re-creates the 64 rays exactly as the trace kernel does, fetches the slice 64 rows at a time into its LDS
// buffer, runs the staged test and merges its per-ray minimum of (distance, ORIGINAL index) -- the tie
// rule of every walk -- into the entry's keys with a 64-bit atomic minimum (distances are positive floats:
// their bit patterns order like the values).  hrt_wide_finish_kernel: one wave per entry turns the keys into
// the trace kernel's result words and the wave's survivor count.
// ===================================================================================
__device__ __forceinline__ void wide_entry_ray(const hrt_kparams &P, const uint32_t b, const unsigned long long entry,
                                               const uint32_t lane, const uint32_t n_in, uint32_t &chunk, uint32_t &k,
                                               uint32_t &wv, uint32_t &i, bool &valid, bool &shadow, F3 &o, F3 &d)
{
    chunk = (uint32_t)(entry >> 32);
    k = ((uint32_t)entry) >> 2;
    wv = ((uint32_t)entry) & 3u;
    i = chunk * HRT_BLOCK + wv * 64u + lane;
    valid = i < n_in;
    shadow = k < P.num_rx;
    const uint32_t cap4 = (uint32_t)P.cap * 4u, i4 = i * 4u;
    o = {0.f, 0.f, 0.f};
    d = {0.f, 0.f, 1.f};
    if (valid) {
        if (b == 0) {
            uint32_t ray, tx_lane;
            launch_ray(P, i, ray, o, d, tx_lane);
        } else {
            const uint32_t pb = b - 1;
            o = {ldf(hit_blk(P, pb), H_OX * cap4, i4), ldf(hit_blk(P, pb), H_OY * cap4, i4),
                 ldf(hit_blk(P, pb), H_OZ * cap4, i4)};
            if (!shadow)
                d = {ldf(hit_blk(P, pb), H_DX * cap4, i4), ldf(hit_blk(P, pb), H_DY * cap4, i4),
                     ldf(hit_blk(P, pb), H_DZ * cap4, i4)};
        }
    }
    if (shadow) {
        float d2rx;
        const F3 w = shadow_dir(o, {P.rx_pos[3 * k], P.rx_pos[3 * k + 1], P.rx_pos[3 * k + 2]}, d2rx);
        if (valid) d = w;
    }
}

constexpr uint32_t kWideGrid = 8192u;   // workgroups of hrt_wide_kernel, 4 waves each (HRT_WIDE_GRID; city 100 k: 1 280 20.9 ms, 2 048 20.4, 4 096 19.7, 8 192 19.2, 16 384 19.1)
static_assert(HRT_WIDE_SLICE == 64u * HRT_FINE_ROWS, "a slice is one round of fine spheres");

// A slice of a USABLE packet (one that was queued because its cone is wide: its walk would take
// milliseconds in one wave) is culled like the fine walk culls -- the slice's 64 fine spheres in one round,
// the rows of near spheres through packet_culls -- but without the plane tree: the rows of the FAR spheres
// are looked at one by one (64 per round): a triangle that is not nearly parallel to the rays
// (|d.n| > Gamma_i for every ray) is rejected by the lemma's first term (DESIGN_ACCEL.md B.4), any other
// is judged by its own ball (p1, l) exactly as the plane tree's leaves judge theirs, and what is not
// cleared is packet-tested and, if it survives, staged-tested.
#ifndef HRT_WIDE_WAVES
#define HRT_WIDE_WAVES 6   /* (5 unconstrained at 89 registers; room of 24 012: 9.4 -> 9.05 ms at 6, 9.3 at 8) */
#endif
__global__ __launch_bounds__(HRT_BLOCK, HRT_WIDE_WAVES) void hrt_wide_kernel(const hrt_kparams P, const uint32_t b)
{
    __shared__ float4 l_rows[HRT_BLOCK / 64u][kCandBuf * 4u];
    const uint32_t *counts = reinterpret_cast<const uint32_t *>(P.ws + P.off_counts);
    const uint32_t nq = min(reinterpret_cast<const uint32_t *>(P.ws + P.off_counts + 2u * P.cnt_stride)[b], P.wide_cap);
    if (nq == 0u) return;
    const uint32_t n_in = (b == 0) ? P.n0 : counts[b];
    const uint32_t T = P.num_tri;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const float4 *__restrict__ tri = reinterpret_cast<const float4 *>(P.tri);
    const float2 *__restrict__ tg = reinterpret_cast<const float2 *>(P.acc.tg);
    const float4 *__restrict__ fine = reinterpret_cast<const float4 *>(P.acc.fine);
    const uint32_t *__restrict__ orig = P.acc.orig;
    const unsigned long long *q = reinterpret_cast<const unsigned long long *>(P.ws + P.off_wide_q);
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(P.ws + P.off_wide_key);
    float4 *cbuf = l_rows[wave];
    const uint64_t n_slices = ((uint64_t)T + HRT_WIDE_SLICE - 1u) / HRT_WIDE_SLICE;
    const uint64_t n_items = n_slices * nq;
    const bool cull = true;
    auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    [[maybe_unused]] const int kind = 1;
    for (uint64_t item = (uint64_t)blockIdx.x * (HRT_BLOCK / 64u) + wave; item < n_items;
         item += (uint64_t)gridDim.x * (HRT_BLOCK / 64u)) {
        const uint32_t slice = (uint32_t)(item / nq), e = (uint32_t)(item - (uint64_t)slice * nq);
        uint32_t chunk, k, wv, i;
        bool valid, shadow;
        F3 o, d;
        wide_entry_ray(P, b, q[e], lane, n_in, chunk, k, wv, i, valid, shadow, o, d);
        const unsigned long long inval = HRT_BALLOT(!valid);
        float best = 1e9f;
        uint32_t who = HRT_NO_HIT, who_o = 0u;
        const uint32_t row0 = slice * HRT_WIDE_SLICE, row1 = min(T, row0 + HRT_WIDE_SLICE);
        Packet Pk;
        Pk.usable = false;
        if (cull) {
            const Ball B = origin_ball(o, valid);
            F3 apex = {0.f, 0.f, 0.f};
            if (shadow) apex = {P.rx_pos[3 * k], P.rx_pos[3 * k + 1], P.rx_pos[3 * k + 2]};
            Pk = packet_bounds(B, d, valid, shadow, apex);
        }
        if (Pk.usable) {
            uint32_t nbuf = 0u;
            auto flush = [&]() {
                for (uint32_t t = 0; t < nbuf; ++t) {
                    const float4 *slot = cbuf + 4u * t;
                    const float4 c0 = slot[0], c1 = slot[1], c2 = slot[2];
                    const uint32_t j = uni(__float_as_uint(slot[3].x));
                    HRT_STAGED_TEST(j, c0, c1, c2)
                }
                nbuf = 0u;
            };
            auto park = [&](bool cand, uint32_t j, float4 c0, float4 c1, float4 c2) {   // all lanes call (uniform)
                const unsigned long long cm = HRT_BALLOT(cand);
                const uint32_t n = (uint32_t)__popcll(cm);
                if (n == 0u) return;
                if (nbuf + n > kCandBuf) flush();
                if (cand) {
                    float4 *slot = cbuf + 4u * (nbuf + lane_prefix(cm));
                    slot[0] = c0; slot[1] = c1; slot[2] = c2;
                    slot[3] = make_float4(__uint_as_float(j), 0.f, 0.f, 0.f);
                }
                nbuf += n;
            };
            const uint32_t s0 = row0 / HRT_FINE_ROWS;
            const uint32_t idx = s0 + lane;
            const bool has = idx < P.acc.num_fine;
            const uint32_t ic = has ? idx : s0;
            const float4 n0 = fine[2u * ic], n1 = fine[2u * ic + 1u];
            const LeafFar f = leaf_far(Pk, n0, n1);
            const bool far = f.m > fmaxf(0.5f * n1.x, kMu * f.S);   // NaN / inf: not far
            unsigned long long near = HRT_BALLOT(has && !far), farm = HRT_BALLOT(has && far);
            const uint32_t g = lane >> 4;
            while (near) {   // the rows of four near spheres make one culling round
                uint32_t s4[4];
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    s4[qq] = HRT_NO_HIT;
                    if (near) {
                        s4[qq] = s0 + (uint32_t)__builtin_ctzll(near);
                        near &= near - 1ull;
                    }
                }
                const uint32_t mine = g == 0u ? s4[0] : (g == 1u ? s4[1] : (g == 2u ? s4[2] : s4[3]));
                const uint32_t jl = mine * HRT_FINE_ROWS + (lane & 15u);
                bool cand = mine != HRT_NO_HIT && jl < T;
                float4 c0 = make_float4(0.f, 0.f, 0.f, 0.f), c1 = c0, c2 = c0;
                if (cand) {
                    c0 = tri[HRT_ROW * jl]; c1 = tri[HRT_ROW * jl + 1]; c2 = tri[HRT_ROW * jl + 2];
                    cand = !packet_culls(Pk, c0, c1, c2, tri[HRT_ROW * jl + 3], tri[HRT_ROW * jl + 4]);
                }
                park(cand, jl, c0, c1, c2);
            }
            while (farm) {   // the rows of four far spheres: the guard, triangle by triangle
                uint32_t s4[4];
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    s4[qq] = HRT_NO_HIT;
                    if (farm) {
                        s4[qq] = s0 + (uint32_t)__builtin_ctzll(farm);
                        farm &= farm - 1ull;
                    }
                }
                const uint32_t mine = g == 0u ? s4[0] : (g == 1u ? s4[1] : (g == 2u ? s4[2] : s4[3]));
                const uint32_t jl = mine * HRT_FINE_ROWS + (lane & 15u);
                bool cand = mine != HRT_NO_HIT && jl < T;
                float4 c0 = make_float4(0.f, 0.f, 0.f, 0.f), c2 = c0;
                if (cand) {
                    c0 = tri[HRT_ROW * jl]; c2 = tri[HRT_ROW * jl + 2];
                    const float2 gq = tg[jl];   // qs, longest edge
                    const F3 n = {c2.y, c2.z, c2.w};
                    const float amin = __builtin_fmaf(fabsf(fdot3(Pk.ax, n)), Pk.cosa, -Pk.sina);
                    bool safe = amin > kGammaPerQs * gq.x;           // the sphere vouches for it
                    if (!safe) {
                        const LeafFar fb = leaf_far(Pk, make_float4(c0.x, c0.y, c0.z, gq.y), make_float4(gq.y, 0.f, 0.f, 0.f));
                        const F3 sb = sub3(Pk.bc, {c0.x, c0.y, c0.z});
                        const float hmin = fabsf(fdot3(n, sb)) - Pk.br * 1.0001f -
                                           2e-6f * ((fabsf(sb.x) + fabsf(sb.y)) + fabsf(sb.z));
                        const float Ti = gq.x * (fb.S + gq.y);
                        safe = (fb.m > 0.f) & ((amin * fb.phi > Ti) | (hmin * fb.kappa > Ti));
                    }
                    cand = !safe;
                }
                if (HRT_BALLOT(cand) != 0ull) {
                    float4 c1 = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (cand) {
                        c1 = tri[HRT_ROW * jl + 1];
                        cand = !packet_culls(Pk, c0, c1, c2, tri[HRT_ROW * jl + 3], tri[HRT_ROW * jl + 4]);
                    }
                    park(cand, jl, c0, c1, c2);
                }
            }
            flush();
        } else {
            for (uint32_t base = row0; base < row1; base += 64u) {
                const uint32_t jl = base + lane;
                if (jl < row1) {
                    cbuf[3u * lane] = tri[HRT_ROW * jl];
                    cbuf[3u * lane + 1u] = tri[HRT_ROW * jl + 1];
                    cbuf[3u * lane + 2u] = tri[HRT_ROW * jl + 2];
                }
                const uint32_t n = min(64u, row1 - base);
                for (uint32_t t = 0; t < n; ++t) {
                    const float4 c0 = cbuf[3u * t], c1 = cbuf[3u * t + 1u], c2 = cbuf[3u * t + 2u];
                    const uint32_t j = base + t;
                    HRT_STAGED_TEST(j, c0, c1, c2)
                }
            }
        }
        if (valid && who != HRT_NO_HIT)
            atomicMin(keys + (uint64_t)e * 64u + lane, ((unsigned long long)__float_as_uint(best) << 32) | who_o);
    }
}

__global__ __launch_bounds__(HRT_BLOCK) void hrt_wide_finish_kernel(const hrt_kparams P, const uint32_t b)
{
    const uint32_t *counts = reinterpret_cast<const uint32_t *>(P.ws + P.off_counts);
    const uint32_t nq = min(reinterpret_cast<const uint32_t *>(P.ws + P.off_counts + 2u * P.cnt_stride)[b], P.wide_cap);
    const uint32_t n_in = (b == 0) ? P.n0 : counts[b];
    const uint32_t T = P.num_tri;
    const uint32_t cap4 = (uint32_t)P.cap * 4u;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long *q = reinterpret_cast<const unsigned long long *>(P.ws + P.off_wide_q);
    const unsigned long long *keys = reinterpret_cast<const unsigned long long *>(P.ws + P.off_wide_key);
    for (uint32_t e = blockIdx.x * (HRT_BLOCK / 64u) + wave; e < nq; e += gridDim.x * (HRT_BLOCK / 64u)) {
        const unsigned long long entry = q[e];
        const uint32_t chunk = (uint32_t)(entry >> 32), k = ((uint32_t)entry) >> 2, wv = ((uint32_t)entry) & 3u;
        const uint32_t i = chunk * HRT_BLOCK + wv * 64u + lane, i4 = i * 4u;
        const bool valid = i < n_in, shadow = k < P.num_rx;
        const unsigned long long key = keys[(uint64_t)e * 64u + lane];
        Hit h = {HRT_NO_HIT, 1e9f};
        if (key != ~0ull) {
            h.tri = P.wide_inv[(uint32_t)key];
            h.t = __uint_as_float((uint32_t)(key >> 32));
        }
        if (valid) {   // (the trace kernel's result words)
            if (shadow) {
                if (T < HRT_HALF_RESULTS_MAX) {
                    const uint32_t code16 = (h.tri == HRT_NO_HIT) ? 0x7fffu : (h.tri | ((h.t <= 1.f) ? 0x8000u : 0u));
                    __builtin_amdgcn_raw_buffer_store_b16((short)code16, res_blk(P, k), (int)(i * 2u), 0, 0);
                } else {
                    const uint32_t code = (h.tri == HRT_NO_HIT) ? 0x7fffffffu : (h.tri | ((h.t <= 1.f) ? 0x80000000u : 0u));
                    stu(res_blk(P, k), 0u, i4, code);
                }
            } else {
                stu(res_blk(P, k), 0u, i4, h.tri);
                stf(res_blk(P, k), cap4, i4, h.t);
            }
        }
        if (!shadow) {
            const unsigned long long hm = __ballot(valid && h.tri != HRT_NO_HIT);
            if (lane == 0) {
                const uint32_t c = (uint32_t)__popcll(hm);
                reinterpret_cast<uint32_t *>(P.ws + P.off_chunk_cnt)[chunk * 4u + wv] = c;
                atomicAdd(reinterpret_cast<uint32_t *>(P.ws + P.off_super_cnt) +
                              (uint64_t)b * P.num_super + (chunk >> HRT_SUPER_SHIFT), c);
            }
        }
    }
}

// ===================================================================================
// SHADE kernel: everything that is per ray and not intersection -- a streaming kernel.
// For entry i of the live list of launch b:
//   (b >= 1) scatter records of bounce b-1 for every RX IN ORDER, carrying theta through the
//            shadow results (src/compute_paths.c:671-723; quirks Q6, Q7, Q8);
//   (b < nb) the bounce itself: incidence angle, Fresnel, FSL, delay, reflect (:611-659), and
//            the stable compaction: every survivor of the 256-entry chunk goes straight to
//            (survivors of all earlier chunks) + (its rank in the chunk) of the next live list.
// LDS: [17*4 float4 materials][num_rx float4 RX pos][4 u32 wave counts][4 u32 wave sums]
// ===================================================================================
#ifndef HRT_SHADE_WAVES
#define HRT_SHADE_WAVES 6   /* at most 80 VGPRs: 6 waves/SIMD (5 is 7 % slower; 7 gains nothing, 8 spills and is
                             * 40 % slower) */
#endif
// NLDS: the per-triangle normals (+ mesh ids) and the mesh rows are staged in LDS (tables of up to
// kShadeLdsTri triangles / kShadeLdsMesh meshes): the record loop then reads nothing from global
// memory between its stores except the shadow result words, which are requested four RXs at a time in
// front of those RXs' records -- loads and stores share vmcnt in issue order, so a load behind a
// record's nine stores waits for them to reach L2.
constexpr uint32_t kShadeLdsTri = 2048u, kShadeLdsMesh = 256u;
template <bool NLDS>
__global__ __launch_bounds__(HRT_BLOCK, HRT_SHADE_WAVES) void hrt_shade_kernel(const hrt_kparams P,
                                                              const uint32_t b)
{
    extern __shared__ float4 lds[];
    const uint32_t tid = threadIdx.x;
    const bool first = (b == 0);
    const bool do_trace = (b < P.num_bounces);
    const uint32_t *counts = reinterpret_cast<const uint32_t *>(P.ws + P.off_counts);
    if (counts[P.num_bounces + 1] & kErrVoid) return;   // a fused launch of this trace timed out: the host redoes the step
    const uint32_t n_in = first ? P.n0 : counts[b];
    if ((uint64_t)blockIdx.x * HRT_BLOCK >= n_in) return;

    // per-lane rows of the triangle and mesh tables: gathered from global memory (L2)
    const Rsrc tri_r = make_rsrc(reinterpret_cast<const uint8_t *>(P.tri));
    const Rsrc mesh_r = make_rsrc(reinterpret_cast<const uint8_t *>(P.mesh));
    const uint32_t cap4 = (uint32_t)P.cap * 4u;   // bytes per field array
    float4 *l_mat = lds;
    float4 *l_rx = l_mat + 4u * HRT_NUM_MATERIALS;
    uint32_t *l_wcnt = reinterpret_cast<uint32_t *>(l_rx + P.num_rx);
    float4 *l_trin = reinterpret_cast<float4 *>(l_wcnt + 8);   // [T]: n.xyz, mesh id (bits)
    float4 *l_mesh = l_trin + P.num_tri;                         // [M]: velocity, material (bits)
    {
        const float4 *g_mat = reinterpret_cast<const float4 *>(P.mat);
        for (uint32_t k = tid; k < 4u * HRT_NUM_MATERIALS; k += HRT_BLOCK) l_mat[k] = g_mat[k];
        for (uint32_t k = tid; k < P.num_rx; k += HRT_BLOCK)
            l_rx[k] = make_float4(P.rx_pos[3 * k], P.rx_pos[3 * k + 1], P.rx_pos[3 * k + 2], 0.f);
        if constexpr (NLDS) {
            for (uint32_t k = tid; k < P.num_tri; k += HRT_BLOCK) {
                const float *row = P.tri + (size_t)k * HRT_TRI_FLOATS;
                l_trin[k] = make_float4(row[9], row[10], row[11], row[19]);
            }
            const float4 *g_mesh = reinterpret_cast<const float4 *>(P.mesh);
            for (uint32_t k = tid; k < P.num_mesh; k += HRT_BLOCK) l_mesh[k] = g_mesh[k];
        }
    }
    __syncthreads();
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    // normal and mesh id of table row j; row of the mesh table
    auto tri_normal = [&](uint32_t j, uint32_t &mesh) -> F3 {
        if constexpr (NLDS) {
            const float4 q = l_trin[j];
            mesh = __float_as_uint(q.w);
            return {q.x, q.y, q.z};
        } else {
            mesh = ldu(tri_r, 0u, j * (HRT_TRI_FLOATS * 4u) + 76u);
            return gather3(tri_r, HRT_TRI_FLOATS * 4u, j, 36u);
        }
    };
    auto mesh_row = [&](uint32_t m) -> float4 {
        if constexpr (NLDS) return l_mesh[m];
        else return gather4(mesh_r, HRT_MESH_FLOATS * 4u, m, 0u);
    };

    for (uint64_t base = (uint64_t)blockIdx.x * HRT_BLOCK; base < n_in;
         base += (uint64_t)gridDim.x * HRT_BLOCK) {
        const uint32_t i = (uint32_t)base + tid;
        const uint32_t i4 = i * 4u;
        const bool valid = i < n_in;

        // ---- ray state ----
        uint32_t ray = 0, htri = 0;
        float theta = 0.f, fs0 = 0.f, tau = 0.f;
        F3 o = {0.f, 0.f, 0.f}, d = {0.f, 0.f, 1.f};
        float a0 = 1.f, a1 = 0.f, a2 = 1.f, a3 = 0.f;
        // the bounce's own trace result: requested in front of the records' stores.  (Bounce only -- the records are
        // hrt_records_kernel's --: the state of an entry is only loaded if it hit: 40-55 % of the entries of C3 did
        // not, and the kernel is bound by its traffic.)
        uint32_t ptri = HRT_NO_HIT;
        float pt = 0.f;
        if (do_trace && valid) {
            ptri = ldu(res_blk(P, P.num_rx), 0u, i4);
            pt = ldf(res_blk(P, P.num_rx), cap4, i4);
        }
        const bool need_state = valid && !(P.records_done && ptri == HRT_NO_HIT);
        if (need_state) {
            if (first) {
                // src/compute_paths.c:452-466 + the launch Doppler term :494-500
                uint32_t tx;
                launch_ray(P, i, ray, o, d, tx);
                const F3 tv = {P.tx_vel[3 * tx], P.tx_vel[3 * tx + 1], P.tx_vel[3 * tx + 2]};
                fs0 = dot3(tv, d) * P.dop_mult;
            } else {
                const uint32_t pb = b - 1;
                ray = __float_as_uint(ldf(hit_blk(P, pb), H_RAY * cap4, i4));
                htri = __float_as_uint(ldf(hit_blk(P, pb), H_TRI * cap4, i4));
                theta = ldf(hit_blk(P, pb), H_THETA * cap4, i4);
                fs0 = ldf(hit_blk(P, pb), H_FS0 * cap4, i4);
                o = {ldf(hit_blk(P, pb), H_OX * cap4, i4), ldf(hit_blk(P, pb), H_OY * cap4, i4),
                     ldf(hit_blk(P, pb), H_OZ * cap4, i4)};
                d = {ldf(hit_blk(P, pb), H_DX * cap4, i4), ldf(hit_blk(P, pb), H_DY * cap4, i4),
                     ldf(hit_blk(P, pb), H_DZ * cap4, i4)};
                a0 = ldf(hit_blk(P, pb), H_A0 * cap4, i4);
                a1 = ldf(hit_blk(P, pb), H_A1 * cap4, i4);
                a2 = ldf(hit_blk(P, pb), H_A2 * cap4, i4);
                a3 = ldf(hit_blk(P, pb), H_A3 * cap4, i4);
                tau = ldf(hit_blk(P, pb), H_TAU * cap4, i4);
            }
        }

        // ---- scatter records of bounce b-1 (unless hrt_records_kernel wrote them) ----
        if (!first && !P.records_done) {
            const uint32_t pb = b - 1;
            F3 n = {0.f, 0.f, 1.f}, mvel = {0.f, 0.f, 0.f};
            float mat_s = 0.f, mat_alpha = 1.f;
            if (valid) {
                uint32_t mesh;
                n = tri_normal(htri, mesh);
                const float4 mm = mesh_row(mesh);
                mvel = {mm.x, mm.y, mm.z};
                const float4 m3 = l_mat[4u * __float_as_uint(mm.w) + 3u];
                mat_s = m3.x;
                mat_alpha = m3.y;
            }
            for (uint32_t rx0 = 0; rx0 < P.num_rx; rx0 += 4u) {
                // the shadow results of four RXs (see the trace kernel): one word or half word each
                uint32_t code[4];
#pragma unroll
                for (uint32_t j = 0; j < 4u; ++j) {
                    code[j] = 0u;
                    if (valid && rx0 + j < P.num_rx) {
                        if (P.num_tri < HRT_HALF_RESULTS_MAX)
                            code[j] = (uint32_t)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(res_blk(P, rx0 + j), (int)(i * 2u), 0, 0);
                        else
                            code[j] = ldu(res_blk(P, rx0 + j), 0u, i4);
                    }
                }
#pragma unroll
                for (uint32_t j = 0; j < 4u; ++j) {
                    const uint32_t rx = rx0 + j;
                    if (rx >= P.num_rx) break;   // wave-uniform
                    bool unblocked = false;
                    if (valid) {
                        const float4 rp = l_rx[rx];
                        float d2rx;
                        const F3 w = shadow_dir(o, {rp.x, rp.y, rp.z}, d2rx);
                        uint32_t stri;
                        bool near1;
                        if (P.num_tri < HRT_HALF_RESULTS_MAX) {
                            stri = code[j] & 0x7fffu;
                            near1 = (code[j] >> 15) != 0u;
                            if (stri == 0x7fffu) stri = HRT_NO_HIT;
                        } else {
                            stri = code[j] & 0x7fffffffu;
                            near1 = (code[j] >> 31) != 0u;
                            if (stri == 0x7fffffffu) stri = HRT_NO_HIT;
                        }
                        if (stri != HRT_NO_HIT && stri >= P.num_tri) {   // cannot happen; never fault
                            atomicOr(const_cast<uint32_t *>(&counts[P.num_bounces + 1]), 1u);
                            stri = HRT_NO_HIT;
                        }
                        if (stri != HRT_NO_HIT) {
                            uint32_t smesh;
                            theta = incidence_angle(tri_normal(stri, smesh), w);
                        }
                        // (issuing a record's stores one record late -- behind the next record's calls, whose
                        // callees begin with s_waitcnt vmcnt(0) -- was measured: 10 more registers, +17 %)
                        if (stri != HRT_NO_HIT && near1) {
                            stf(rec_blk(P, pb, rx), R_A0 * cap4, i4, 0.f);
                            stf(rec_blk(P, pb, rx), R_A1 * cap4, i4, 0.f);
                            stf(rec_blk(P, pb, rx), R_A2 * cap4, i4, 0.f);
                            stf(rec_blk(P, pb, rx), R_A3 * cap4, i4, 0.f);
                            stf(rec_blk(P, pb, rx), R_TAU * cap4, i4, 0.f);
                        } else {
                            unblocked = true;
                            const float th_s = acos_f_ool(dot3(w, n));
                            const float4 S = scatter_pattern(mat_s, mat_alpha, th_s, theta);
                            float o0 = a0 * S.x - a1 * S.y;
                            float o1 = a0 * S.y + a1 * S.x;
                            float o2 = a2 * S.z - a3 * S.w;
                            float o3 = a2 * S.w + a3 * S.z;
                            float f2 = P.fsl_mult * d2rx;
                            f2 *= f2;
                            if (f2 > 1.f) { o0 /= f2; o1 /= f2; o2 /= f2; o3 /= f2; }
                            stf(rec_blk(P, pb, rx), R_A0 * cap4, i4, o0);
                            stf(rec_blk(P, pb, rx), R_A1 * cap4, i4, o1);
                            stf(rec_blk(P, pb, rx), R_A2 * cap4, i4, o2);
                            stf(rec_blk(P, pb, rx), R_A3 * cap4, i4, o3);
                            stf(rec_blk(P, pb, rx), R_TAU * cap4, i4, tau + d2rx / kC);
                            stf(rec_blk(P, pb, rx), R_DX * cap4, i4, -w.x);
                            stf(rec_blk(P, pb, rx), R_DY * cap4, i4, -w.y);
                            stf(rec_blk(P, pb, rx), R_DZ * cap4, i4, -w.z);
                            stf(rec_blk(P, pb, rx), R_DFS * cap4, i4, dot3(sub3(w, d), mvel) * P.dop_mult);
                        }
                    }
                    const unsigned long long m = __ballot(unblocked);
                    if (lane == 0 && valid) mask_words(P, pb, rx)[i >> 6] = m;
                }
            }
        }

        // ---- the bounce (src/compute_paths.c:611-659) ----
        if (do_trace) {
            bool hit = false;
            uint32_t ntri = 0;
            float nth = 0.f;
            if (valid) {
                if (ptri != HRT_NO_HIT && ptri >= P.num_tri) {       // cannot happen; never fault
                    atomicOr(const_cast<uint32_t *>(&counts[P.num_bounces + 1]), 2u);
                    ptri = HRT_NO_HIT;
                }
                if (ptri != HRT_NO_HIT) {
                    hit = true;
                    ntri = ptri;
                    uint32_t mesh;
                    const F3 n = tri_normal(ptri, mesh);
                    nth = incidence_angle(n, d);
                    const uint32_t mat = __float_as_uint(mesh_row(mesh).w);
                    float4 R = fresnel(l_mat[4u * mat], l_mat[4u * mat + 1u], l_mat[4u * mat + 2u], nth);
                    float fsl = P.fsl_mult * pt;
                    fsl *= fsl;
                    if (fsl > 1.f) { R.x /= fsl; R.y /= fsl; R.z /= fsl; R.w /= fsl; }
                    const float b0 = a0 * R.x - a1 * R.y;
                    const float b1 = a0 * R.y + a1 * R.x;
                    const float b2 = a2 * R.z - a3 * R.w;
                    const float b3 = a2 * R.w + a3 * R.z;
                    a0 = b0; a1 = b1; a2 = b2; a3 = b3;
                    tau += pt / kC;
                    o = add3(mul3(d, pt), o);
                    const float dn = dot3(d, n);
                    d = sub3(d, mul3(n, 2.f * dn));
                    o = add3(o, mul3(d, 1e-4f));
                }
            }
            // STABLE compaction: survivors of this 256-entry chunk, in input order, go to
            // offset(chunk) + rank of the next live list, offset(chunk) = survivors of all earlier
            // chunks = sum of the earlier super-chunks' counts + sum of the earlier chunks' counts
            // of this super-chunk (both written by the trace kernel of this launch): a few loads
            // per thread and one workgroup reduction, no scan pass -- and the next list keeps the
            // (coherent) order of this one, without a staging copy.
            const uint32_t chunk = (uint32_t)(base / HRT_BLOCK);
            const uint32_t sup = chunk >> HRT_SUPER_SHIFT;
            uint32_t part = 0;
            {
                const uint32_t *scnt = reinterpret_cast<const uint32_t *>(P.ws + P.off_super_cnt) +
                                       (uint64_t)b * P.num_super;
                const uint32_t *ccnt = reinterpret_cast<const uint32_t *>(P.ws + P.off_chunk_cnt);
                // chunk 0 also owes the total (all super-chunks): counts[b+1]
                const uint32_t n_sup =
                    (chunk == 0) ? ((n_in + HRT_BLOCK - 1) / HRT_BLOCK + HRT_SUPER_CHUNKS - 1u) >> HRT_SUPER_SHIFT : sup;
                for (uint32_t q = tid; q < n_sup; q += HRT_BLOCK) part += scnt[q];
                if (P.cnt_per_wave) {   // (the fine walk: one word per wave of a chunk)
                    const uint32_t cw = ((sup << HRT_SUPER_SHIFT) << 2) + tid;
                    if (chunk != 0 && tid < 4u * HRT_SUPER_CHUNKS && (cw >> 2) < chunk) part += ccnt[cw];
                } else {
                    const uint32_t c = (sup << HRT_SUPER_SHIFT) + tid;
                    if (chunk != 0 && tid < HRT_SUPER_CHUNKS && c < chunk) part += ccnt[c];
                }
            }
            part = wave_sum_u32(part);
            const unsigned long long m = __ballot(hit);
            if (lane == 0) {
                l_wcnt[wave] = (uint32_t)__popcll(m);
                l_wcnt[4 + wave] = part;
            }
            __syncthreads();
            uint32_t before = 0;
#pragma unroll
            for (uint32_t w = 0; w < HRT_BLOCK / 64u; ++w) before += (w < wave) ? l_wcnt[w] : 0u;
            uint32_t chunk_off = l_wcnt[4] + l_wcnt[5] + l_wcnt[6] + l_wcnt[7];
            __syncthreads();
            if (chunk == 0) {
                if (tid == 0) const_cast<uint32_t *>(counts)[b + 1] = chunk_off;
                chunk_off = 0;
            }
            if (hit) {
                const uint32_t k4 = (chunk_off + before + lane_prefix(m)) * 4u;
                // (the block the survivors go to: hit block b, or the scratch the re-sort reads from)
                stf(hit_out(P, b), H_RAY * cap4, k4, __uint_as_float(ray));
                stf(hit_out(P, b), H_TRI * cap4, k4, __uint_as_float(ntri));
                stf(hit_out(P, b), H_THETA * cap4, k4, nth);
                stf(hit_out(P, b), H_FS0 * cap4, k4, fs0);
                stf(hit_out(P, b), H_OX * cap4, k4, o.x);
                stf(hit_out(P, b), H_OY * cap4, k4, o.y);
                stf(hit_out(P, b), H_OZ * cap4, k4, o.z);
                stf(hit_out(P, b), H_DX * cap4, k4, d.x);
                stf(hit_out(P, b), H_DY * cap4, k4, d.y);
                stf(hit_out(P, b), H_DZ * cap4, k4, d.z);
                stf(hit_out(P, b), H_A0 * cap4, k4, a0);
                stf(hit_out(P, b), H_A1 * cap4, k4, a1);
                stf(hit_out(P, b), H_A2 * cap4, k4, a2);
                stf(hit_out(P, b), H_A3 * cap4, k4, a3);
                stf(hit_out(P, b), H_TAU * cap4, k4, tau);
            }
        }
    }
}

// ===================================================================================
// IMAGE kernel (patch tables): the primary rays of launch 1.  They left a TX and were mirrored once, so every
// one of them leaves the image of its TX in the plane of the triangle it sits on: the lane locates its patch,
// checks that its line does pass the image (patch_locate), and the wave walks the union of the lanes'
// (image apex, patch) masks -- no packet is formed, and a wave that mixes surfaces pays for the union of a few
// small sets.  Results and survivor counts as hrt_trace_kernel writes them for the shade kernel.
// LDS: [T x 5 float4 rows][4 u32].
// ===================================================================================
__global__ __launch_bounds__(HRT_BLOCK, 8) void hrt_image_kernel(const hrt_kparams P, const uint32_t b)
{
    extern __shared__ float4 lds[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t *counts = reinterpret_cast<const uint32_t *>(P.ws + P.off_counts);
    if (counts[P.num_bounces + 1] & kErrVoid) return;   // a fused launch of this trace timed out: the host redoes the step
    const uint32_t n_in = counts[b];
    const uint32_t n_chunks = (n_in + HRT_BLOCK - 1) / HRT_BLOCK;
    if (blockIdx.x >= n_chunks) return;
    const uint32_t T = P.num_tri;
    const uint32_t cap4 = (uint32_t)P.cap * 4u;
    float4 *l_tri = lds;
    uint32_t *l_wcnt = reinterpret_cast<uint32_t *>(lds + HRT_ROW * T);
    {
        const float4 *g_tri = reinterpret_cast<const float4 *>(P.tri);
        for (uint32_t k = tid; k < HRT_ROW * T; k += HRT_BLOCK) l_tri[k] = g_tri[k];
    }
    __syncthreads();
    const float4 *tri = l_tri;
    const uint32_t pb = b - 1;
    for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        const uint32_t i = chunk * HRT_BLOCK + tid;
        const uint32_t i4 = i * 4u;
        const bool valid = i < n_in;
        F3 o = {0.f, 0.f, 0.f}, d = {0.f, 0.f, 1.f};
        uint32_t tx = 0u;
        PatchRef ref = {false, 0u};
        if (valid) {
            o = {ldf(hit_blk(P, pb), H_OX * cap4, i4), ldf(hit_blk(P, pb), H_OY * cap4, i4), ldf(hit_blk(P, pb), H_OZ * cap4, i4)};
            d = {ldf(hit_blk(P, pb), H_DX * cap4, i4), ldf(hit_blk(P, pb), H_DY * cap4, i4), ldf(hit_blk(P, pb), H_DZ * cap4, i4)};
            const uint32_t htri = ldu(hit_blk(P, pb), H_TRI * cap4, i4);
            if (P.num_tx != 1u) tx = min(ldu(hit_blk(P, pb), H_RAY * cap4, i4) / P.num_local, P.num_tx - 1u);
            // the apex: the image of the lane's own TX in the plane of the triangle it left
            ref = patch_locate(tri, P.patch, T, htri, o, true, {P.tx_pos[3 * tx], P.tx_pos[3 * tx + 1], P.tx_pos[3 * tx + 2]}, d);
        }
        const Hit h = closest_hit_patch(tri, P.acc.orig, P.patch, ref, P.num_rx + tx, T, o, d, valid, lane, 1);
        if (valid) {
            stu(res_blk(P, P.num_rx), 0u, i4, h.tri);
            stf(res_blk(P, P.num_rx), cap4, i4, h.t);
        }
        // survivors of the chunk and of its super-chunk (hrt_trace_kernel's protocol: the shade kernel sums them)
        const unsigned long long hm = __ballot(valid && h.tri != HRT_NO_HIT);
        if (lane == 0) l_wcnt[tid >> 6] = (uint32_t)__popcll(hm);
        __syncthreads();
        if (tid == 0) {
            const uint32_t c = l_wcnt[0] + l_wcnt[1] + l_wcnt[2] + l_wcnt[3];
            reinterpret_cast<uint32_t *>(P.ws + P.off_chunk_cnt)[chunk] = c;
            atomicAdd(reinterpret_cast<uint32_t *>(P.ws + P.off_super_cnt) + (uint64_t)b * P.num_super + (chunk >> HRT_SUPER_SHIFT), c);
        }
        __syncthreads();
    }
}

#ifndef HRT_RECORDS_PREFETCH
#define HRT_RECORDS_PREFETCH 1   /* masks of RX r + 1 requested before the walk of RX r: C3 1.125 -> 1.112 ms at 6 waves (5 waves: 1.155) */
#endif
#ifndef HRT_RECORDS_WAVES
#define HRT_RECORDS_WAVES 6
#endif
// LDS: [T x 5 float4 rows][num_rx RX pos][17 x 4 float4 materials]
__global__ __launch_bounds__(HRT_BLOCK, HRT_RECORDS_WAVES) void hrt_records_kernel(const hrt_kparams P, const uint32_t b)
{
    extern __shared__ float4 lds[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t *counts = reinterpret_cast<const uint32_t *>(P.ws + P.off_counts);
    if (counts[P.num_bounces + 1] & kErrVoid) return;   // a fused launch of this trace timed out: the host redoes the step
    const uint32_t n_in = counts[b];
    const uint32_t n_chunks = (n_in + HRT_BLOCK - 1) / HRT_BLOCK;
    if (blockIdx.x >= n_chunks) return;
    const uint32_t T = P.num_tri;
    const uint32_t cap4 = (uint32_t)P.cap * 4u;
    const Rsrc mesh_r = make_rsrc(reinterpret_cast<const uint8_t *>(P.mesh));
    float4 *l_tri = lds;
    float4 *l_rx = lds + HRT_ROW * T;
    float4 *l_mat = l_rx + P.num_rx;
    // (the reference-order index of every row, for the tie rule: in LDS, so that the exact stage of the walk has no
    // global load -- whose s_waitcnt vmcnt(0) would also wait for the masks requested ahead)
    uint32_t *l_orig = reinterpret_cast<uint32_t *>(l_mat + 4u * HRT_NUM_MATERIALS);
    {
        const float4 *g_tri = reinterpret_cast<const float4 *>(P.tri);
        const float4 *g_mat = reinterpret_cast<const float4 *>(P.mat);
        for (uint32_t k = tid; k < HRT_ROW * T; k += HRT_BLOCK) l_tri[k] = g_tri[k];
        for (uint32_t k = tid; k < P.num_rx; k += HRT_BLOCK)
            l_rx[k] = make_float4(P.rx_pos[3 * k], P.rx_pos[3 * k + 1], P.rx_pos[3 * k + 2], 0.f);
        for (uint32_t k = tid; k < 4u * HRT_NUM_MATERIALS; k += HRT_BLOCK) l_mat[k] = g_mat[k];
        for (uint32_t k = tid; k < T; k += HRT_BLOCK) l_orig[k] = P.acc.orig[k];
    }
    __syncthreads();
    const float4 *tri = l_tri;
    const uint32_t pb = b - 1;
    for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        const uint32_t i = chunk * HRT_BLOCK + tid;
        const uint32_t i4 = i * 4u;
        const bool valid = i < n_in;
        F3 o = {0.f, 0.f, 0.f}, d = {0.f, 0.f, 1.f}, n = {0.f, 0.f, 1.f}, mvel = {0.f, 0.f, 0.f};
        float theta = 0.f, tau = 0.f, a0 = 1.f, a1 = 0.f, a2 = 1.f, a3 = 0.f, mat_s = 0.f, mat_alpha = 1.f;
        PatchRef ref = {false, 0u};
        if (valid) {
            const uint32_t htri = ldu(hit_blk(P, pb), H_TRI * cap4, i4);
            theta = ldf(hit_blk(P, pb), H_THETA * cap4, i4);
            o = {ldf(hit_blk(P, pb), H_OX * cap4, i4), ldf(hit_blk(P, pb), H_OY * cap4, i4),
                 ldf(hit_blk(P, pb), H_OZ * cap4, i4)};
            d = {ldf(hit_blk(P, pb), H_DX * cap4, i4), ldf(hit_blk(P, pb), H_DY * cap4, i4),
                 ldf(hit_blk(P, pb), H_DZ * cap4, i4)};
            a0 = ldf(hit_blk(P, pb), H_A0 * cap4, i4);
            a1 = ldf(hit_blk(P, pb), H_A1 * cap4, i4);
            a2 = ldf(hit_blk(P, pb), H_A2 * cap4, i4);
            a3 = ldf(hit_blk(P, pb), H_A3 * cap4, i4);
            tau = ldf(hit_blk(P, pb), H_TAU * cap4, i4);
            ref = patch_locate(tri, P.patch, T, htri, o, false, o, o);
            if (htri < T) {   // (always: the list holds rows of the table; never fault)
                const float4 q2 = tri[HRT_ROW * htri + 2];
                n = {q2.y, q2.z, q2.w};
                const float4 mm = gather4(mesh_r, HRT_MESH_FLOATS * 4u, __float_as_uint(tri[HRT_ROW * htri + 4].w), 0u);
                mvel = {mm.x, mm.y, mm.z};
                const float4 m3 = l_mat[4u * __float_as_uint(mm.w) + 3u];
                mat_s = m3.x;
                mat_alpha = m3.y;
            }
        }
#if HRT_RECORDS_PREFETCH
        uint32_t wnext[8];   // the masks of the NEXT RX are requested before this RX's walk (a global gather: ~1-2 us)
        patch_load(P.patch, ref, 0u, T, valid, wnext);
#endif
        for (uint32_t rx = 0; rx < P.num_rx; ++rx) {
            const float4 rp = l_rx[rx];
            float d2rx;
            F3 w = shadow_dir(o, {rp.x, rp.y, rp.z}, d2rx);
            if (!valid) w = {0.f, 0.f, 1.f};
#if HRT_RECORDS_PREFETCH
            uint32_t wcur[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) wcur[q] = wnext[q];
            if (rx + 1u < P.num_rx) patch_load(P.patch, ref, rx + 1u, T, valid, wnext);
            const Hit h = closest_hit_words(tri, l_orig, wcur, T, o, w, valid, lane, 2);
#else
            const Hit h = closest_hit_patch(tri, l_orig, P.patch, ref, rx, T, o, w, valid, lane, 2);
#endif
            bool unblocked = false;
            if (valid) {
                if (h.tri != HRT_NO_HIT) {   // quirk Q7: any shadow hit, at any distance, overwrites theta
                    const float4 q2 = tri[HRT_ROW * h.tri + 2];
                    theta = incidence_angle({q2.y, q2.z, q2.w}, w);
                }
                if (h.tri != HRT_NO_HIT && h.t <= 1.f) {   // quirk Q6: blocked within one metre
                    stf(rec_blk(P, pb, rx), R_A0 * cap4, i4, 0.f);
                    stf(rec_blk(P, pb, rx), R_A1 * cap4, i4, 0.f);
                    stf(rec_blk(P, pb, rx), R_A2 * cap4, i4, 0.f);
                    stf(rec_blk(P, pb, rx), R_A3 * cap4, i4, 0.f);
                    stf(rec_blk(P, pb, rx), R_TAU * cap4, i4, 0.f);
                } else {
                    unblocked = true;
                    const float th_s = acos_f_ool(dot3(w, n));
                    const float4 S = scatter_pattern(mat_s, mat_alpha, th_s, theta);
                    float o0 = a0 * S.x - a1 * S.y;
                    float o1 = a0 * S.y + a1 * S.x;
                    float o2 = a2 * S.z - a3 * S.w;
                    float o3 = a2 * S.w + a3 * S.z;
                    float f2 = P.fsl_mult * d2rx;
                    f2 *= f2;
                    if (f2 > 1.f) { o0 /= f2; o1 /= f2; o2 /= f2; o3 /= f2; }
                    stf(rec_blk(P, pb, rx), R_A0 * cap4, i4, o0);
                    stf(rec_blk(P, pb, rx), R_A1 * cap4, i4, o1);
                    stf(rec_blk(P, pb, rx), R_A2 * cap4, i4, o2);
                    stf(rec_blk(P, pb, rx), R_A3 * cap4, i4, o3);
                    stf(rec_blk(P, pb, rx), R_TAU * cap4, i4, tau + d2rx / kC);
                    stf(rec_blk(P, pb, rx), R_DX * cap4, i4, -w.x);
                    stf(rec_blk(P, pb, rx), R_DY * cap4, i4, -w.y);
                    stf(rec_blk(P, pb, rx), R_DZ * cap4, i4, -w.z);
                    stf(rec_blk(P, pb, rx), R_DFS * cap4, i4, dot3(sub3(w, d), mvel) * P.dop_mult);
                }
            }
            const unsigned long long m = __ballot(unblocked);
            if (lane == 0 && valid) mask_words(P, pb, rx)[i >> 6] = m;
        }
    }
}

// ===================================================================================
// FUSED kernels: one launch = ONE kernel (trace + shading + stable compaction).
//
// The split into a geometry kernel and a shading kernel pays where the geometry is heavy (a table
// of hundreds of triangles: the trace kernel then runs at 7-8 waves/SIMD with nothing but a ray in
// its registers).  Where it is light it only costs traffic and launches: at launch 0 (no shadow
// rays, state generated in registers) the pair reads the 12-byte direction twice and moves 8 bytes
// of trace result per ray through HBM to hand a triangle index from one kernel to the next; on a
// table of one culling round the same holds for every launch.  Fused, a launch reads its input
// once and writes records and survivors once.
//
// What made the split necessary was the STABLE compaction -- a survivor's place in the next live
// list is (survivors of all earlier chunks) + (rank in its chunk), and the first term needs every
// earlier chunk's count.  Here it comes from SUMS ON THREE LEVELS, one workgroup per 256-entry chunk:
//   * as soon as a chunk knows its count c it stores (DONE | c) in its own status word;
//   * the prefix of chunk i = sum of the supergroups (4096 chunks) before its own + sum of the groups
//     (64 chunks) before its own inside its supergroup + sum of the chunks before it inside its
//     group: at most 64 (128) words per level, all loaded at once by one wave, polled until DONE;
//   * the LAST chunk of a group polls all other chunk words of its group anyway: when they are done
//     it stores (DONE | group total) in the group's word; the last chunk of a supergroup does the
//     same with the group words for the supergroup's word.  One writer per word: plain stores.
// So a chunk waits for the traces of the chunks in front of it (the stragglers among them) plus one
// round trip (three near a supergroup boundary), and nothing is a chain.  Tried first: a chained
// scan ("decoupled look-back") walks ~30 dependent rounds back through the ~1 800 workgroups in
// flight (1.6x slower than two kernels), and sums kept by atomic adds into per-group / per-
// supergroup words serialise at ~60 ns per add on the hot words (C4: 4 ms).
// All words are written and read with relaxed agent-scope stores and loads (sc1: they bypass the
// per-XCD L2), so no fence and no L2 write-back are involved -- the fence is what made a "last
// workgroup scans" hand-off cost 0.5 ms on this 8-XCD part.  Progress: workgroups are dispatched in
// index order, a chunk only ever waits for lower-numbered chunks, and those never wait for it; one
// chunk per workgroup, no grid-stride loop (a resident workgroup must never wait for a chunk whose
// workgroup still needs a slot).  The shading of a chunk's hits sits between "publish" and "sum",
// which hides most of the wait.
// ===================================================================================
__device__ __forceinline__ void los_pairs(const hrt_kparams &P, const uint32_t block);   // (below)

constexpr uint32_t kLbDone = 1u << 31;
// words of launch b (u32, from the launch's base): [nc chunk words][nc/64 group words][128 supergroup
// words]; nc = lb_chunks (a multiple of 64); the host sizes and zeroes them (hrt_layout_query).
// (128 supergroups: 128 * 4096 chunks = 1.3e8 entries, beyond the 7.1e7 a shard can hold)

__device__ __forceinline__ uint32_t lb_load(const uint32_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void lb_store(uint32_t *p, uint32_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct LbWords { uint32_t *chunk, *group, *super; };
__device__ __forceinline__ LbWords lb_words(const hrt_kparams &P, uint32_t b)
{
    LbWords W;
    W.chunk = reinterpret_cast<uint32_t *>(P.ws + P.off_lb) + (uint64_t)b * P.lb_stride;
    W.group = W.chunk + P.lb_chunks;
    W.super = W.group + P.lb_chunks / 64u;
    return W;
}

// Exclusive prefix of `chunk` (survivors of all earlier chunks); c = the chunk's own count (already
// published in its word).  Called by all 64 lanes of ONE wave (uniform control flow); the result is
// wave-uniform.
// Bounded: the chunks in front are waited for at most `max_polls` polls (hrt_ktune.lb_max_polls: about 10 ms; the wait is
// microseconds when the kernel owns the GPU).  A chunk only ever waits for lower-numbered chunks, and those have
// been dispatched -- per XCD: the XCDs dispatch their shares of a grid independently, so a resident workgroup can
// spin on one that still waits for a slot on another XCD.  Alone on the GPU that slot comes; when several fused
// kernels (four processes sharing one GPU) fill each other's slots with spinning workgroups it does not until a
// time slice ends: C4 took 5.8 s per step instead of 0.2 ms.  On a timeout the launch is declared VOID: the
// error word of the trace (counts[nb + 1]) gets kErrFuseTimeout, every spinning workgroup that sees it leaves,
// every later kernel of the trace returns at once, and the host -- which sees the bit in the counts it reads
// anyway, and a flag word in pinned memory without any synchronisation -- runs the step again as two kernels
// per launch and keeps fusion off from then on.  (Tickets drawn at workgroup start make the order exact, but a
// returning atomic on one address per workgroup cost 10 % of C4's step and 30 % of C2's: profiles/HISTORY.md r4.)
constexpr uint32_t kLbAbort = 0xffffffffu;
// the step is void: the bit in the trace's error word, and the host's flag word of that kind (host_flag[0]: a
// fused launch, [1]: the chain kernel) -- pinned memory, seen by the host without synchronising
__device__ __forceinline__ void give_up(uint32_t *err_word, uint32_t *host_flag, uint32_t err_bit)
{
    atomicOr(err_word, err_bit);
    if (host_flag)
        __hip_atomic_store(host_flag + (err_bit == kErrChainTimeout ? 1 : 0), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ uint32_t lb_exclusive(const LbWords &W, uint32_t chunk, uint32_t c, uint32_t lane,
                                                 uint32_t *err_word, uint32_t *host_flag, const uint32_t max_polls,
                                                 const uint32_t err_bit = kErrFuseTimeout)
{
    uint32_t polls = 0u;
    const uint32_t g = chunk >> 6, sg = chunk >> 12;
    const bool want_c = (g << 6) + lane < chunk, want_g = (sg << 6) + lane < g;
    const bool want_s0 = lane < sg, want_s1 = lane + 64u < sg;
    bool g_owed = (chunk & 63u) == 63u, s_owed = (chunk & 4095u) == 4095u;   // this chunk closes its group / supergroup
    for (;;) {
        // (all four loads are in flight together)
        const uint32_t cw = want_c ? lb_load(W.chunk + (g << 6) + lane) : kLbDone;
        const uint32_t gw = want_g ? lb_load(W.group + (sg << 6) + lane) : kLbDone;
        const uint32_t s0 = want_s0 ? lb_load(W.super + lane) : kLbDone;
        const uint32_t s1 = want_s1 ? lb_load(W.super + lane + 64u) : kLbDone;
        const bool ok_c = HRT_BALLOT(!(cw & kLbDone)) == 0ull, ok_g = HRT_BALLOT(!(gw & kLbDone)) == 0ull;
        const bool ok_s = HRT_BALLOT(!(s0 & s1 & kLbDone)) == 0ull;
        uint32_t sum_c = 0u, sum_g = 0u;
        // (sum_c is formed whenever it is consumed below: by the group word, by the supergroup word --
        // which may be owed while super[sg - 1] is still missing -- or by the result)
        if (ok_c && (g_owed || (ok_g && (s_owed || ok_s)))) sum_c = wave_sum_u32(cw & ~kLbDone);
        if (ok_c && g_owed) {
            if (lane == 0) lb_store(W.group + g, kLbDone | (sum_c + c));
            g_owed = false;
        }
        if (ok_c && ok_g && (s_owed || ok_s)) sum_g = wave_sum_u32(gw & ~kLbDone);
        if (ok_c && ok_g && s_owed) {
            if (lane == 0) lb_store(W.super + sg, kLbDone | (sum_g + sum_c + c));
            s_owed = false;
        }
        if (ok_c && ok_g && ok_s) return sum_c + sum_g + wave_sum_u32((s0 & ~kLbDone) + (s1 & ~kLbDone));
        if (lb_load(err_word) & kErrVoid) return kLbAbort;        // somebody gave up: the launch is void
        if (++polls > max_polls) {
            if (lane == 0) give_up(err_word, host_flag, err_bit);
            return kLbAbort;
        }
        __builtin_amdgcn_s_sleep(4);
    }
}

// LDS image shared by the fused kernels (hrt_hip_launch_fused sizes it):
// [T x 5 float4 rows | T float2 guard pairs, padded to 16 B | 2 float4 per leaf] (only if staged)
// [num_rx float4 RX pos][4 waves x 16 u64 masks][4 waves x 32 float4 per-wave scratch][64 u32]
// [17 x 4 float4 materials][64 float4 TX pos]
constexpr uint32_t kLdsTx = 64u;
struct FusedLds {
    float4 *tri, *leaf, *rx, *wleaf, *mat, *tx;
    float2 *tg;
    unsigned long long *mask;
    uint32_t *wcnt;
};
template <bool TRI_IN_LDS, bool CAND_BUF>
__device__ __forceinline__ FusedLds fused_lds(float4 *lds, uint32_t T, uint32_t n_leaf, uint32_t num_rx, uint32_t wave)
{
    FusedLds L;
    L.tri = lds;
    L.tg = reinterpret_cast<float2 *>(lds + HRT_ROW * T);
    L.leaf = lds + HRT_ROW * T + (T + 1u) / 2u;
    L.rx = TRI_IN_LDS ? L.leaf + 2u * n_leaf : lds;
    unsigned long long *m0 = reinterpret_cast<unsigned long long *>(L.rx + num_rx);
    L.mask = m0 + wave * kMaskRounds;
    float4 *w0 = reinterpret_cast<float4 *>(m0 + (HRT_BLOCK / 64u) * kMaskRounds);
    L.wleaf = w0 + wave * wave_scratch4(CAND_BUF);   // (per-wave scratch, + the candidate buffer of the fine walk)
    L.wcnt = reinterpret_cast<uint32_t *>(w0 + (HRT_BLOCK / 64u) * wave_scratch4(CAND_BUF));
    L.mat = reinterpret_cast<float4 *>(L.wcnt + 64);
    L.tx = L.mat + 4u * HRT_NUM_MATERIALS;   // the first kLdsTx TX positions (launch 0)
    return L;
}

// The bounce itself for a ray that hit triangle `ptri` at distance `pt` (src/compute_paths.c:
// 611-659): incidence angle, Fresnel, free-space loss, delay, reflection.  Same sequence as the
// shade kernel's, in two steps: bounce_fetch requests what comes from memory (the triangle's normal,
// its mesh's material: `tri` is the table in LDS or in global memory), bounce_apply computes -- the
// fused kernel issues the fetches of all its packets in front of the wait for its prefix, so that
// the shading loop behind it has no load in front of its stores (loads and stores share vmcnt).
template <typename TriPtr>
__device__ __forceinline__ void bounce_fetch(TriPtr tri, Rsrc mesh_r, uint32_t ptri, F3 &n, uint32_t &mat)
{
    const float4 q2 = tri[HRT_ROW * ptri + 2];
    n = {q2.y, q2.z, q2.w};
    const uint32_t mesh = __float_as_uint(tri[HRT_ROW * ptri + 4].w);
    mat = ldu(mesh_r, 0u, mesh * (HRT_MESH_FLOATS * 4u) + 12u);
}
template <bool INL>
__device__ __forceinline__ void bounce_apply(const float4 *l_mat, float fsl_mult, F3 n, uint32_t mat, float pt, F3 &o,
                                             F3 &d, float &a0, float &a1, float &a2, float &a3, float &tau, float &nth)
{
    // (INL: the two shading functions inlined -- an out-of-line one begins with s_waitcnt vmcnt(0), i.e.
    // waits for the previous packet's fifteen stores)
    nth = INL ? incidence_angle_inl(n, d) : incidence_angle(n, d);
    float4 R = INL ? fresnel_inl(l_mat[4u * mat], l_mat[4u * mat + 1u], l_mat[4u * mat + 2u], nth)
                   : fresnel(l_mat[4u * mat], l_mat[4u * mat + 1u], l_mat[4u * mat + 2u], nth);
    float fsl = fsl_mult * pt;
    fsl *= fsl;
    if (fsl > 1.f) { R.x /= fsl; R.y /= fsl; R.z /= fsl; R.w /= fsl; }
    const float b0 = a0 * R.x - a1 * R.y;
    const float b1 = a0 * R.y + a1 * R.x;
    const float b2 = a2 * R.z - a3 * R.w;
    const float b3 = a2 * R.w + a3 * R.z;
    a0 = b0; a1 = b1; a2 = b2; a3 = b3;
    tau += pt / kC;
    o = add3(mul3(d, pt), o);
    const float dn = dot3(d, n);
    d = sub3(d, mul3(n, 2.f * dn));
    o = add3(o, mul3(d, 1e-4f));
}

template <int AUX = 0>
__device__ __forceinline__ void store_survivor(Rsrc out, uint32_t cap4, uint32_t k4, uint32_t ray, uint32_t ntri,
                                               float nth, float fs0, F3 o, F3 d, float a0, float a1, float a2,
                                               float a3, float tau)
{
    stf_x<AUX>(out, H_RAY * cap4, k4, __uint_as_float(ray));
    stf_x<AUX>(out, H_TRI * cap4, k4, __uint_as_float(ntri));
    stf_x<AUX>(out, H_THETA * cap4, k4, nth);
    stf_x<AUX>(out, H_FS0 * cap4, k4, fs0);
    stf_x<AUX>(out, H_OX * cap4, k4, o.x);
    stf_x<AUX>(out, H_OY * cap4, k4, o.y);
    stf_x<AUX>(out, H_OZ * cap4, k4, o.z);
    stf_x<AUX>(out, H_DX * cap4, k4, d.x);
    stf_x<AUX>(out, H_DY * cap4, k4, d.y);
    stf_x<AUX>(out, H_DZ * cap4, k4, d.z);
    stf_x<AUX>(out, H_A0 * cap4, k4, a0);
    stf_x<AUX>(out, H_A1 * cap4, k4, a1);
    stf_x<AUX>(out, H_A2 * cap4, k4, a2);
    stf_x<AUX>(out, H_A3 * cap4, k4, a3);
    stf_x<AUX>(out, H_TAU * cap4, k4, tau);
}

constexpr uint32_t kShortList = 256u * 1024u;   // entries: one packet per wave fills the chip once
#ifndef HRT_FUSED_INLINE0
#define HRT_FUSED_INLINE0 1
#endif
#ifndef HRT_FUSED_WAVES0
#define HRT_FUSED_WAVES0 7   /* launch 0: at most 72 VGPRs (8 waves spill and are slower) */
#endif
#ifndef HRT_FUSED_WAVESB
#define HRT_FUSED_WAVESB 4
#endif
// Launch b as one kernel.  b == 0 (FIRST): the launch set (src/compute_paths.c:442-472 state init,
// :596-664 the first bounce).  b >= 1 (tables of one culling block): per entry of the live list the
// num_rx shadow rays of bounce b-1 IN ORDER with the theta carry (:671-723, quirks Q6-Q8), and
// bounce b.  A wave is a ray packet for every one of its traces (closest_hit).
//
// A workgroup owns a MACRO-CHUNK of K * 256 consecutive entries, K per thread (sub-chunk k =
// entries k*256 .. k*256+255 of the macro-chunk, so a wave's k-th packet is 64 consecutive entries).
// The launch is bound by the latency chain of a workgroup -- parameters, table staging, state loads,
// the wait for the chunks in front, stores: ~14 us against ~1.5 us of instructions per packet -- so
// K packets per wave share the chain (their loads are in flight together) and the grid is K times
// smaller (an empty workgroup of a later launch still costs its dispatch: 17 us for the 62 500 of
// C4 at K = 1).  Order inside a workgroup: (1) state loads; (2) the K bounce traces, count of the
// survivors, publish; (3) b >= 1: shadow traces + records of the K entries -- the chunks in front
// publish meanwhile; (4) prefix of the macro-chunk; (5) Fresnel / reflection of the survivors and
// their stores at (prefix + rank): sub-chunk-major, i.e. in entry order (stable).
// One macro-chunk of launch b = csrc/hrt_fused_body.inc: the body of hrt_fused_kernel, and -- CHAIN -- of
// hrt_chain_kernel's loops: there the tables are staged once per workgroup by the kernel, and the live list
// travels between the bounces with sc1 accesses.
template <bool TRI_IN_LDS, int VARIANT, bool FIRST, int K>
__global__ __launch_bounds__(HRT_BLOCK, (FIRST ? HRT_FUSED_WAVES0 : HRT_FUSED_WAVESB)) void hrt_fused_kernel(const hrt_kparams P, const uint32_t b)
{
    extern __shared__ float4 lds[];
    constexpr bool CHAIN = false;
    const uint32_t chunk_c = 0u, n_in_c = 0u, Ke_c = 0u, n_chunks_c = 0u;
    const bool sc1_in = false;
#include "hrt_fused_body.inc"
}

// ===================================================================================
// Launches b0 .. num_bounces as ONE kernel (tables on which whole bounces are fused): a persistent grid of G
// workgroups -- G = what the chip holds at once, from the occupancy query: every workgroup is resident, so the
// waits below are waits for running code -- loops over the bounces; inside a bounce workgroup j takes the
// macro-chunks j, j + G, ... (ascending, so a chunk still only waits for chunks that are running or done), and
// between two bounces stands a grid barrier (chain_barrier: flags, two round trips).  No fence: a bounce's
// survivors are written and read with sc1 accesses (fused_chunk<CHAIN>), the
// records -- which nobody reads in here -- stay ordinary stores.  The loop ends with the first empty live list.
// What it saves is the fixed cost of a launch (~6 us) and of dispatching a grid sized for the worst case (the
// host does not know the list's length: 31 250 empty workgroups = 8 us on C4), per bounce: C4's five tail
// launches, all of a short step.  The waits are bounded like lb_exclusive's (shared GPU: void step, the host
// falls back); with a timer every launch runs as its own kernel (hrt_fused_kernel), so per-launch times exist.
// ===================================================================================
// Grid barrier of hrt_chain_kernel: no atomics (1 024 returning adds on one word serialise at ~60 ns each: 61 us
// per barrier, measured), no fence -- flags.  Workgroup j stores `tag` into word j; the first workgroup of every 64
// waits (one load per lane) for its group's words and stores the group's word; everybody waits for the <= 64
// group words: two round trips.  The words are launch (b - 1)'s status words (W): dead once every workgroup is past
// that launch, and a tag (bit 30 set, bit 31 clear) is neither a fresh word (0) nor a status (bit 31 set).
// All threads call it; false = the wait ran out (the grid is not resident: shared GPU) or somebody else gave up.
constexpr uint32_t kChainTag = 1u << 30;
constexpr uint32_t kChainMaxGrid = 4096u;
__device__ __forceinline__ bool chain_barrier(const LbWords &W, const uint32_t tag, const uint32_t G, uint32_t *err_word,
                                              uint32_t *host_flag, const uint32_t max_polls, uint32_t *lds_flag)
{
    __builtin_amdgcn_s_waitcnt(0);   // this thread's stores have left (vmcnt 0) ...
    __syncthreads();                 // ... and so have the workgroup's
    const uint32_t tid = threadIdx.x;
    if (tid < 64u) {
        const uint32_t j = blockIdx.x, lane = tid;
        if (lane == 0) lb_store(W.chunk + j, tag);
        uint32_t polls = 0u;
        bool ok = true;
        if ((j & 63u) == 0u) {   // the group's first workgroup collects the group
            const bool want = j + lane < G;
            for (;;) {
                const uint32_t v = want ? lb_load(W.chunk + j + lane) : tag;
                if (HRT_BALLOT(v != tag) == 0ull) break;
                if (lb_load(err_word) & kErrVoid) { ok = false; break; }
                if (++polls > max_polls) {
                    if (lane == 0) give_up(err_word, host_flag, kErrChainTimeout);
                    ok = false;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            if (ok && lane == 0) lb_store(W.group + (j >> 6), tag);
        }
        if (ok) {
            const bool want = lane < ((G + 63u) >> 6);
            for (;;) {
                const uint32_t v = want ? lb_load(W.group + lane) : tag;
                if (HRT_BALLOT(v != tag) == 0ull) break;
                if (lb_load(err_word) & kErrVoid) { ok = false; break; }
                if (++polls > max_polls) {
                    if (lane == 0) give_up(err_word, host_flag, kErrChainTimeout);
                    ok = false;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        if (lane == 0) *lds_flag = ok ? 1u : 0u;
    }
    __syncthreads();
    return *lds_flag != 0u;
}

template <bool TRI_IN_LDS, int VARIANT, int K>
__global__ __launch_bounds__(HRT_BLOCK, HRT_FUSED_WAVESB) void hrt_chain_kernel(const hrt_kparams P, const uint32_t b0)
{
    extern __shared__ float4 lds[];
    constexpr bool FIRST = false, CHAIN = true;
    const uint32_t tid = threadIdx.x;
    uint32_t *counts = reinterpret_cast<uint32_t *>(P.ws + P.off_counts);
    uint32_t *err_word = &counts[P.num_bounces + 1];
    if (*err_word & kErrVoid) return;
    const uint32_t G = gridDim.x;   // (<= kChainMaxGrid and <= lb_chunks: the shim)
    const FusedLds L = fused_lds<TRI_IN_LDS, (VARIANT == 9)>(lds, P.num_tri, P.acc.num_leaf, P.num_rx, tid >> 6);
    {   // the tables, once
        const uint32_t T = P.num_tri, n_leaf = P.acc.num_leaf;
        if (TRI_IN_LDS) {
            const float4 *g_tri = reinterpret_cast<const float4 *>(P.tri);
            for (uint32_t k = tid; k < HRT_ROW * T; k += HRT_BLOCK) L.tri[k] = g_tri[k];
            if constexpr (VARIANT >= 4) {
                const float2 *g_tg = reinterpret_cast<const float2 *>(P.acc.tg);
                const float4 *g_leaf = reinterpret_cast<const float4 *>(P.acc.leaf);
                for (uint32_t k = tid; k < T; k += HRT_BLOCK) L.tg[k] = g_tg[k];
                for (uint32_t k = tid; k < 2u * n_leaf; k += HRT_BLOCK) L.leaf[k] = g_leaf[k];
            }
        }
        for (uint32_t k = tid; k < P.num_rx; k += HRT_BLOCK)
            L.rx[k] = make_float4(P.rx_pos[3 * k], P.rx_pos[3 * k + 1], P.rx_pos[3 * k + 2], 0.f);
        const float4 *g_mat = reinterpret_cast<const float4 *>(P.mat);
        for (uint32_t k = tid; k < 4u * HRT_NUM_MATERIALS; k += HRT_BLOCK) L.mat[k] = g_mat[k];
    }
    // Roll call before any work: is the whole grid resident?  Alone on the GPU it is within microseconds; when
    // other kernels hold the slots (processes sharing the GPU, a long kernel on another stream) it may never be,
    // and the answer should not take the 10 ms the waits inside the work are given: ~0.3 ms, then the step is
    // void (HRT_ERR_CHAIN_TIMEOUT) and the host goes on with a kernel per launch.
    const uint32_t roll_polls = P.tune.lb_max_polls < 256u ? P.tune.lb_max_polls : 256u;
    if (!chain_barrier(lb_words(P, b0 - 1u), kChainTag | 0xffffu, G, err_word, P.host_flag, roll_polls, &L.wcnt[9])) return;
    for (uint32_t b = b0; b <= P.num_bounces; ++b) {
        const uint32_t n_in_c = (b == b0) ? counts[b] : lb_load(&counts[b]);   // (uniform over the grid)
        if (n_in_c == 0u) return;
        const uint32_t Ke_c = (K > 1 && n_in_c <= kShortList && (uint64_t)G * HRT_BLOCK >= n_in_c) ? 1u : (uint32_t)K;
        const uint32_t n_chunks_c = (n_in_c + Ke_c * HRT_BLOCK - 1u) / (Ke_c * HRT_BLOCK);
        const bool sc1_in = b != b0;
        for (uint32_t chunk_c = blockIdx.x; chunk_c < n_chunks_c; chunk_c += G) {
#include "hrt_fused_body.inc"
        }
        if (b == P.num_bounces) return;
        if (!chain_barrier(lb_words(P, b - 1u), kChainTag | b, G, err_word, P.host_flag, P.tune.lb_max_polls, &L.wcnt[9])) return;
    }
}

// LoS pass (src/compute_paths.c:515-577): one WAVE per (rx, tx) pair, the lanes share the
// triangle loop (lane l tests triangles l, l+64, ... with the reference's plain sequence), then
// a lexicographic (distance, index) minimum over the wave -- the same winner as the reference's
// sequential scan with its strict '<' (lowest index on equal distance).
// Output per pair: HRT_LOS_FLOATS floats {status, a, tau, dir_tx xyz, freq_shift, -}.
__device__ __forceinline__ void los_pairs(const hrt_kparams &P, const uint32_t block)
{
    const float4 *tri = reinterpret_cast<const float4 *>(P.tri);
    float *out = reinterpret_cast<float *>(P.ws + P.off_los);
    const uint32_t n = P.num_rx * P.num_tx;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t off = block * (HRT_BLOCK / 64u) + (threadIdx.x >> 6);
    if (off >= n) return;   // whole wave
    const uint32_t rx = off / P.num_tx, tx = off - rx * P.num_tx;
    const F3 o = {P.tx_pos[3 * tx], P.tx_pos[3 * tx + 1], P.tx_pos[3 * tx + 2]};
    const F3 r = {P.rx_pos[3 * rx], P.rx_pos[3 * rx + 1], P.rx_pos[3 * rx + 2]};
    const F3 d = sub3(r, o);
    float *q = out + 8u * off;
    uint32_t status;
    float a = 0.f, tau = 0.f, fs = 0.f;
    F3 u = {0.f, 0.f, 0.f};
    if (dot3(d, d) < kEps) {   // wave-uniform: every lane holds the same pair
        status = 0u;           // coincident: unit gain, zero delay (:531-544)
        a = 1.f;
    } else {
        float best = 1e9f;
        uint32_t who = HRT_NO_HIT;
        for (uint32_t j = lane; j < P.num_tri; j += 64u) {
            const float4 q0 = tri[HRT_ROW * j], q1 = tri[HRT_ROW * j + 1], q2 = tri[HRT_ROW * j + 2];
            const F3 v1 = {q0.x, q0.y, q0.z};
            const F3 e1 = {q0.w, q1.x, q1.y};
            const F3 e2 = {q1.z, q1.w, q2.x};
            const F3 pv = cross3(d, e2);
            const float det = dot3(e1, pv);
            if (det > -kEps && det < kEps) continue;
            const F3 s = sub3(o, v1);
            const float uu = dot3(s, pv) / det;
            if (uu < -kEps || uu > kOnePlusEps) continue;
            const F3 qq = cross3(s, e1);
            const float vv = dot3(d, qq) / det;
            const float ww = uu + vv;
            if (vv < -kEps || ww > kOnePlusEps) continue;
            const float dist = dot3(e2, qq) / det;
            if (dist > kEps && dist < best) { best = dist; who = j; }   // ascending j per lane
        }
        // accepted distances are positive floats: their bit patterns order like the values
        uint32_t kd = __float_as_uint(best), kj = who;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const uint32_t od = (uint32_t)__shfl_xor((int)kd, m), oj = (uint32_t)__shfl_xor((int)kj, m);
            const bool take = (od < kd) || (od == kd && oj < kj);
            kd = take ? od : kd;
            kj = take ? oj : kj;
        }
        const float t = __uint_as_float(kd);
        if (kj != HRT_NO_HIT && t <= 1.f) {
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
    if (lane == 0) {
        q[0] = __uint_as_float(status);
        q[1] = a; q[2] = tau; q[3] = u.x; q[4] = u.y; q[5] = u.z; q[6] = fs; q[7] = 0.f;
    }
}
__global__ __launch_bounds__(HRT_BLOCK) void hrt_los_kernel(const hrt_kparams P) { los_pairs(P, blockIdx.x); }

// Big tables (one wave per pair walks 100 002 triangles in 0.8 ms -- 3.5 % of a step of the generated city): the
// table is cut into kLosSlices x 4 wave-slices per pair; only the SMALLEST accepted distance matters (blocked
// iff it is <= 1, quirk Q6), so the waves merge the complement of its bit pattern with an atomic maximum into
// a zeroed word, and the wave that counts in last writes the pair's record.  Words: the free tail of the
// counter block (off_counts + 3 cnt_stride: {max of ~distance bits, waves done} per pair, up to 32 pairs).
constexpr uint32_t kLosSlices = 64u;
__global__ __launch_bounds__(HRT_BLOCK) void hrt_los_big_kernel(const hrt_kparams P)
{
    const float4 *tri = reinterpret_cast<const float4 *>(P.tri);
    float *out = reinterpret_cast<float *>(P.ws + P.off_los);
    uint32_t *words = reinterpret_cast<uint32_t *>(P.ws + P.off_counts + 3u * P.cnt_stride);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t off = blockIdx.x / kLosSlices, slice = blockIdx.x - off * kLosSlices;
    const uint32_t ws_id = slice * (HRT_BLOCK / 64u) + (threadIdx.x >> 6), n_ws = kLosSlices * (HRT_BLOCK / 64u);
    const uint32_t rx = off / P.num_tx, tx = off - rx * P.num_tx;
    const F3 o = {P.tx_pos[3 * tx], P.tx_pos[3 * tx + 1], P.tx_pos[3 * tx + 2]};
    const F3 r = {P.rx_pos[3 * rx], P.rx_pos[3 * rx + 1], P.rx_pos[3 * rx + 2]};
    const F3 d = sub3(r, o);
    const bool coincident = dot3(d, d) < kEps;   // wave-uniform
    if (!coincident) {
        const uint32_t per = (P.num_tri + n_ws - 1u) / n_ws;
        const uint32_t j0 = ws_id * per, j1 = min(P.num_tri, j0 + per);
        float best = 1e9f;
        for (uint32_t j = j0 + lane; j < j1; j += 64u) {
            const float4 q0 = tri[HRT_ROW * j], q1 = tri[HRT_ROW * j + 1], q2 = tri[HRT_ROW * j + 2];
            const F3 v1 = {q0.x, q0.y, q0.z};
            const F3 e1 = {q0.w, q1.x, q1.y};
            const F3 e2 = {q1.z, q1.w, q2.x};
            const F3 pv = cross3(d, e2);
            const float det = dot3(e1, pv);
            if (det > -kEps && det < kEps) continue;
            const F3 s = sub3(o, v1);
            const float uu = dot3(s, pv) / det;
            if (uu < -kEps || uu > kOnePlusEps) continue;
            const F3 qq = cross3(s, e1);
            const float vv = dot3(d, qq) / det;
            const float ww = uu + vv;
            if (vv < -kEps || ww > kOnePlusEps) continue;
            const float dist = dot3(e2, qq) / det;
            if (dist > kEps && dist < best) best = dist;
        }
        uint32_t kd = best < 1e9f ? __float_as_uint(best) : 0xffffffffu;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) kd = min(kd, (uint32_t)__shfl_xor((int)kd, m));
        if (lane == 0 && kd != 0xffffffffu) atomicMax(words + 2u * off, ~kd);
    }
    uint32_t last = 0u;
    if (lane == 0) {
        __threadfence();
        last = atomicAdd(words + 2u * off + 1u, 1u) == n_ws - 1u ? 1u : 0u;
    }
    if (!__builtin_amdgcn_readfirstlane((int)last)) return;
    __threadfence();
    const uint32_t mx = __hip_atomic_load(words + 2u * off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t status;
    float a = 0.f, tau = 0.f, fs = 0.f;
    F3 u = {0.f, 0.f, 0.f};
    if (coincident) {
        status = 0u;
        a = 1.f;
    } else if (mx != 0u && __uint_as_float(~mx) <= 1.f) {
        status = 1u;   // blocked (:548-554)
    } else {
        status = 2u;
        const float dist = sqrtf(dot3(d, d));
        u = {d.x / dist, d.y / dist, d.z / dist};
        const float fsl = P.fsl_mult * dist;
        a = (fsl > 1.f) ? 1.f / fsl : 1.f;
        tau = dist / kC;
        const F3 tv = {P.tx_vel[0], P.tx_vel[1], P.tx_vel[2]};
        const F3 rv = {P.rx_vel[0], P.rx_vel[1], P.rx_vel[2]};
        fs = (dot3(tv, u) - dot3(rv, u)) * P.dop_mult;
    }
    if (lane == 0) {
        float *q = out + 8u * off;
        q[0] = __uint_as_float(status);
        q[1] = a; q[2] = tau; q[3] = u.x; q[4] = u.y; q[5] = u.z; q[6] = fs; q[7] = 0.f;
    }
}

// ===================================================================================
// Launch directions on the device.  The reference evaluates
//   k = p + .5f; phi = (float)acos(1.f - 2.f*k/N); theta = pi_f*(1.f + sqrtf(5.f))*k   (float)
//   d = ((float)(cos(theta)*sin(phi)), (float)(sin(theta)*sin(phi)), (float)cos(phi))   (double libm)
// (src/compute_paths.c:444-451).  The float steps are exact IEEE operations and identical here.
// The double-precision acos/cos/sin are the DEVICE library's, which may differ from glibc's in
// the last bits -- harmless unless the value is about to be rounded to float right next to a
// rounding boundary.  So every double -> float rounding is checked: if the double lies within
// `guard` relative (2^-44, thousands of double ulps; both libraries are good to a few) of the
// midpoint between two floats, the ray is put on a fix list and the HOST recomputes it with
// its own libm.  Expected list length ~ 5e-7 per ray (measured: tests/test_gpu_launch_dirs.py);
// everything else is provably the same float.  An overfull list makes the caller fall back to
// the host generator.
// ===================================================================================
__device__ __forceinline__ bool near_float_boundary(double x, float f)
{
    const double up = 0.5 * ((double)f + (double)nextafterf(f, 3.0e38f));
    const double dn = 0.5 * ((double)f + (double)nextafterf(f, -3.0e38f));
    const double dist = fmin(fabs(x - up), fabs(x - dn));
    return !(dist > fabs(x) * 0x1p-44 + 1e-300);
}

__global__ void hrt_launch_dirs_kernel(uint64_t num_paths, uint32_t rank, uint32_t count,
                                       uint32_t chunk, uint64_t num_local, float *dirs,
                                       uint32_t *fix_count, uint32_t *fix_list, uint32_t fix_cap)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= num_local) return;
    const uint64_t p = ((i / chunk) * count + rank) * chunk + i % chunk;
    const float k = (float)p + .5f;
    const float arg = 1.f - 2.f * k / (float)num_paths;
    const double ph_d = acos((double)arg);
    const float phi = (float)ph_d;
    const float theta = kPi * (1.f + sqrtf(5.f)) * k;
    const double sp = sin((double)phi);
    const double x = cos((double)theta) * sp, y = sin((double)theta) * sp, z = cos((double)phi);
    const float fx = (float)x, fy = (float)y, fz = (float)z;
    dirs[3 * i] = fx;
    dirs[3 * i + 1] = fy;
    dirs[3 * i + 2] = fz;
    // the products inherit ~2 ulp from each factor: still far inside the guard
    if (near_float_boundary(ph_d, phi) || near_float_boundary(x, fx) ||
        near_float_boundary(y, fy) || near_float_boundary(z, fz)) {
        const uint32_t slot = atomicAdd(fix_count, 1u);
        if (slot < fix_cap) fix_list[slot] = (uint32_t)i;
    }
}

// ===================================================================================
// Coherent launch order on the device (the job of hrt_launch_order_host, problem.c): the launch
// set is walked in z-bands, serpentine in azimuth, so that 64 consecutive positions are a narrow
// packet.  On a Fibonacci sphere the polar coordinate is monotone in the path index, so a band is a
// RANGE of (local) indices: the host cuts the shard into segments -- ranges of at most 32768 rays
// that do not cross a band boundary, found by bisection -- and one workgroup sorts one segment by
// azimuth in LDS (bitonic, keys = 17 bits of azimuth | 15 bits of offset in the segment: unique, so
// the result is deterministic).  Results do not depend on the order (records carry ray ids).
// ===================================================================================
#define HRT_ORDER_SEG 32768u
__global__ __launch_bounds__(1024) void hrt_launch_order_kernel(const uint32_t *seg_start,
                                                                 const uint32_t *seg_band,
                                                                 uint64_t num_paths, uint32_t rank,
                                                                 uint32_t count, uint32_t chunk,
                                                                 uint32_t *order)
{
    extern __shared__ uint32_t okeys[];
    const uint32_t s0 = seg_start[blockIdx.x], n = seg_start[blockIdx.x + 1] - s0;
    const bool flip = seg_band[blockIdx.x] & 1u;
    uint32_t npad = 2u;
    while (npad < n) npad <<= 1;
    const float golden = kPi * (1.f + sqrtf(5.f));
    for (uint32_t k = threadIdx.x; k < npad; k += blockDim.x) {
        uint32_t key = 0xffffffffu;
        if (k < n) {
            const uint64_t i = (uint64_t)s0 + k;
            const uint64_t p = ((i / chunk) * count + rank) * chunk + i % chunk;
            const float kf = (float)p + .5f;
            const double th = (double)(golden * kf) * 0.15915494309189533577;   // turns
            double fr = th - floor(th);
            uint32_t az = (uint32_t)(fr * 131072.0);
            if (az > 131071u) az = 131071u;
            if (flip) az = 131071u - az;
            key = (az << 15) | k;
        }
        okeys[k] = key;
    }
    for (uint32_t size = 2u; size <= npad; size <<= 1)
        for (uint32_t stride = size >> 1; stride > 0u; stride >>= 1) {
            __syncthreads();
            for (uint32_t t = threadIdx.x; t < (npad >> 1); t += blockDim.x) {
                const uint32_t a = 2u * t - (t & (stride - 1u)), b2 = a + stride;
                const bool up = (a & size) == 0u;
                const uint32_t ka = okeys[a], kb = okeys[b2];
                if ((ka > kb) == up) { okeys[a] = kb; okeys[b2] = ka; }
            }
        }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) order[s0 + k] = s0 + (okeys[k] & 0x7fffu);
    (void)num_paths;
}

// Builder of the per-RX direction tables (hrt_krxt): one wave per (rx, cell); the lanes take the
// triangles 64 at a time through packet_culls with the CELL as the packet -- lines through the ball
// (rx, ro_bin) with directions within the cell's cone (widened by the largest packet half-angle
// served), origins anywhere in the scene's ball (so the tolerances are the largest any packet can
// have, and the "behind the origin" part of the test can never fire).  Whatever is not culled here
// cannot be culled for any packet inside the cell.  Output: candidate bit masks [rx][cell][T / 64].
__global__ __launch_bounds__(64) void hrt_rxt_build_kernel(const float *tri_f, uint32_t num_tri,
                                                           const float *rx_pos, const float *bin_dir4,
                                                           const float *bin_cs2, const float *ro_bin,
                                                           float cx, float cy, float cz, float region_r,
                                                           unsigned long long *masks)
{
    const float4 *tri = reinterpret_cast<const float4 *>(tri_f);
    const uint32_t cell = blockIdx.x, rx = blockIdx.y, lane = threadIdx.x;
    const uint32_t W = (num_tri + 63u) / 64u;
    Packet P;
    P.oc = {rx_pos[3 * rx], rx_pos[3 * rx + 1], rx_pos[3 * rx + 2]};
    P.ro = ro_bin[rx];
    P.bc = {cx, cy, cz};
    P.br = region_r;
    P.ax = {bin_dir4[4 * cell], bin_dir4[4 * cell + 1], bin_dir4[4 * cell + 2]};
    P.cosa = bin_cs2[2 * cell];
    P.sina = bin_cs2[2 * cell + 1];
    P.usable = true;
    for (uint32_t r = 0; r < W; ++r) {
        const uint32_t j = r * 64u + lane;
        bool cand = false;
        if (j < num_tri)
            cand = !packet_culls(P, tri[HRT_ROW * j], tri[HRT_ROW * j + 1], tri[HRT_ROW * j + 2],
                                 tri[HRT_ROW * j + 3], tri[HRT_ROW * j + 4]);
        const unsigned long long m = __builtin_amdgcn_ballot_w64(cand);
        if (lane == 0) masks[((uint64_t)rx * HRT_RXT_BINS + cell) * W + r] = m;
    }
}

// Builder of the TX cell masks (hrt_kpatch.txcell): one wave per (TX, cube-map cell); the packet = lines through the TX
// (origin = the TX itself: launch rays start there exactly) with directions in the cell's cone.
__global__ __launch_bounds__(64) void hrt_txcell_build_kernel(const float *tri_f, uint32_t num_tri, const float *tx_pos,
                                                              const float *bin_dir4, const float *bin_cs2, unsigned long long *masks)
{
    const float4 *tri = reinterpret_cast<const float4 *>(tri_f);
    const uint32_t cell = blockIdx.x, tx = blockIdx.y, lane = threadIdx.x;
    Packet P;
    P.oc = {tx_pos[3 * tx], tx_pos[3 * tx + 1], tx_pos[3 * tx + 2]};
    P.bc = P.oc;
    P.ro = 4e-6f * ((fabsf(P.oc.x) + fabsf(P.oc.y)) + fabsf(P.oc.z)) + 4e-6f;   // (rounding of anything formed from the TX)
    P.br = P.ro;
    P.ax = {bin_dir4[4 * cell], bin_dir4[4 * cell + 1], bin_dir4[4 * cell + 2]};
    P.cosa = bin_cs2[2 * cell];
    P.sina = bin_cs2[2 * cell + 1];
    P.usable = true;
    for (uint32_t r = 0; r < HRT_PATCH_WORDS; ++r) {
        const uint32_t j = r * 64u + lane;
        bool cand = false;
        if (j < num_tri)
            cand = !packet_culls(P, tri[HRT_ROW * j], tri[HRT_ROW * j + 1], tri[HRT_ROW * j + 2], tri[HRT_ROW * j + 3],
                                 tri[HRT_ROW * j + 4]);
        const unsigned long long m = __builtin_amdgcn_ballot_w64(cand);
        if (lane == 0) masks[((uint64_t)tx * HRT_RXT_BINS + cell) * HRT_PATCH_WORDS + r] = m;
    }
}

// Builder of the patch tables (hrt_kpatch): one thread per (patch, apex).  The patch is cell (iu, iv) of
// the grid of triangle j, enlarged by HRT_PATCH_MARGIN of a cell on every side and by hball out of the
// plane (the lanes are only served inside that), its packet { origins in the cell's ball, lines that meet
// ball(apex, ro), directions within the cone the two balls span } goes through packet_culls against
// every row of the table.  Apexes: the RXs (rays towards the apex), then per TX its image in the plane of
// triangle j (rays leaving the apex).  A packet the test cannot serve (cone beyond 60 degrees: the apex
// is next to the patch) keeps every triangle.  Output [apex][patch][HRT_PATCH_WORDS].
__global__ __launch_bounds__(256) void hrt_patch_build_kernel(const float *tri_f, uint32_t num_tri, const float *pdef_f,
                                                              const uint32_t *patch_tri, uint32_t num_patch,
                                                              const float *apex_pos, uint32_t num_rx, float hball,
                                                              float ro_rx, float ro_img, unsigned long long *masks)
{
    const float4 *tri = reinterpret_cast<const float4 *>(tri_f);
    const float4 *pdef = reinterpret_cast<const float4 *>(pdef_f);
    const uint32_t pid = blockIdx.x * 256u + threadIdx.x, a = blockIdx.y;
    if (pid >= num_patch) return;
    const uint32_t j = patch_tri[pid];
    const float4 q0 = tri[HRT_ROW * j], q1 = tri[HRT_ROW * j + 1], q2 = tri[HRT_ROW * j + 2];
    const float4 p0 = pdef[2u * j], p1 = pdef[2u * j + 1u];
    const uint32_t base = __float_as_uint(p0.w), bits = __float_as_uint(p1.w), nu = bits & 0xffffu, nv = bits >> 16;
    const uint32_t c = pid - base, iv = c / nu, iu = c - iv * nu;
    const F3 v1 = {q0.x, q0.y, q0.z}, e1 = {q0.w, q1.x, q1.y}, e2 = {q1.z, q1.w, q2.x};
    const float fu = ((float)iu + 0.5f) / (float)nu, fv = ((float)iv + 0.5f) / (float)nv;
    const float hu = (0.5f + HRT_PATCH_MARGIN) / (float)nu, hv = (0.5f + HRT_PATCH_MARGIN) / (float)nv;
    Packet P;
    P.bc = add3(v1, add3(mul3(e1, fu), mul3(e2, fv)));
    const F3 d1 = add3(mul3(e1, hu), mul3(e2, hv)), d2 = sub3(mul3(e1, hu), mul3(e2, hv));
    P.br = sqrtf(fmaxf(fdot3(d1, d1), fdot3(d2, d2))) * 1.001f + hball +
           4e-6f * ((fabsf(P.bc.x) + fabsf(P.bc.y)) + fabsf(P.bc.z));
    const bool image = a >= num_rx;
    const F3 ap = {apex_pos[3 * a], apex_pos[3 * a + 1], apex_pos[3 * a + 2]};
    P.oc = image ? image_of(ap, q0, q2) : ap;
    P.ro = image ? ro_img : ro_rx;
    const F3 w = sub3(P.oc, P.bc);
    const float dist = sqrtf(fdot3(w, w));
    const float inv = (image ? -1.f : 1.f) / fmaxf(dist, 1e-30f);
    P.ax = {w.x * inv, w.y * inv, w.z * inv};
    // directions of the lines between the two balls: sin(half-angle) <= (br + ro) / dist
    const float sa = (P.br + P.ro) / fmaxf(dist, 1e-30f) * 1.001f + 1e-5f;
    P.usable = sa < 0.86f;   // (NaN: not usable)
    P.sina = P.usable ? sa : 0.86f;
    P.cosa = sqrtf(fmaxf(0.f, 1.f - P.sina * P.sina)) * 0.9999f - 1e-6f;
    unsigned long long m[HRT_PATCH_WORDS];
#pragma unroll
    for (uint32_t r = 0; r < HRT_PATCH_WORDS; ++r) m[r] = 0ull;
    for (uint32_t t = 0; t < num_tri; ++t) {
        const bool cand = !P.usable || !packet_culls(P, tri[HRT_ROW * t], tri[HRT_ROW * t + 1], tri[HRT_ROW * t + 2],
                                                    tri[HRT_ROW * t + 3], tri[HRT_ROW * t + 4]);
#pragma unroll
        for (uint32_t r = 0; r < HRT_PATCH_WORDS; ++r)
            if ((t >> 6) == r && cand) m[r] |= 1ull << (t & 63u);
    }
    unsigned long long *out = masks + ((uint64_t)a * num_patch + pid) * HRT_PATCH_WORDS;
#pragma unroll
    for (uint32_t r = 0; r < HRT_PATCH_WORDS; ++r) out[r] = m[r];
}

// ===================================================================================
// Re-sort of the survivors of bounce b (hrt_ksort): keys, then a stable radix sort of (key, index)
// (the kernels below), then the permutation of the 15 field arrays from the scratch block into hit
// block b.  The number of survivors is only known on the device (counts[b + 1]): every kernel is
// launched over the block's capacity and works on the survivors only.
// ===================================================================================
__device__ __forceinline__ uint32_t spread5(uint32_t v)   // 5 bits -> every third bit
{
    v &= 31u;
    v = (v | (v << 8)) & 0x100fu;
    v = (v | (v << 4)) & 0x10c3u;
    v = (v | (v << 2)) & 0x1249u;
    return v;
}

__global__ void hrt_sort_keys_kernel(const hrt_kparams P, const uint32_t b)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.cap) return;
    const uint32_t *counts = reinterpret_cast<const uint32_t *>(P.ws + P.off_counts);
    const uint32_t n = counts[b + 1];
    uint32_t *keys = reinterpret_cast<uint32_t *>(P.ws + P.sort.off_keys);
    uint32_t *idx = keys + 2u * P.cap;
    uint32_t key = 0xffffffffu;
    if (i >= n) return;   // the sort's size is the survivor count
    {
        const uint32_t cap4 = (uint32_t)P.cap * 4u, i4 = i * 4u;
        const Rsrc r = make_rsrc(P.ws + P.sort.off_scratch);
        const uint32_t ray = ldu(r, H_RAY * cap4, i4);
        const F3 o = {ldf(r, H_OX * cap4, i4), ldf(r, H_OY * cap4, i4), ldf(r, H_OZ * cap4, i4)};
        const F3 d = {ldf(r, H_DX * cap4, i4), ldf(r, H_DY * cap4, i4), ldf(r, H_DZ * cap4, i4)};
        const uint32_t tx = ray / P.num_local;
        // cell of the origin: bits[k] bits per axis (cells of about equal edge: the host gives the
        // longer axes more bits), concatenated x | y | z, split into a coarse part (the top `hi` bits
        // of every axis) and a fine part
        auto cell = [](float x, float lo, float inv, uint32_t bits) {
            const float u = (x - lo) * inv, top = (float)((1u << bits) - 1u);
            return (uint32_t)(u > 0.f ? (u < top ? (int)u : (int)top) : 0);   // NaN -> 0
        };
        const uint32_t bx = P.sort.bits[0], by = P.sort.bits[1], bz = P.sort.bits[2];
        const uint32_t cx = cell(o.x, P.sort.lo[0], P.sort.inv_cell[0], bx);
        const uint32_t cy = cell(o.y, P.sort.lo[1], P.sort.inv_cell[1], by);
        const uint32_t cz = cell(o.z, P.sort.lo[2], P.sort.inv_cell[2], bz);
        // interleave the axes' bits from the least significant up (an axis that runs out of bits
        // drops out): a Morton code for cells of unequal counts; its low `nfine` bits sort BEHIND
        // the direction bin
        uint32_t code = 0u, pos = 0u;
#pragma unroll
        for (uint32_t lv = 0; lv < 15u; ++lv) {
            if (lv < bx) code |= ((cx >> lv) & 1u) << pos++;
            if (lv < by) code |= ((cy >> lv) & 1u) << pos++;
            if (lv < bz) code |= ((cz >> lv) & 1u) << pos++;
        }
        const uint32_t fbits = P.sort.nfine;
        const uint32_t coarse = code >> fbits, fine = code & ((1u << fbits) - 1u);
        // direction bin: cube face (3 bits) x a dir_res x dir_res grid of the two minor components
        const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
        uint32_t f = 0u;
        float mj = d.x, c1 = d.y, c2 = d.z;
        if (ay > ax && ay >= az) { f = 1u; mj = d.y; c1 = d.z; c2 = d.x; }
        else if (az > ax && az > ay) { f = 2u; mj = d.z; c1 = d.x; c2 = d.y; }
        const float inv = 1.f / mj, res = (float)P.sort.dir_res;
        auto q = [&](float c) {
            const float u = (c * inv * 0.5f + 0.5f) * res;
            return (uint32_t)(u > 0.f ? (u < res - 1.f ? (int)u : (int)res - 1) : 0);   // NaN -> 0
        };
        const uint32_t db = (((f + (mj < 0.f ? 3u : 0u)) * P.sort.dir_res + q(c1)) * P.sort.dir_res) + q(c2);
        key = (tx << P.sort.tx_shift) | (((coarse << P.sort.dir_bits) | db) << fbits) | fine;
    }
    keys[i] = key;
    idx[i] = i;
}

// ---- the sort itself: a stable LSD radix sort of (key, index) pairs whose SIZE is the survivor
// count on the device (counts[b + 1]) -- launched over the block's capacity, workgroups past the
// count leave at once.  Digits of up to 11 bits: the usual 20-bit key is two passes.  Per pass three
// kernels over tiles of kSortTile pairs:
//   hist     per tile the histogram of the pass's digit            -> hist[digit][tile]
//   scan     per digit (one workgroup each) the exclusive scan over the tiles, in place, and the
//            digit's total                                          -> hist[digit][tile], total[digit]
//   scatter  per tile: offsets of the digits from the totals, and every pair to
//            base(digit) + hist[digit][tile] + (its rank among the tile's pairs of that digit)
// The rank is found without a local sort: the four waves of a workgroup take the four quarters of
// the tile, a wave walks its quarter 64 pairs at a time, the lanes with the same digit find each
// other with one ballot per digit bit, rank = pairs of that digit earlier in the quarter + lanes of
// the match mask below one's own -- order of arrival kept, so every pass is stable.
// Buffers (hrt_ksort.off_keys): keys A, keys B, index A, index B (cap words each); pass p reads
// A/B by parity.  hist: kSortBins x tiles words at hrt_ksort.off_tmp, the kSortBins totals behind.
constexpr uint32_t kSortTile = 2048u, kSortBins = 2048u;

struct SortArgs {
    uint32_t *key_a, *key_b, *idx_a, *idx_b, *hist, *total;
    const uint32_t *count;   // counts[b + 1]
    uint32_t tiles_max, shift, bits, parity;
};

__global__ __launch_bounds__(256) void hrt_sort_hist_kernel(const SortArgs A)
{
    __shared__ uint32_t h[kSortBins];
    const uint32_t n = *A.count, tile = blockIdx.x, tid = threadIdx.x;
    const uint32_t i0 = tile * kSortTile, bins = 1u << A.bits, dmask = bins - 1u;
    if (i0 >= n) return;
    const uint32_t *keys = A.parity ? A.key_b : A.key_a;
    for (uint32_t d = tid; d < bins; d += 256u) h[d] = 0u;
    __syncthreads();
#pragma unroll
    for (uint32_t r = 0; r < kSortTile / 256u; ++r) {
        const uint32_t i = i0 + r * 256u + tid;
        if (i < n) atomicAdd(&h[(keys[i] >> A.shift) & dmask], 1u);
    }
    __syncthreads();
    for (uint32_t d = tid; d < bins; d += 256u) A.hist[d * A.tiles_max + tile] = h[d];
}

// inclusive scan of x over the workgroup's threads in thread order; `all` = the total
template <uint32_t WAVES>
__device__ __forceinline__ uint32_t block_scan_incl(uint32_t x, uint32_t *part, uint32_t tid, uint32_t &all)
{
    const uint32_t lane = tid & 63u, wave = tid >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = (uint32_t)__shfl_up((int)x, d);
        if ((int)lane >= d) x += y;
    }
    __syncthreads();   // (part may still be read from the previous call)
    if (lane == 63u) part[wave] = x;
    __syncthreads();
    uint32_t before = 0u;
    all = 0u;
#pragma unroll
    for (uint32_t w = 0; w < WAVES; ++w) {
        const uint32_t pw = part[w];
        before += w < wave ? pw : 0u;
        all += pw;
    }
    return before + x;
}

// one workgroup per digit: exclusive scan of hist[digit][0 .. tiles) in place, total[digit]
__global__ __launch_bounds__(1024) void hrt_sort_scan_kernel(const SortArgs A)
{
    __shared__ uint32_t part[16];
    const uint32_t n = *A.count, tid = threadIdx.x;
    const uint32_t tiles = (n + kSortTile - 1u) / kSortTile;
    uint32_t *row = A.hist + blockIdx.x * A.tiles_max;
    uint32_t carry = 0u;
    for (uint32_t base = 0; base < tiles; base += 1024u) {   // (uniform trip count)
        const uint32_t t = base + tid;
        const uint32_t v = t < tiles ? row[t] : 0u;
        uint32_t all;
        const uint32_t incl = block_scan_incl<16>(v, part, tid, all);
        if (t < tiles) row[t] = carry + incl - v;
        carry += all;
    }
    if (tid == 0) A.total[blockIdx.x] = carry;
}

__global__ __launch_bounds__(256) void hrt_sort_scatter_kernel(const SortArgs A)
{
    __shared__ uint32_t base[kSortBins], cnt[4][kSortBins], part[4];
    const uint32_t n = *A.count, tile = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t i0 = tile * kSortTile, bins = 1u << A.bits, dmask = bins - 1u;
    if (i0 >= n) return;
    const uint32_t *kin = A.parity ? A.key_b : A.key_a, *iin = A.parity ? A.idx_b : A.idx_a;
    uint32_t *kout = A.parity ? A.key_a : A.key_b, *iout = A.parity ? A.idx_a : A.idx_b;
    // base[d] = pairs with a smaller digit (exclusive scan of the totals) + this tile's offset in d;
    // thread t owns the digits 8 t .. 8 t + 7
    {
        uint32_t v[8], sum = 0u;
#pragma unroll
        for (uint32_t j = 0; j < 8u; ++j) {
            const uint32_t d = tid * 8u + j;
            v[j] = d < bins ? A.total[d] : 0u;
            sum += v[j];
        }
        uint32_t all;
        uint32_t run = block_scan_incl<4>(sum, part, tid, all) - sum;
#pragma unroll
        for (uint32_t j = 0; j < 8u; ++j) {
            const uint32_t d = tid * 8u + j;
            if (d < bins) base[d] = run + A.hist[d * A.tiles_max + tile];
            run += v[j];
        }
        for (uint32_t d = tid; d < bins; d += 256u) { cnt[0][d] = 0u; cnt[1][d] = 0u; cnt[2][d] = 0u; cnt[3][d] = 0u; }
    }
    __syncthreads();
    // the wave's quarter of the tile: count its digits, then the quarters in front of it
    constexpr uint32_t Q = kSortTile / 4u;
    const uint32_t q0 = i0 + wave * Q;
    uint32_t key[Q / 64u], idx[Q / 64u];
#pragma unroll
    for (uint32_t r = 0; r < Q / 64u; ++r) {
        const uint32_t i = q0 + r * 64u + lane;
        key[r] = i < n ? kin[i] : 0u;
        idx[r] = i < n ? iin[i] : 0u;
        if (i < n) atomicAdd(&cnt[wave][(key[r] >> A.shift) & dmask], 1u);
    }
    __syncthreads();
    for (uint32_t d = tid; d < bins; d += 256u) {
        // cnt[w][d] becomes: pairs of digit d in the quarters in front of quarter w (running counter of wave w)
        const uint32_t c0 = cnt[0][d], c1 = cnt[1][d], c2 = cnt[2][d];
        cnt[0][d] = 0u; cnt[1][d] = c0; cnt[2][d] = c0 + c1; cnt[3][d] = c0 + c1 + c2;
    }
    __syncthreads();
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (uint32_t r = 0; r < Q / 64u; ++r) {
        const uint32_t i = q0 + r * 64u + lane;
        const bool have = i < n;
        const uint32_t d = (key[r] >> A.shift) & dmask;
        unsigned long long m = HRT_BALLOT(have);
        for (uint32_t bit = 0; bit < A.bits; ++bit) {
            const unsigned long long bb = HRT_BALLOT((d >> bit) & 1u);
            m &= ((d >> bit) & 1u) ? bb : ~bb;
        }
        // (m: the lanes of this round with my digit; all lanes of a match group read the counter before
        // its leader bumps it: same wave, program order)
        uint32_t at = 0u;
        if (have) at = cnt[wave][d];
        const uint32_t rank = (uint32_t)__popcll(m & lt);
        if (have && rank == 0u) cnt[wave][d] = at + (uint32_t)__popcll(m);
        if (have) {
            const uint32_t dst = base[d] + at + rank;
            kout[dst] = key[r];
            iout[dst] = idx[r];
        }
    }
}

// moves the 15 field arrays of the survivors from the scratch block into hit block b, in sorted order
__global__ void hrt_sort_permute_kernel(const hrt_kparams P, const uint32_t b, const uint32_t *perm)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t *counts = reinterpret_cast<const uint32_t *>(P.ws + P.off_counts);
    if (i >= counts[b + 1]) return;
    const uint32_t src4 = perm[i] * 4u, dst4 = i * 4u, cap4 = (uint32_t)P.cap * 4u;
    const Rsrc in = make_rsrc(P.ws + P.sort.off_scratch), out = hit_blk(P, b);
#pragma unroll
    for (uint32_t f = 0; f < 15u; ++f) stu(out, f * cap4, dst4, ldu(in, f * cap4, src4));
}

// launch Doppler term of the scatter records, src/compute_paths.c:494-500: out[p] =
// dot(tx_vel, d_p) * f/c in the reference's float sequence (no contraction in this file)
__global__ void hrt_fs0_kernel(const float *dirs, uint64_t n, float vx, float vy, float vz, float mult,
                               float *out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const F3 tv = {vx, vy, vz};
    const F3 d = {dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]};
    out[i] = dot3(tv, d) * mult;
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
    case 5: { float c_; hrt_sincosf(x, &y, &c_); break; }
    case 6: { float s_; hrt_sincosf(x, &s_, &y); break; }
    case 7: y = hrt_cosf_nb(x); break;
    default: {   // src/compute_paths.c:281-283 with dot(n, d) = x
        float th = (float)acos((double)x);
        if (th > kPi * 0.5f) th = kPi - th;
        y = th;
    }
    }
    out[i] = y;
}

template <bool LDS, int V>
static void launch_trace_t(const hrt_kparams *P, uint32_t bounce, uint32_t blocks, size_t lds,
                           hipStream_t st, hipError_t *err)
{
    if (lds > 64u * 1024u) {
        *err = hipFuncSetAttribute(reinterpret_cast<const void *>(&hrt_trace_kernel<LDS, V>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (*err != hipSuccess) return;
    }
    hipLaunchKernelGGL((hrt_trace_kernel<LDS, V>), dim3(blocks), dim3(HRT_BLOCK), lds, st, *P,
                       bounce);
}

// launch `bounce` as ONE kernel (hrt_fused_kernel): one workgroup per macro-chunk of K * 256 entries
#ifndef HRT_FUSED_K0
#define HRT_FUSED_K0 4   /* packets per wave at launch 0 */
#endif
#ifndef HRT_FUSED_KB
#define HRT_FUSED_KB 2   /* ... at later launches (15 words of state per entry stay in registers) */
#endif
template <bool LDS, int V>
static void launch_fused_t(const hrt_kparams *P, uint32_t bounce, size_t lds, hipStream_t st, hipError_t *err)
{
    const uint64_t n_max = (bounce == 0) ? P->n0 : P->cap;
    if (bounce == 0) {
        constexpr int K = HRT_FUSED_K0;
        // (a short launch set: one packet per wave, see the kernel) + the workgroups of the LoS pass
        const uint32_t blocks = (uint32_t)((n_max + (n_max <= kShortList ? 1 : K) * HRT_BLOCK - 1) /
                                           ((n_max <= kShortList ? 1 : K) * HRT_BLOCK)) + P->los_blocks;
        if (lds > 64u * 1024u) {
            *err = hipFuncSetAttribute(reinterpret_cast<const void *>(&hrt_fused_kernel<LDS, V, true, K>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (*err != hipSuccess) return;
        }
        hipLaunchKernelGGL((hrt_fused_kernel<LDS, V, true, K>), dim3(blocks), dim3(HRT_BLOCK), lds, st, *P, bounce);
    } else if constexpr (LDS && (V == 0 || V == 1 || V == 2 || V == 4)) {   // later launches: one-block tables in LDS
        constexpr int K = HRT_FUSED_KB;
        const uint32_t kk = n_max <= kShortList ? 1u : (uint32_t)K;
        const uint32_t blocks = (uint32_t)((n_max + kk * HRT_BLOCK - 1) / (kk * HRT_BLOCK));
        if (lds > 64u * 1024u) {
            *err = hipFuncSetAttribute(reinterpret_cast<const void *>(&hrt_fused_kernel<LDS, V, false, K>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (*err != hipSuccess) return;
        }
        hipLaunchKernelGGL((hrt_fused_kernel<LDS, V, false, K>), dim3(blocks), dim3(HRT_BLOCK), lds, st, *P, bounce);
    } else {
        *err = hipErrorNotSupported;   // later launches are fused on one-block tables in LDS only: the host falls back
    }
}

// launches b0 .. num_bounces as one persistent kernel (hrt_chain_kernel); -1: not on this problem (the caller
// launches them one by one)
template <int V>
static void launch_chain_t(const hrt_kparams *P, uint32_t b0, size_t lds, hipStream_t st, hipError_t *err)
{
    constexpr int K = HRT_FUSED_KB;
    const void *fn = reinterpret_cast<const void *>(&hrt_chain_kernel<true, V, K>);
    // the grid = what is resident at once (asked once per kernel and LDS size; the device is the current one)
    // (per host thread: the drop-in call drives several devices from several threads)
    static thread_local size_t known_lds = ~(size_t)0;
    static thread_local int known_dev = -1;
    static thread_local uint32_t known_grid = 0;
    int dev = 0;
    if ((*err = hipGetDevice(&dev)) != hipSuccess) return;
    if (known_lds != lds || known_dev != dev) {
        if (lds > 64u * 1024u &&
            (*err = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return;
        int per_cu = 0, cus = 0;
        if ((*err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, (int)HRT_BLOCK, lds)) != hipSuccess) return;
        if ((*err = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return;
        if (per_cu < 1 || cus < 1) { *err = hipErrorNotSupported; return; }
        known_grid = (uint32_t)per_cu * (uint32_t)cus;
        known_lds = lds;
        known_dev = dev;
    }
    uint32_t grid = known_grid;
    // (never more workgroups than macro-chunks of one packet per wave the list can have)
    const uint64_t most = (P->cap + HRT_BLOCK - 1) / HRT_BLOCK;
    if (grid > most) grid = (uint32_t)(most ? most : 1u);
    if (grid > kChainMaxGrid) grid = kChainMaxGrid;   // (64 groups of 64: chain_barrier)
    hipLaunchKernelGGL((hrt_chain_kernel<true, V, K>), dim3(grid), dim3(HRT_BLOCK), lds, st, *P, b0);
}

// ===================================================================================
// Export of a rank's compact result as ONE contiguous run of 32-bit words (csrc/host/gather.c; layout in
// include/hrt_device.h): the only exchange step of the sharded path is the gather of these runs to one rank.
// ===================================================================================
struct ExportSeg { unsigned long long src_off; unsigned long long dst_word; unsigned long long n_words; };

// segment-wise copy workspace -> export: blockIdx.y = segment, blockIdx.x strides over its words
__global__ __launch_bounds__(256) void hrt_export_copy_kernel(const uint8_t *ws, const ExportSeg *segs, uint32_t *out)
{
    const ExportSeg g = segs[blockIdx.y];
    const uint32_t *src = reinterpret_cast<const uint32_t *>(ws + g.src_off);
    for (unsigned long long k = (unsigned long long)blockIdx.x * 256u + threadIdx.x; k < g.n_words; k += (unsigned long long)gridDim.x * 256u)
        out[g.dst_word + k] = src[k];
}

// per (bounce, rx) -- blockIdx.x -- the exclusive prefix of the popcounts of its "unblocked" mask words
// (prefix[(b nrx + rx) cap/64 + w]) and their total (totals[b nrx + rx]); one workgroup of 1024 per pair
__global__ __launch_bounds__(1024) void hrt_export_prefix_kernel(const uint8_t *ws, unsigned long long off_counts,
                                                                 unsigned long long off_masks, unsigned long long cap,
                                                                 uint32_t num_rx, uint32_t *prefix, uint32_t *totals)
{
    __shared__ uint32_t part[16];
    __shared__ uint32_t base_s;
    const uint32_t pair = blockIdx.x, b = pair / num_rx, tid = threadIdx.x;
    const uint32_t H = reinterpret_cast<const uint32_t *>(ws + off_counts)[b + 1];
    const uint32_t nw = (H + 63u) / 64u;
    const unsigned long long *mw = reinterpret_cast<const unsigned long long *>(ws + off_masks) + (unsigned long long)pair * (cap / 64u);
    uint32_t *pw = prefix + (unsigned long long)pair * (cap / 64u);
    if (tid == 0) base_s = 0u;
    __syncthreads();
    for (uint32_t w0 = 0; w0 < nw; w0 += 1024u) {
        const uint32_t w = w0 + tid;
        unsigned long long m = w < nw ? mw[w] : 0ull;
        if (w + 1u == nw && (H & 63u)) m &= (1ull << (H & 63u)) - 1ull;   // (bits past the list's end are not the kernels' to define)
        const uint32_t c = (uint32_t)__popcll(m);
        uint32_t all;
        const uint32_t incl = block_scan_incl<16>(c, part, tid, all);
        if (w < nw) pw[w] = base_s + incl - c;
        __syncthreads();
        if (tid == 0) base_s += all;
        __syncthreads();
    }
    if (tid == 0) totals[pair] = base_s;
}

// HRT_EXPORT_UNBLOCKED: per (bounce, rx) -- blockIdx.y -- the unblocked records, compacted in hit order:
// [U] hit indices, then HRT_REC_FIELDS x [U] values, at dst_word[pair] of the export
__global__ __launch_bounds__(256) void hrt_export_compact_kernel(const uint8_t *ws, unsigned long long off_counts,
                                                                 unsigned long long off_masks, unsigned long long off_recs,
                                                                 unsigned long long rec_block_bytes, unsigned long long cap,
                                                                 uint32_t num_rx, const uint32_t *prefix, const uint32_t *totals,
                                                                 const unsigned long long *dst_word, uint32_t *out)
{
    const uint32_t pair = blockIdx.y, b = pair / num_rx, rx = pair - b * num_rx;
    const uint32_t H = reinterpret_cast<const uint32_t *>(ws + off_counts)[b + 1];
    const uint32_t U = totals[pair];
    const unsigned long long *mw = reinterpret_cast<const unsigned long long *>(ws + off_masks) + (unsigned long long)pair * (cap / 64u);
    const uint32_t *pw = prefix + (unsigned long long)pair * (cap / 64u);
    const uint32_t *rec = reinterpret_cast<const uint32_t *>(ws + off_recs + (unsigned long long)b * rec_block_bytes +
                                                             (unsigned long long)rx * 9u * cap * 4u);
    uint32_t *dst = out + dst_word[pair];
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < H; i += gridDim.x * 256u) {
        const unsigned long long m = mw[i >> 6];
        if (!((m >> (i & 63u)) & 1ull)) continue;
        const uint32_t pos = pw[i >> 6] + (uint32_t)__popcll(m & ((1ull << (i & 63u)) - 1ull));
        dst[pos] = i;
#pragma unroll
        for (uint32_t k = 0; k < 9u; ++k) dst[(unsigned long long)(1u + k) * U + pos] = rec[(unsigned long long)k * cap + i];
    }
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
int hrt_hip_host_malloc(void **p, uint64_t bytes) { return (int)hipHostMalloc(p, bytes ? bytes : 1, hipHostMallocDefault); }
int hrt_hip_host_free(void *p) { return p ? (int)hipHostFree(p) : 0; }
int hrt_hip_host_malloc_mapped(void **host, void **dev, uint64_t bytes)   /* page-locked and visible to the device */
{
    int rc = (int)hipHostMalloc(host, bytes ? bytes : 1, hipHostMallocMapped);
    if (!rc) rc = (int)hipHostGetDevicePointer(dev, *host, 0);
    return rc;
}
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
int hrt_hip_d2h_async(void *dst, const void *src, uint64_t bytes, void *stream)
{
    return (int)hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream);
}
int hrt_hip_h2d_async(void *dst, const void *src, uint64_t bytes, void *stream)
{
    return (int)hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream);
}
int hrt_hip_stream_create(void **stream)
{
    hipStream_t s = nullptr;
    const int rc = (int)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    *stream = (void *)s;
    return rc;
}
int hrt_hip_stream_destroy(void *stream) { return (int)hipStreamDestroy((hipStream_t)stream); }
int hrt_hip_mem_info(uint64_t *free_b, uint64_t *total_b)
{
    size_t f = 0, t = 0;
    const int rc = (int)hipMemGetInfo(&f, &t);
    *free_b = f;
    *total_b = t;
    return rc;
}

static uint64_t env_u64(const char *name, uint64_t dflt)
{
    const char *v = getenv(name);
    return (v && *v) ? (uint64_t)atoll(v) : dflt;
}

int hrt_hip_launch_los(const hrt_kparams *P, void *stream)
{
    const uint32_t pairs = P->num_rx * P->num_tx, per_block = HRT_BLOCK / 64u;
    const uint64_t big_min = P->tune.los_big_min_tri;
    if (P->num_tri >= big_min && pairs != 0u && pairs <= 32u) {
        hipLaunchKernelGGL(hrt_los_big_kernel, dim3(pairs * kLosSlices), dim3(HRT_BLOCK), 0, (hipStream_t)stream, *P);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(hrt_los_kernel, dim3((pairs + per_block - 1) / per_block), dim3(HRT_BLOCK), 0,
                       (hipStream_t)stream, *P);
    return (int)hipGetLastError();
}

// does this problem walk the fine leaves (tables beyond LDS with fine spheres built, no big-table trees)?
static bool walks_fine(const hrt_kparams *P)
{
    const int variant = P->tune.variant;
    const uint64_t lds_max = P->tune.lds_tri_bytes_max;
    const uint64_t T = P->num_tri;
    const uint64_t tri_bytes = T * HRT_TRI_FLOATS * 4u + ((T + 1u) / 2u) * 16u +
                               (uint64_t)P->acc.num_leaf * HRT_NODE_FLOATS * 4u;
    const bool in_lds = T * HRT_TRI_FLOATS * 4u <= lds_max && tri_bytes <= 144u * 1024u;
    const bool trees = P->acc.big && (variant >= 4) && variant != 9;
    return !in_lds && !trees && P->acc.fine != nullptr && (variant == 7 || variant == 9);
}

// patch tables on a table that takes the small-table packet kernel: shadow traces + records of launch
// `bounce` >= 1 are hrt_records_kernel's (launched by hrt_hip_launch_trace), the shade kernel skips them
static bool records_in_own_kernel(const hrt_kparams *P, uint32_t bounce)
{
    const int variant = P->tune.variant;
    const uint64_t lds_max = P->tune.lds_tri_bytes_max;
    const uint64_t T = P->num_tri;
    const uint64_t tri_bytes = T * HRT_TRI_FLOATS * 4u + ((T + 1u) / 2u) * 16u +
                               (uint64_t)P->acc.num_leaf * HRT_NODE_FLOATS * 4u;
    const bool in_lds = T * HRT_TRI_FLOATS * 4u <= lds_max && tri_bytes <= 144u * 1024u;
    const bool one_block = P->num_tri <= kMaskRounds * 64u;
    const bool trees = P->acc.big && (variant >= 4) && variant != 9;
    const bool flat = variant == 2 || variant == 3 || (variant == 7 && !trees && one_block);
    return bounce != 0 && P->patch.mask != nullptr && in_lds && flat && one_block && !walks_fine(P);
}

// geometry of launch `bounce`: the trace kernel (all shadow + primary traces of the live list)
int hrt_hip_launch_trace(const hrt_kparams *P_in, uint32_t bounce, void *stream)
{
    hrt_kparams Pc = *P_in;
    Pc.cnt_per_wave = walks_fine(P_in) ? 1u : 0u;
    const hrt_kparams *P = &Pc;
    // Shapes are validated by the host (hrt_trace); here only the launch geometry.
    const uint64_t n_max = (bounce == 0) ? P->n0 : P->cap;
    const uint64_t kinds = (bounce == 0) ? 1u : (bounce < P->num_bounces ? P->num_rx + 1u : P->num_rx);
    uint64_t blocks = ((n_max + HRT_BLOCK - 1) / HRT_BLOCK) * kinds;
    const uint64_t max_grid = P->tune.trace_grid ? P->tune.trace_grid : HRT_TRACE_GRID;
    if (blocks > max_grid) blocks = max_grid;
    if (blocks == 0) blocks = 1;
    // HRT_TRACE_VARIANT: unset = auto (see below).
    // 6 = trees wherever built, else 4; 4 = packet culling behind the leaf spheres + guard; 2 = flat
    // packet culling; 1 = staged tests over all triangles; 0 = the reference's plain sequence.  All
    // give bit-identical results; the GPU tests run all of them.
    const int variant = P->tune.variant;
    const uint64_t T = P->num_tri;
    // staged image: rows + guard pairs (padded to 16 B) + leaf records
    const uint64_t tri_bytes = T * HRT_TRI_FLOATS * 4u + ((T + 1u) / 2u) * 16u +
                               (uint64_t)P->acc.num_leaf * HRT_NODE_FLOATS * 4u;
    const uint64_t lds_max = P->tune.lds_tri_bytes_max;
    const bool in_lds = T * HRT_TRI_FLOATS * 4u <= lds_max && tri_bytes <= 144u * 1024u;
    const bool one_block = P->num_tri <= kMaskRounds * 64u;   // packet culling: single-block build
    const bool fine_pre = !in_lds && !(P->acc.big && variant >= 4 && variant != 9) && P->acc.fine != nullptr &&
                          (variant == 7 || variant == 9);
    const size_t lds = (in_lds ? (size_t)tri_bytes : 0u) + (size_t)P->num_rx * 16u +
                       (HRT_BLOCK / 64u) * (kMaskRounds * 8u + wave_scratch4(fine_pre) * 16u) + 16u;
    hipStream_t st = (hipStream_t)stream;
    hipError_t err = hipSuccess;
    const uint32_t nb = (uint32_t)blocks;
    // auto (7): trees where the host built them; the leaf spheres + guard on tables of more than one
    // block of 1024 triangles (where the live list is re-sorted between bounces, so that most leaves
    // ARE far from a packet: city of 25 002 triangles 22.2 -> 17.9 ms, of 100 002 100 -> 87); the flat
    // walk on small tables (C3: 1.67 vs 1.81 ms)
    const bool trees = P->acc.big && (variant >= 4) && variant != 9;
    const bool flat = variant == 2 || variant == 3 || (variant == 7 && !trees && one_block);
    // fine leaves + plane tree where the host built them (beyond HRT_FINE_MIN_TRI triangles): auto, or variant 9
    const bool fine = !in_lds && !trees && P->acc.fine != nullptr && (variant == 7 || variant == 9);
    if (fine) {
        launch_trace_t<false, 9>(P, bounce, nb, lds, st, &err);
        if (err == hipSuccess && P->wide_cap != 0u) {   // the packets the walk queued as too wide to cull
            const uint64_t wide_grid = P->tune.wide_grid ? P->tune.wide_grid : kWideGrid;
            hipLaunchKernelGGL(hrt_wide_kernel, dim3((uint32_t)wide_grid), dim3(HRT_BLOCK), 0, st, *P, bounce);
            hipLaunchKernelGGL(hrt_wide_finish_kernel, dim3(256), dim3(HRT_BLOCK), 0, st, *P, bounce);
        }
    } else if (in_lds) {
        if (variant == 0) launch_trace_t<true, 0>(P, bounce, nb, lds, st, &err);
        else if (variant == 1) launch_trace_t<true, 1>(P, bounce, nb, lds, st, &err);
        else if (flat) {
            if (records_in_own_kernel(P, bounce)) {
                // patch tables: shadow traces + records are hrt_records_kernel's (hrt_hip_launch_records), this
                // kernel keeps the primary rays
                if (bounce < P->num_bounces) {
                    uint64_t pblocks = (n_max + HRT_BLOCK - 1) / HRT_BLOCK;
                    if (pblocks > max_grid) pblocks = max_grid;
                    if (bounce == 1 && P->patch.num_img != 0u)   // first-order images of the TXs: hrt_image_kernel
                        hipLaunchKernelGGL(hrt_image_kernel, dim3((uint32_t)pblocks), dim3(HRT_BLOCK),
                                           (size_t)T * HRT_TRI_FLOATS * 4u + 16u, st, *P, bounce);
                    else
                        launch_trace_t<true, 2>(P, bounce, (uint32_t)pblocks, lds, st, &err);
                }
            } else if (one_block) launch_trace_t<true, 2>(P, bounce, nb, lds, st, &err);
            else launch_trace_t<true, 3>(P, bounce, nb, lds, st, &err);
        } else if (trees) launch_trace_t<true, 6>(P, bounce, nb, lds, st, &err);
        else if (one_block) launch_trace_t<true, 4>(P, bounce, nb, lds, st, &err);
        else launch_trace_t<true, 5>(P, bounce, nb, lds, st, &err);
    } else {
        if (variant == 0) launch_trace_t<false, 0>(P, bounce, nb, lds, st, &err);
        else if (variant == 1) launch_trace_t<false, 1>(P, bounce, nb, lds, st, &err);
        else if (flat) launch_trace_t<false, 3>(P, bounce, nb, lds, st, &err);
        else if (trees) launch_trace_t<false, 6>(P, bounce, nb, lds, st, &err);
        else launch_trace_t<false, 5>(P, bounce, nb, lds, st, &err);   // tables beyond LDS are > 1 block
    }
    if (err != hipSuccess) return (int)err;
    return (int)hipGetLastError();
}

// shadow traces + scatter records of launch `bounce` >= 1 as their own kernel (patch tables); -1: not this
// problem / launch (then the trace and shade kernels do that work), nothing launched
int hrt_hip_launch_records(const hrt_kparams *P, uint32_t bounce, void *stream)
{
    if (!records_in_own_kernel(P, bounce)) return -1;
    const uint64_t max_grid = P->tune.trace_grid ? P->tune.trace_grid : HRT_TRACE_GRID;
    const uint64_t T = P->num_tri;
    uint64_t rblocks = (P->cap + HRT_BLOCK - 1) / HRT_BLOCK;
    if (rblocks > 2u * max_grid) rblocks = 2u * max_grid;   // (1 024 .. 4 096 workgroups: 2-5 % slower)
    const size_t rlds = (size_t)T * HRT_TRI_FLOATS * 4u + (size_t)P->num_rx * 16u +
                        (size_t)(HRT_NUM_MATERIALS * HRT_MAT_FLOATS * 4u) + (size_t)T * 4u;
    if (rlds > 64u * 1024u) return (int)hipErrorInvalidValue;   // (cannot happen: T <= HRT_PATCH_MAX_TRI)
    hipLaunchKernelGGL(hrt_records_kernel, dim3((uint32_t)rblocks), dim3(HRT_BLOCK), rlds, (hipStream_t)stream, *P, bounce);
    return (int)hipGetLastError();
}

// shading of launch `bounce`: records of bounce-1, the bounce itself, compaction step 1
int hrt_hip_launch_shade(const hrt_kparams *P_in, uint32_t bounce, void *stream)
{
    hrt_kparams Pc = *P_in;
    Pc.cnt_per_wave = walks_fine(P_in) ? 1u : 0u;
    Pc.records_done = records_in_own_kernel(P_in, bounce) ? 1u : 0u;
    if (Pc.records_done && bounce >= Pc.num_bounces) return 0;   // the last launch: records only, nothing left to shade
    const hrt_kparams *P = &Pc;
    const uint64_t n_max = (bounce == 0) ? P->n0 : P->cap;
    uint64_t blocks = (n_max + HRT_BLOCK - 1) / HRT_BLOCK;
    const uint64_t max_grid = P->tune.shade_grid ? P->tune.shade_grid : HRT_SHADE_GRID;
    if (blocks > max_grid) blocks = max_grid;
    if (blocks == 0) blocks = 1;
    const size_t lds0 = (size_t)(HRT_NUM_MATERIALS * HRT_MAT_FLOATS * 4u) + (size_t)P->num_rx * 16u + 32u;
    const uint64_t nlds_off = P->tune.shade_global_normals;
    if (!nlds_off && P->num_tri <= kShadeLdsTri && P->num_mesh <= kShadeLdsMesh)
        hipLaunchKernelGGL(hrt_shade_kernel<true>, dim3((uint32_t)blocks), dim3(HRT_BLOCK),
                           lds0 + ((size_t)P->num_tri + P->num_mesh) * 16u, (hipStream_t)stream, *P, bounce);
    else
        hipLaunchKernelGGL(hrt_shade_kernel<false>, dim3((uint32_t)blocks), dim3(HRT_BLOCK), lds0,
                           (hipStream_t)stream, *P, bounce);
    return (int)hipGetLastError();
}

int hrt_hip_launch_fused(const hrt_kparams *P_in, uint32_t bounce, void *stream)
{
    hrt_kparams Pc = *P_in;
#ifdef HRT_PHASE_STATS
    Pc.phase_bounce = (uint32_t)env_u64("HRT_PHASE_BOUNCE", 0);
#endif
    Pc.records_done = records_in_own_kernel(P_in, bounce) ? 1u : 0u;
    if (bounce != 0) Pc.los_blocks = 0;
    else if (Pc.los_blocks) Pc.los_blocks = (Pc.num_rx * Pc.num_tx + HRT_BLOCK / 64u - 1u) / (HRT_BLOCK / 64u);
    const hrt_kparams *P = &Pc;
    if (P->cap / HRT_BLOCK + 1u > P->lb_chunks) return (int)hipErrorInvalidValue;   // (one word per chunk)
    const int variant = P->tune.variant;
    const uint64_t T = P->num_tri;
    const uint64_t tri_bytes = T * HRT_TRI_FLOATS * 4u + ((T + 1u) / 2u) * 16u +
                               (uint64_t)P->acc.num_leaf * HRT_NODE_FLOATS * 4u;
    const uint64_t lds_max = P->tune.lds_tri_bytes_max;
    const bool in_lds = T * HRT_TRI_FLOATS * 4u <= lds_max && tri_bytes <= 144u * 1024u;
    const bool one_block = P->num_tri <= kMaskRounds * 64u;
    const bool fine_pre = !in_lds && !(P->acc.big && variant >= 4 && variant != 9) && P->acc.fine != nullptr &&
                          (variant == 7 || variant == 9);
    const size_t lds = (in_lds ? (size_t)tri_bytes : 0u) + (size_t)P->num_rx * 16u +
                       (HRT_BLOCK / 64u) * (kMaskRounds * 8u + wave_scratch4(fine_pre) * 16u) + 64u * 4u +
                       (size_t)(HRT_NUM_MATERIALS * HRT_MAT_FLOATS * 4u) + kLdsTx * 16u;
    hipStream_t st = (hipStream_t)stream;
    hipError_t err = hipSuccess;
    // the same choice of intersection loop as hrt_hip_launch_trace; on tables of a handful of
    // triangles (auto) the staged walk over all of them is cheaper than a culling round
    const uint64_t staged_max = P->tune.fuse_staged_max_tri;
    const bool trees = P->acc.big && (variant >= 4) && variant != 9;
    const bool flat = variant == 2 || variant == 3 || (variant == 7 && !trees && one_block);
    const bool staged = variant == 1 || (variant == 7 && T <= staged_max);
    const bool fine = !in_lds && !trees && P->acc.fine != nullptr && (variant == 7 || variant == 9);
    // (the fine walk wants the trace kernel's registers and its queue of wide packets: fused, launch 0 of
    // the 100 002-triangle city took 6.4 ms against 2.1 + 0.1 for the two kernels)
    if (fine) return -1;
    else if (in_lds) {
        if (variant == 0) launch_fused_t<true, 0>(P, bounce, lds, st, &err);
        else if (staged) launch_fused_t<true, 1>(P, bounce, lds, st, &err);
        else if (flat) {
            if (one_block) launch_fused_t<true, 2>(P, bounce, lds, st, &err);
            else launch_fused_t<true, 3>(P, bounce, lds, st, &err);
        } else if (trees) launch_fused_t<true, 6>(P, bounce, lds, st, &err);
        else if (one_block) launch_fused_t<true, 4>(P, bounce, lds, st, &err);
        else launch_fused_t<true, 5>(P, bounce, lds, st, &err);
    } else {
        if (variant == 0) launch_fused_t<false, 0>(P, bounce, lds, st, &err);
        else if (staged) launch_fused_t<false, 1>(P, bounce, lds, st, &err);
        else if (flat) launch_fused_t<false, 3>(P, bounce, lds, st, &err);
        else if (trees) launch_fused_t<false, 6>(P, bounce, lds, st, &err);
        else launch_fused_t<false, 5>(P, bounce, lds, st, &err);
    }
    if (err == hipErrorNotSupported) return -1;   // not a HIP failure: this launch runs as two kernels
    if (err != hipSuccess) return (int)err;
    return (int)hipGetLastError();
}

int hrt_hip_launch_chain(const hrt_kparams *P_in, uint32_t b0, void *stream)
{
    if (b0 == 0 || b0 > P_in->num_bounces) return -1;
    hrt_kparams Pc = *P_in;
    Pc.records_done = 0u;
    Pc.los_blocks = 0;
    const hrt_kparams *P = &Pc;
    if (records_in_own_kernel(P_in, b0)) return -1;
    if (P->cap / HRT_BLOCK + 1u > P->lb_chunks) return (int)hipErrorInvalidValue;
    const int variant = P->tune.variant;
    const uint64_t T = P->num_tri;
    const uint64_t tri_bytes = T * HRT_TRI_FLOATS * 4u + ((T + 1u) / 2u) * 16u +
                               (uint64_t)P->acc.num_leaf * HRT_NODE_FLOATS * 4u;
    const bool in_lds = T * HRT_TRI_FLOATS * 4u <= P->tune.lds_tri_bytes_max && tri_bytes <= 144u * 1024u;
    const bool one_block = P->num_tri <= kMaskRounds * 64u;
    if (!in_lds || !one_block || walks_fine(P)) return -1;
    const size_t lds = (size_t)tri_bytes + (size_t)P->num_rx * 16u +
                       (HRT_BLOCK / 64u) * (kMaskRounds * 8u + wave_scratch4(false) * 16u) + 64u * 4u +
                       (size_t)(HRT_NUM_MATERIALS * HRT_MAT_FLOATS * 4u) + kLdsTx * 16u;
    const bool trees = P->acc.big && (variant >= 4) && variant != 9;
    const bool flat = variant == 2 || variant == 3 || (variant == 7 && !trees && one_block);
    const bool staged = variant == 1 || (variant == 7 && T <= P->tune.fuse_staged_max_tri);
    hipStream_t st = (hipStream_t)stream;
    hipError_t err = hipSuccess;
    if (variant == 0) launch_chain_t<0>(P, b0, lds, st, &err);
    else if (staged) launch_chain_t<1>(P, b0, lds, st, &err);
    else if (flat) launch_chain_t<2>(P, b0, lds, st, &err);
    else if (trees) return -1;
    else launch_chain_t<4>(P, b0, lds, st, &err);
    if (err == hipErrorNotSupported) return -1;
    if (err != hipSuccess) return (int)err;
    return (int)hipGetLastError();
}

int hrt_hip_launch_dirs(uint64_t num_paths, uint32_t rank, uint32_t count, uint32_t chunk,
                        uint64_t num_local, float *d_dirs, uint32_t *d_fix_count,
                        uint32_t *d_fix_list, uint32_t fix_cap, void *stream)
{
    if (num_local == 0) return 0;
    hipLaunchKernelGGL(hrt_launch_dirs_kernel, dim3((uint32_t)((num_local + 255) / 256)), dim3(256),
                       0, (hipStream_t)stream, num_paths, rank, count, chunk, num_local, d_dirs,
                       d_fix_count, d_fix_list, fix_cap);
    return (int)hipGetLastError();
}

uint64_t hrt_hip_sort_temp_bytes(uint64_t cap)
{
    const uint64_t tiles = (cap + kSortTile - 1u) / kSortTile;
    return ((uint64_t)kSortBins * tiles + kSortBins) * 4u;   // hist[bins][tiles] + total[bins]
}

int hrt_hip_sort_hits(const hrt_kparams *P, uint32_t bounce, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    const uint32_t blocks = (uint32_t)((P->cap + 255) / 256);
    hipLaunchKernelGGL(hrt_sort_keys_kernel, dim3(blocks), dim3(256), 0, st, *P, bounce);
    uint32_t *keys = reinterpret_cast<uint32_t *>(P->ws + P->sort.off_keys);
    SortArgs A;
    A.key_a = keys; A.key_b = keys + P->cap; A.idx_a = keys + 2 * P->cap; A.idx_b = keys + 3 * P->cap;
    A.tiles_max = (uint32_t)((P->cap + kSortTile - 1u) / kSortTile);
    A.hist = reinterpret_cast<uint32_t *>(P->ws + P->sort.off_tmp);
    A.total = A.hist + (uint64_t)kSortBins * A.tiles_max;
    A.count = reinterpret_cast<const uint32_t *>(P->ws + P->off_counts) + bounce + 1u;
    if (((uint64_t)kSortBins * A.tiles_max + kSortBins) * 4u > P->sort.tmp_bytes) return (int)hipErrorInvalidValue;
    // digits of up to 11 bits: 2 passes up to 22 key bits (the usual key has 20), 3 beyond
    const uint32_t kb = P->sort.key_bits ? P->sort.key_bits : 1u;
    const uint32_t passes = kb <= 11u ? 1u : (kb <= 22u ? 2u : 3u);
    const uint32_t bits = (kb + passes - 1u) / passes;
    for (uint32_t p = 0; p < passes; ++p) {
        A.shift = bits * p;
        A.bits = bits;
        A.parity = p & 1u;
        hipLaunchKernelGGL(hrt_sort_hist_kernel, dim3(A.tiles_max), dim3(256), 0, st, A);
        hipLaunchKernelGGL(hrt_sort_scan_kernel, dim3(1u << bits), dim3(1024), 0, st, A);
        hipLaunchKernelGGL(hrt_sort_scatter_kernel, dim3(A.tiles_max), dim3(256), 0, st, A);
    }
    // the sorted indices: buffer B after an odd number of passes, A after an even one
    hipLaunchKernelGGL(hrt_sort_permute_kernel, dim3(blocks), dim3(256), 0, st, *P, bounce,
                       (const uint32_t *)((passes & 1u) ? A.idx_b : A.idx_a));
    return (int)hipGetLastError();
}

int hrt_hip_rxt_build(const float *d_tri, uint32_t num_tri, const float *d_rx_pos, uint32_t num_rx,
                      const float *d_bin_dir4, const float *d_bin_cs2, const float *d_ro_bin,
                      float cx, float cy, float cz, float region_r, unsigned long long *d_masks, void *stream)
{
    hipLaunchKernelGGL(hrt_rxt_build_kernel, dim3(HRT_RXT_BINS, num_rx), dim3(64), 0, (hipStream_t)stream,
                       d_tri, num_tri, d_rx_pos, d_bin_dir4, d_bin_cs2, d_ro_bin, cx, cy, cz, region_r, d_masks);
    return (int)hipGetLastError();
}

int hrt_hip_txcell_build(const float *d_tri, uint32_t num_tri, const float *d_tx_pos, uint32_t num_tx, const float *d_bin_dir4,
                         const float *d_bin_cs2, unsigned long long *d_masks, void *stream)
{
    if (num_tx == 0 || num_tx > 65535u) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(hrt_txcell_build_kernel, dim3(HRT_RXT_BINS, num_tx), dim3(64), 0, (hipStream_t)stream, d_tri, num_tri,
                       d_tx_pos, d_bin_dir4, d_bin_cs2, d_masks);
    return (int)hipGetLastError();
}

int hrt_hip_patch_build(const float *d_tri, uint32_t num_tri, const float *d_pdef, const uint32_t *d_patch_tri,
                        uint32_t num_patch, const float *d_apex, uint32_t num_rx, uint32_t num_img, float hball,
                        float ro_rx, float ro_img, unsigned long long *d_masks, void *stream)
{
    if (num_patch == 0 || num_rx + num_img == 0 || num_rx + num_img > 65535u) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(hrt_patch_build_kernel, dim3((num_patch + 255u) / 256u, num_rx + num_img), dim3(256), 0,
                       (hipStream_t)stream, d_tri, num_tri, d_pdef, d_patch_tri, num_patch, d_apex, num_rx, hball,
                       ro_rx, ro_img, d_masks);
    return (int)hipGetLastError();
}

int hrt_hip_export_copy(const void *d_ws, const void *d_segs, uint32_t num_segs, uint64_t max_words, void *d_out, void *stream)
{
    if (num_segs == 0) return 0;
    uint64_t bx = (max_words + 255u) / 256u;
    if (bx > 1024u) bx = 1024u;
    if (bx == 0) bx = 1;
    hipLaunchKernelGGL(hrt_export_copy_kernel, dim3((uint32_t)bx, num_segs), dim3(256), 0, (hipStream_t)stream,
                       (const uint8_t *)d_ws, (const ExportSeg *)d_segs, (uint32_t *)d_out);
    return (int)hipGetLastError();
}
int hrt_hip_export_prefix(const void *d_ws, uint64_t off_counts, uint64_t off_masks, uint64_t cap, uint32_t num_bounces,
                          uint32_t num_rx, uint32_t *d_prefix, uint32_t *d_totals, void *stream)
{
    hipLaunchKernelGGL(hrt_export_prefix_kernel, dim3(num_bounces * num_rx), dim3(1024), 0, (hipStream_t)stream,
                       (const uint8_t *)d_ws, off_counts, off_masks, cap, num_rx, d_prefix, d_totals);
    return (int)hipGetLastError();
}
int hrt_hip_export_compact(const void *d_ws, uint64_t off_counts, uint64_t off_masks, uint64_t off_recs, uint64_t rec_block_bytes,
                           uint64_t cap, uint32_t num_bounces, uint32_t num_rx, uint64_t max_hits, const uint32_t *d_prefix,
                           const uint32_t *d_totals, const uint64_t *d_dst_word, void *d_out, void *stream)
{
    uint64_t bx = (max_hits + 255u) / 256u;
    if (bx > 2048u) bx = 2048u;
    if (bx == 0) return 0;
    hipLaunchKernelGGL(hrt_export_compact_kernel, dim3((uint32_t)bx, num_bounces * num_rx), dim3(256), 0, (hipStream_t)stream,
                       (const uint8_t *)d_ws, off_counts, off_masks, off_recs, rec_block_bytes, cap, num_rx, d_prefix, d_totals,
                       (const unsigned long long *)d_dst_word, (uint32_t *)d_out);
    return (int)hipGetLastError();
}
int hrt_hip_d2d_async(void *dst, const void *src, uint64_t bytes, void *stream)
{
    return (int)hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream);
}

int hrt_hip_launch_fs0(const float *d_dirs, uint64_t n, const float *tx_vel3, float mult, float *d_out, void *stream)
{
    if (n == 0) return 0;
    hipLaunchKernelGGL(hrt_fs0_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       d_dirs, n, tx_vel3[0], tx_vel3[1], tx_vel3[2], mult, d_out);
    return (int)hipGetLastError();
}

int hrt_hip_launch_order(const uint32_t *d_seg_start, const uint32_t *d_seg_band, uint32_t num_seg,
                         uint64_t num_paths, uint32_t rank, uint32_t count, uint32_t chunk,
                         uint32_t *d_order, void *stream)
{
    if (num_seg == 0) return 0;
    {   // per launch: the attribute is per device, and with HRT_DEVICES one host thread per device comes here
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&hrt_launch_order_kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)(HRT_ORDER_SEG * 4u));
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(hrt_launch_order_kernel, dim3(num_seg), dim3(1024), HRT_ORDER_SEG * 4u,
                       (hipStream_t)stream, d_seg_start, d_seg_band, num_paths, rank, count, chunk, d_order);
    return (int)hipGetLastError();
}

int hrt_hip_selftest_math(int fn, const float *d_in, float *d_out, uint64_t n, void *stream)
{
    if (n == 0) return 0;
    hipLaunchKernelGGL(hrt_selftest_math_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, fn, d_in, d_out, n);
    return (int)hipGetLastError();
}

int hrt_hip_read_stats(unsigned long long *out, int reset)
{
#ifdef HRT_PHASE_STATS
    if (const char *pf = getenv("HRT_PHASE_FILE")) {
        static unsigned long long host_phase[65536][8];
        if (hipMemcpyFromSymbol(host_phase, HIP_SYMBOL(g_phase), sizeof host_phase) == hipSuccess) {
            if (FILE *f = fopen(pf, "wb")) { fwrite(host_phase, 1, sizeof host_phase, f); fclose(f); }
        }
    }
#endif
#ifdef HRT_UNIT_CLOCKS
    if (const char *uf = getenv("HRT_UNIT_FILE")) {
        static unsigned long long host_unit[1u << 21][2];
        unsigned int n = 0;
        if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_unit_n), sizeof n) == hipSuccess &&
            hipMemcpyFromSymbol(host_unit, HIP_SYMBOL(g_unit), sizeof host_unit) == hipSuccess) {
            if (n > (1u << 21)) n = 1u << 21;
            if (FILE *f = fopen(uf, "wb")) { fwrite(host_unit, 16, n, f); fclose(f); }
        }
        if (reset) {
            n = 0;
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_unit_n), &n, sizeof n);
            void *gu = nullptr;
            if (hipGetSymbolAddress(&gu, HIP_SYMBOL(g_unit)) == hipSuccess) (void)hipMemset(gu, 0, sizeof(unsigned long long) * 2u << 21);
        }
    }
#endif
#if defined(HRT_KERNEL_STATS) || defined(HRT_PHASE_STATS)
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stats), sizeof(unsigned long long) * 3 * HRT_STATS_COLS);
    if (e != hipSuccess) return (int)e;
    if (reset) {
        unsigned long long z[3 * HRT_STATS_COLS] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_stats), z, sizeof z);
    }
    return (int)e;
#else
    (void)reset;
    for (int i = 0; i < 3 * HRT_STATS_COLS; ++i) out[i] = 0;
    return 0;
#endif
}

int hrt_hip_event_create(void **ev)
{
    hipEvent_t e;
    const int rc = (int)hipEventCreate(&e);
    *ev = (void *)e;
    return rc;
}
int hrt_hip_event_create_sync(void **ev)   /* ordering only (no time stamps): cheaper to record */
{
    hipEvent_t e;
    const int rc = (int)hipEventCreateWithFlags(&e, hipEventDisableTiming);
    *ev = (void *)e;
    return rc;
}
int hrt_hip_stream_wait_event(void *stream, void *ev)
{
    return (int)hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0);
}
int hrt_hip_event_destroy(void *ev) { return (int)hipEventDestroy((hipEvent_t)ev); }
int hrt_hip_event_record(void *ev, void *stream)
{
    return (int)hipEventRecord((hipEvent_t)ev, (hipStream_t)stream);
}
int hrt_hip_event_sync(void *ev) { return (int)hipEventSynchronize((hipEvent_t)ev); }
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
