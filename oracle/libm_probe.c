/* libm_probe.c -- pins hermespy-rt_amd/csrc/hrt_libm.h against the HOST libm.
 *
 * TEST INFRASTRUCTURE (oracle/): compiles the product's device math header for the host and
 * compares hrt_sinf/hrt_cosf/hrt_sincosf/hrt_cosf_nb/hrt_expf/hrt_acosf with the libm this machine's reference build
 * would call, bit for bit, over the domain the tracer can produce.
 *
 *   libm_probe            every 1009th float of each domain (about 2 s)
 *   libm_probe --full     EVERY float of each domain (about 3 min on one core)
 * Exit status 0 iff there is no mismatch.  NaN results compare equal to NaN.
 *
 * Build: gcc -O2 -mfma -ffp-contract=off (the header's fma() calls must be real fused
 * operations and nothing else may be contracted).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../hermespy-rt_amd/csrc/hrt_libm.h"

typedef float (*fn)(float);

static unsigned long long sweep(const char *name, fn cand, fn ref, float lo, float hi,
                                uint32_t stride)
{
    unsigned long long bad = 0, n = 0;
    const uint32_t a = hrt_f2u(lo), b = hrt_f2u(hi);
    float first = 0.f;
    for (uint64_t u = a; u < b; u += stride)
        for (int neg = 0; neg < 2; ++neg) {
            const float x = hrt_u2f((uint32_t)u | (neg ? 0x80000000u : 0u));
            const float r = ref(x), c = cand(x);
            ++n;
            if (hrt_f2u(r) != hrt_f2u(c) && !(r != r && c != c)) {
                if (!bad) first = x;
                ++bad;
            }
        }
    printf("%-9s |x| in [%g, %g)  checked %llu  mismatches %llu", name, lo, hi, n, bad);
    if (bad) printf("  first at %a", first);
    printf("\n");
    return bad;
}

static float sc_sin(float x) { float s, c; hrt_sincosf(x, &s, &c); return s; }
static float sc_cos(float x) { float s, c; hrt_sincosf(x, &s, &c); return c; }

int main(int argc, char **argv)
{
    const uint32_t stride = (argc > 1 && !strcmp(argv[1], "--full")) ? 1u : 1009u;
    unsigned long long bad = 0;
    bad += sweep("sinf", hrt_sinf, sinf, 0.f, 120.f, stride);
    bad += sweep("cosf", hrt_cosf, cosf, 0.f, 120.f, stride);
    bad += sweep("sincosf.s", sc_sin, sinf, 0.f, 120.f, stride);
    bad += sweep("sincosf.c", sc_cos, cosf, 0.f, 120.f, stride);
    bad += sweep("cosf_nb", hrt_cosf_nb, cosf, 0.f, 120.f, stride);
    bad += sweep("expf", hrt_expf, expf, 0.f, 88.f, stride);
    bad += sweep("acosf", hrt_acosf, acosf, 0.f, 1.0000001f, stride);
    printf("%s\n", bad ? "MISMATCH" : "OK: device math header == host libm on every probed input");
    return bad ? 1 : 0;
}
