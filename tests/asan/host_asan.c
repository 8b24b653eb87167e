/* Sanitizer driver for the host-only C of the product: scene_io.c, sionna_import.c,
 * materials.c, accel.c -- the files that parse untrusted input or do hand-rolled offset arithmetic
 * -- compiled together with this file under -fsanitize=address,undefined (tests/asan/Makefile;
 * there is no GPU AddressSanitizer on the pool, and none of this code touches the device).
 *
 *   host_asan load FILE         scene_load(FILE), round trip through scene_save, free
 *                               (exit 8 = the loader's own "bad file" exit, the reference's code)
 *   host_asan sionna XML        hrt_scene_import_sionna(XML): 0 or a clean error, then free
 *   host_asan accel T SEED      hrt_accel_order + hrt_accel_build on T random rows (NaNs, zero-area
 *                               and duplicated triangles included), big mode forced
 *   host_asan eta               the eta table of all 17 materials at three frequencies
 * Any sanitizer report aborts with a non-zero status that is neither 0 nor 8. */
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hrt_internal.h"

/* the two helpers these files take from problem.c (which needs the HIP shim) */
static __thread char g_err[512];
int hrt_fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
const char *hrt_last_error(void) { return g_err; }

static uint32_t rnd(uint64_t *s)
{
    *s = *s * 6364136223846793005ull + 1442695040888963407ull;
    return (uint32_t)(*s >> 33);
}
static float frand(uint64_t *s, float lo, float hi) { return lo + (hi - lo) * (float)(rnd(s) & 0xffffff) / 16777216.f; }

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    if (!strcmp(argv[1], "load") && argc >= 3) {
        Scene sc = scene_load(argv[2]);
        char out[4096];
        snprintf(out, sizeof out, "%s.roundtrip", argv[2]);
        scene_save(&sc, out);
        Scene sc2 = scene_load(out);
        if (sc2.num_meshes != sc.num_meshes) return 3;
        free_scene(&sc);
        free_scene(&sc2);
        remove(out);
        return 0;
    }
    if (!strcmp(argv[1], "sionna") && argc >= 3) {
        Scene sc;
        memset(&sc, 0, sizeof sc);
        const int rc = hrt_scene_import_sionna(argv[2], &sc);
        if (rc == HRT_OK) free_scene(&sc);
        else if (!hrt_last_error()[0]) return 4;   /* an error without a message */
        return 0;
    }
    if (!strcmp(argv[1], "accel") && argc >= 4) {
        const uint32_t T = (uint32_t)atoi(argv[2]);
        uint64_t s = (uint64_t)atoll(argv[3]) * 2654435761u + 1;
        float *rows = (float *)calloc((size_t)(T ? T : 1) * HRT_TRI_FLOATS, sizeof(float));
        if (!rows) return 5;
        for (uint32_t j = 0; j < T; ++j) {
            float *r = rows + (size_t)j * HRT_TRI_FLOATS;
            for (int k = 0; k < 3; ++k) r[k] = frand(&s, -100.f, 100.f);
            for (int k = 3; k < 9; ++k) r[k] = frand(&s, -5.f, 5.f);
            const uint32_t what = rnd(&s) % 50u;
            if (what == 0) r[rnd(&s) % 9u] = NAN;
            if (what == 1) r[rnd(&s) % 9u] = INFINITY;
            if (what == 2) { r[6] = r[3]; r[7] = r[4]; r[8] = r[5]; }   /* zero area */
            if (what == 3 && j) memcpy(r, r - HRT_TRI_FLOATS, HRT_TRI_FLOATS * sizeof(float));   /* duplicate */
            const float c[3] = {r[4] * r[8] - r[5] * r[7], r[5] * r[6] - r[3] * r[8], r[3] * r[7] - r[4] * r[6]};
            const float len = sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
            r[9] = c[0] / len; r[10] = c[1] / len; r[11] = c[2] / len;
        }
        setenv("HRT_TUNE", "accel_big=0", 1);   /* inner levels + plane tree on every table */
        hrt_tune tune;
        if (hrt_tune_load(&tune)) return 12;
        hrt_accel a;
        int rc = hrt_accel_order(&a, rows, T, 1);
        if (rc) return 6;
        float *tab = (float *)calloc((size_t)(T ? T : 1) * HRT_TRI_FLOATS, sizeof(float));
        if (!tab) return 5;
        for (uint32_t k = 0; k < T; ++k) {
            if (a.orig[k] >= T || a.newidx[a.orig[k]] != k) return 7;   /* a permutation and its inverse */
            memcpy(tab + (size_t)k * HRT_TRI_FLOATS, rows + (size_t)a.orig[k] * HRT_TRI_FLOATS, HRT_TRI_FLOATS * sizeof(float));
        }
        rc = hrt_accel_build(&a, tab, &tune);
        if (rc) return 9;
        if (T && (!a.big || a.pl_levels == 0 || a.pl_count[a.pl_levels - 1] > 64)) return 10;
        /* every row appears exactly once in the plane tree's index */
        if (T) {
            unsigned char *seen = (unsigned char *)calloc(T, 1);
            for (uint32_t i = 0; i < a.pl_num_leaf * 64u; ++i) {
                const uint32_t j = a.pl_index[i];
                if (j == HRT_NO_HIT) continue;
                if (j >= T || seen[j]) return 11;
                seen[j] = 1;
            }
            for (uint32_t j = 0; j < T; ++j) if (!seen[j]) return 12;
            free(seen);
        }
        hrt_accel_free(&a);
        free(rows); free(tab);
        return 0;
    }
    if (!strcmp(argv[1], "eta")) {
        const float f[3] = {3.0f, 3.5f, 70.f};
        double acc = 0;
        for (int k = 0; k < 3; ++k)
            for (uint32_t m = 0; m < HRT_NUM_MATERIALS; ++m) {
                hrt_eta e;
                hrt_material_eta(m, f[k], &e);
                acc += e.eta_abs;
            }
        return acc > 0 ? 0 : 13;
    }
    return 2;
}
