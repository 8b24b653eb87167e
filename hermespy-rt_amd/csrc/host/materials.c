/* materials.c -- radio material parameters and the per-frequency permittivity precompute.
 *
 * The numbers are ITU-R P.2040-3 table 3 (a, b, c, d) plus the reference's scattering
 * parameters; they must equal src/materials.c:3-89 of the reference value for value (the
 * index is what .hrt files store).  Only the fields the hot path reads are kept
 * (s1, s2, s3, s3_alpha and the names are unused by compute_paths).
 */
#include <math.h>

#include "hrt_internal.h"

const hrt_material hrt_materials[HRT_NUM_MATERIALS] = {
    [0]  = {1.f,    0.f,   0.f,        0.001f,  0.1f, 2},   /* air */
    [1]  = {5.24f,  0.f,   0.0462f,    0.7822f, 0.5f, 4},   /* concrete */
    [2]  = {3.91f,  0.f,   0.0238f,    0.16f,   0.4f, 3},   /* brick */
    [3]  = {2.73f,  0.f,   0.0085f,    0.9395f, 0.3f, 3},   /* plasterboard */
    [4]  = {1.99f,  0.f,   0.0047f,    1.0718f, 0.2f, 2},   /* wood */
    [5]  = {6.31f,  0.f,   0.0036f,    1.3394f, 0.3f, 3},   /* glass 1 */
    [6]  = {5.79f,  0.f,   0.0004f,    1.658f,  0.3f, 3},   /* glass 2 */
    [7]  = {1.48f,  0.f,   0.0011f,    1.0750f, 0.2f, 2},   /* ceiling board 1 */
    [8]  = {1.52f,  0.f,   0.0029f,    1.029f,  0.2f, 2},   /* ceiling board 2 */
    [9]  = {2.58f,  0.f,   0.0217f,    0.7800f, 0.4f, 3},   /* chipboard */
    [10] = {2.71f,  0.f,   0.33f,      0.f,     0.3f, 3},   /* plywood */
    [11] = {7.074f, 0.f,   0.0055f,    0.9262f, 0.3f, 3},   /* marble */
    [12] = {3.66f,  0.f,   0.0044f,    1.3515f, 0.3f, 3},   /* floorboard */
    [13] = {1.f,    0.f,   10000000.f, 0.f,     0.f,  1},   /* metal */
    [14] = {3.f,    0.f,   0.00015f,   2.52f,   0.4f, 4},   /* very dry ground */
    [15] = {15.f,  -0.1f,  0.035f,     1.63f,   0.5f, 4},   /* medium dry ground */
    [16] = {30.f,  -0.4f,  0.15f,      1.30f,   0.5f, 4},   /* wet ground */
};

/* Complex square root given the modulus (src/compute_paths.c:136-151). */
static void sqrt_c(float re, float im, float mod, float *out_re, float *out_im)
{
    const float eps = 1.1920928955078125e-07f;
    *out_re = sqrtf((re + mod) / 2.f);
    if (fabsf(im) < eps && re >= -eps) {
        *out_im = 0.f;
        return;
    }
    float v = sqrtf((mod - re) / 2.f);
    *out_im = im < 0.f ? -v : v;
}

/* eta(f) and its derived quantities for one material (src/compute_paths.c:183-204). */
void hrt_material_eta(uint32_t mi, float f_ghz, hrt_eta *e)
{
    const hrt_material *m = &hrt_materials[mi];
    e->eta_re = m->a * powf(f_ghz, m->b);
    /* eq. 12: sigma / (2 pi eps0 f) with the constant folded as in the reference */
    e->eta_im = (m->c * powf(f_ghz, m->d)) / (0.0556325027352135f * f_ghz);
    e->eta_abs_pow2 = e->eta_re * e->eta_re + e->eta_im * e->eta_im;
    e->eta_abs = sqrtf(e->eta_abs_pow2);
    e->eta_abs_inv_sqrt = 1.f / sqrtf(e->eta_abs);
    sqrt_c(e->eta_re, e->eta_im, e->eta_abs, &e->eta_sqrt_re, &e->eta_sqrt_im);
    e->eta_inv_re = e->eta_re / e->eta_abs_pow2;
    e->eta_inv_im = -e->eta_im / e->eta_abs_pow2;
    sqrt_c(e->eta_inv_re, e->eta_inv_im, 1.f / e->eta_abs, &e->eta_inv_sqrt_re,
           &e->eta_inv_sqrt_im);
    e->r = 1.f - m->s;
}
