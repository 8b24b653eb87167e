/* tune.c -- the developer / test switches of the library, in ONE environment variable:
 *
 *     HRT_TUNE="key=value,key=value,..."
 *
 * read once per problem (hrt_problem_create) into a struct that travels with the problem and its traces, so
 * that two problems of one process may differ and nothing is latched in function-local statics.  None of
 * these changes a result; they select code paths that the test-suite wants to force (every intersection
 * variant must give the same bits) and a few sizes.  The SUPPORTED variables are the dozen in
 * INTEGRATION.md; an unknown key here is an error (HRT_E_INVALID), not silently ignored. */
#include <stdlib.h>
#include <string.h>

#include "hrt_internal.h"

void hrt_tune_defaults(hrt_tune *t)
{
    memset(t, 0, sizeof *t);
    t->k.variant = HRT_TRACE_VARIANT_DEFAULT;
    t->k.lds_tri_bytes_max = HRT_LDS_TRI_BYTES_MAX;
    t->k.trace_grid = HRT_TRACE_GRID;
    t->k.shade_grid = HRT_SHADE_GRID;
    t->k.wide_grid = 0;             /* 0: the kernels' default */
    t->k.los_big_min_tri = 16384;   /* LoS pass sliced over 256 waves per pair from here on (100 k triangles: 0.8 -> 0.05 ms) */
    t->k.fuse_staged_max_tri = 6;   /* fused launches: the staged walk over all rows up to here (a culling round costs more) */
    t->k.shade_global_normals = 0;
    t->k.lb_max_polls = 8000;       /* ~10 ms; the wait is microseconds when the kernel owns the GPU (hrt_kernels.hip: lb_exclusive) */
    t->wide_cos = 0.0;              /* 0: HRT_WIDE_COS / HRT_WIDE_COS_BIG by table size (hrt_kparams.h holds the measurements) */
    t->wide_cap = -1;
    t->sort_rays = -1;              /* -1: tables beyond HRT_SORT_MIN_TRI */
    t->sort_fine = 6;
    t->sort_dir_res = 2;
    t->rxt_min_rays = UINT64_MAX;   /* UINT64_MAX: by table size (problem.c) */
    t->rxt_max_tri = HRT_RXT_MAX_TRI;
    t->patch_size = 0.5;
    t->accel_sparse = HRT_ACCEL_SPARSE;
    t->accel_big = HRT_ACCEL_BIG;
    t->accel_fine = -1;
    t->accel_fine_min = UINT64_MAX;
}

int hrt_tune_load(hrt_tune *t)
{
    hrt_tune_defaults(t);
    const char *env = getenv("HRT_TUNE");
    if (!env || !*env) return HRT_OK;
    char *buf = strdup(env);
    if (!buf) return hrt_fail(HRT_E_NOMEM, "out of host memory");
    int rc = HRT_OK;
    char *save = NULL;
    for (char *tok = strtok_r(buf, ",; ", &save); tok && !rc; tok = strtok_r(NULL, ",; ", &save)) {
        char *eq = strchr(tok, '=');
        const char *val = eq ? eq + 1 : "1";
        if (eq) *eq = 0;
        const double v = atof(val);
        const unsigned long long u = strtoull(val, NULL, 10);
#define KEY(name) (strcmp(tok, name) == 0)
        if KEY("variant") t->k.variant = (int32_t)v;
        else if KEY("lds_tri_bytes") t->k.lds_tri_bytes_max = u;
        else if KEY("trace_grid") t->k.trace_grid = (uint32_t)u;
        else if KEY("shade_grid") t->k.shade_grid = (uint32_t)u;
        else if KEY("wide_grid") t->k.wide_grid = (uint32_t)u;
        else if KEY("los_big_min_tri") t->k.los_big_min_tri = (uint32_t)u;
        else if KEY("fuse_staged_max_tri") t->k.fuse_staged_max_tri = (uint32_t)u;
        else if KEY("shade_global_normals") t->k.shade_global_normals = (uint32_t)u;
        else if KEY("lb_max_polls") t->k.lb_max_polls = (uint32_t)u;
        else if KEY("wide_cos") t->wide_cos = v;
        else if KEY("wide_cap") t->wide_cap = (int64_t)strtoll(val, NULL, 10);
        else if KEY("sort_rays") t->sort_rays = (int)v;
        else if KEY("sort_fine") t->sort_fine = (int)v;
        else if KEY("sort_dir_res") t->sort_dir_res = (int)v;
        else if KEY("rxt_min_rays") t->rxt_min_rays = u;
        else if KEY("rxt_max_tri") t->rxt_max_tri = (uint32_t)u;
        else if KEY("no_rxt") t->no_rxt = (int)v;
        else if KEY("no_txt") t->no_txt = (int)v;
        else if KEY("no_reorder") t->no_reorder = (int)v;
        else if KEY("no_patch") t->no_patch = (int)v;
        else if KEY("patch_size") t->patch_size = v;
        else if KEY("accel_sparse") t->accel_sparse = v;
        else if KEY("accel_big") t->accel_big = u;
        else if KEY("accel_fine") t->accel_fine = (int)v;
        else if KEY("accel_fine_min") t->accel_fine_min = u;
        else if KEY("no_bounce_prefetch") t->no_bounce_prefetch = (int)v;
        else if KEY("no_scatter") t->no_scatter = (int)v;
        else if KEY("no_chain") t->no_chain = (int)v;
        else if KEY("chain_from") t->chain_from = (int)v;
        else rc = hrt_fail(HRT_E_INVALID, "HRT_TUNE: unknown key '%s'", tok);
#undef KEY
    }
    free(buf);
    return rc;
}
