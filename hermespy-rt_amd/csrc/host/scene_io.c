/* scene_io.c -- .hrt scene files (drop-in for src/scene.c:7-83 of the reference).
 *
 * File format, little-endian, unaligned, no version field:
 *     "HRT" | u32 num_meshes (1..1000)
 *     per mesh: u32 num_vertices | num_vertices * 3 f32
 *               u32 num_triangles | num_triangles * 3 u32 (vertex indices)
 *               u32 material_index | 3 f32 velocity
 * Normals are not stored (Mesh.ns == NULL after loading).
 *
 * Error behaviour is the reference's: a message on stderr and exit(8).
 */
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hrt_internal.h"

static void die8(const char *msg)
{
    if (errno) perror(msg);
    else fprintf(stderr, "%s\n", msg);
    exit(8);
}

static void put(FILE *fp, const void *p, size_t n)
{
    if (n && fwrite(p, 1, n, fp) != n) die8("Error: cannot write scene file");
}

void scene_save(Scene *scene, const char *filepath)
{
    FILE *fp = fopen(filepath, "wb");
    if (!fp) die8("Error: cannot open file");
    put(fp, "HRT", 3);
    put(fp, &scene->num_meshes, 4);
    for (uint32_t i = 0; i < scene->num_meshes; ++i) {
        const Mesh *m = &scene->meshes[i];
        put(fp, &m->num_vertices, 4);
        put(fp, m->vs, (size_t)m->num_vertices * sizeof(Vec3));
        put(fp, &m->num_triangles, 4);
        put(fp, m->is, (size_t)m->num_triangles * 3 * sizeof(uint32_t));
        put(fp, &m->material_index, 4);
        put(fp, &m->velocity, sizeof(Vec3));
    }
    if (fclose(fp) != 0) die8("Error: cannot write scene file");
}

static void get(FILE *fp, void *p, size_t n)
{
    if (n && fread(p, 1, n, fp) != n) {
        errno = 0;
        die8("Could not read scene file: truncated");
    }
}

Scene scene_load(const char *filepath)
{
    FILE *fp = fopen(filepath, "rb");
    if (!fp) die8("Could not open scene file");
    errno = 0;
    char magic[3];
    get(fp, magic, 3);
    if (memcmp(magic, "HRT", 3) != 0) die8("Not an HRT scene file");
    Scene sc;
    get(fp, &sc.num_meshes, 4);
    if (sc.num_meshes == 0) die8("Scene has no meshes");
    if (sc.num_meshes > 1000) die8("Scene has too many meshes");
    sc.meshes = (Mesh *)calloc(sc.num_meshes, sizeof(Mesh));
    if (!sc.meshes) die8("Out of memory loading scene");
    for (uint32_t i = 0; i < sc.num_meshes; ++i) {
        Mesh *m = &sc.meshes[i];
        get(fp, &m->num_vertices, 4);
        m->vs = (Vec3 *)malloc((size_t)(m->num_vertices ? m->num_vertices : 1) * sizeof(Vec3));
        if (!m->vs) die8("Out of memory loading scene");
        get(fp, m->vs, (size_t)m->num_vertices * sizeof(Vec3));
        get(fp, &m->num_triangles, 4);
        m->is = (uint32_t *)malloc((size_t)(m->num_triangles ? m->num_triangles : 1) * 3 *
                                   sizeof(uint32_t));
        if (!m->is) die8("Out of memory loading scene");
        get(fp, m->is, (size_t)m->num_triangles * 3 * sizeof(uint32_t));
        get(fp, &m->material_index, 4);
        get(fp, &m->velocity, sizeof(Vec3));
        m->ns = NULL;
    }
    fclose(fp);
    return sc;
}
