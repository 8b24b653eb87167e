/* sionna_import.c -- Sionna / Mitsuba scene (XML + binary PLY + CSV) -> Scene.
 *
 * Host-side data-format row next to the hot path: what the reference's
 * importer CLI does (src/scene_fromSionna.c:103-488), as a library call that returns a status
 * instead of exiting, plus the CLI in ../../tools/hrt_import_sionna.c.
 *
 * Input conventions (those of the reference):
 *   scene.xml   every "<shape" element contributes, in file order, a mesh: its name is the first
 *               name="..." after "<shape", its PLY file the value="..." of the following
 *               <string name="filename" .../> (relative to the XML's directory), its material
 *               the text after the next  id="mat-itu_  up to the closing quote (:255-367).
 *               Material names map to ITU-R P.2040 table rows: air concrete brick plasterboard
 *               wood glass1 glass2 ceiling_board1 ceiling_board2 chipboard plywood marble
 *               floorboard metal very_dry_ground medium_dry_ground wet_ground; anything else
 *               is air (src/materials.c:96-122).
 *   *.ply       binary_little_endian 1.0; "element vertex N" of 5 floats (x y z s t, s/t
 *               skipped), "element face M" of uchar 3 + 3 x int32 (:103-164).
 *   scene.csv   optional per-mesh overrides "name,material_index,velocity_x,velocity_y,
 *               velocity_z" after that exact header line (:181-243).
 * Two scene names are built in and need no files: box.xml and simple_reflector.xml (:15-82);
 * they reproduce scenes/box.hrt and scenes/simple_reflector.hrt byte for byte.
 *
 * Deliberate differences from the reference tool (its defects): a missing CSV
 * is "no overrides" (the reference exits), a complete 5-field CSV line is applied (the
 * reference rejects it because of an off-by-one in its sscanf check and only accepts lines
 * WITHOUT velocity_z, whose z it then leaves uninitialised -- here such a line gets z = 0),
 * a material_index > 16 is an error.  For every input the reference tool accepts and fully
 * defines (header-only CSV), the produced .hrt is byte-identical (tests/test_sionna_import.py).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hrt_internal.h"

static const char *const k_material_names[HRT_NUM_MATERIALS] = {
    "air", "concrete", "brick", "plasterboard", "wood", "glass1", "glass2", "ceiling_board1",
    "ceiling_board2", "chipboard", "plywood", "marble", "floorboard", "metal",
    "very_dry_ground", "medium_dry_ground", "wet_ground"};

static uint32_t material_from_name(const char *s, size_t n)
{
    for (uint32_t i = 0; i < HRT_NUM_MATERIALS; ++i)
        if (strlen(k_material_names[i]) == n && memcmp(k_material_names[i], s, n) == 0) return i;
    return 0;   /* unknown -> air */
}

static void scene_release(Scene *sc)
{
    if (!sc->meshes) return;
    for (uint32_t i = 0; i < sc->num_meshes; ++i) {
        free(sc->meshes[i].vs);
        free(sc->meshes[i].is);
    }
    free(sc->meshes);
    sc->meshes = NULL;
    sc->num_meshes = 0;
}

static int mesh_from_arrays(Mesh *m, const float *vs, uint32_t nv, const uint32_t *is, uint32_t nt,
                            uint32_t material)
{
    memset(m, 0, sizeof *m);
    m->vs = (Vec3 *)malloc((size_t)nv * sizeof(Vec3));
    m->is = (uint32_t *)malloc((size_t)nt * 3 * sizeof(uint32_t));
    if (!m->vs || !m->is) return hrt_fail(HRT_E_NOMEM, "out of host memory");
    memcpy(m->vs, vs, (size_t)nv * sizeof(Vec3));
    memcpy(m->is, is, (size_t)nt * 3 * sizeof(uint32_t));
    m->num_vertices = nv;
    m->num_triangles = nt;
    m->material_index = material;
    return HRT_OK;
}

/* the two built-in scenes (geometry as data: a 10 x 10 x 5 m concrete box seen from inside,
 * a 1 x 1 m concrete plate) */
static int builtin_scene(const char *name, Scene *out)
{
    static const float box_v[] = {5, 5, 0, -5, 5, 0, -5, -5, 0, 5, -5, 0,
                                  5, 5, 5, -5, 5, 5, -5, -5, 5, 5, -5, 5};
    static const uint32_t box_i[] = {0, 1, 2, 0, 2, 3, 0, 4, 5, 0, 5, 1, 1, 5, 6, 1, 6, 2,
                                     2, 6, 7, 2, 7, 3, 3, 7, 4, 3, 4, 0, 4, 7, 6, 4, 6, 5};
    static const float refl_v[] = {-.5f, -.5f, 0, .5f, -.5f, 0, .5f, .5f, 0, -.5f, .5f, 0};
    static const uint32_t refl_i[] = {0, 1, 2, 0, 2, 3};
    int which = !strcmp(name, "box.xml") ? 0 : !strcmp(name, "simple_reflector.xml") ? 1 : -1;
    if (which < 0) return 1;   /* not built in */
    out->meshes = (Mesh *)calloc(1, sizeof(Mesh));
    if (!out->meshes) return hrt_fail(HRT_E_NOMEM, "out of host memory");
    out->num_meshes = 1;
    int rc = which == 0 ? mesh_from_arrays(&out->meshes[0], box_v, 8, box_i, 12, 1)
                        : mesh_from_arrays(&out->meshes[0], refl_v, 4, refl_i, 2, 1);
    if (rc) scene_release(out);
    return rc;
}

static int read_ply(const char *path, Mesh *m)
{
    memset(m, 0, sizeof *m);
    FILE *f = fopen(path, "rb");
    if (!f) return hrt_fail(HRT_E_INVALID, "cannot open PLY file %s", path);
    char line[256];
    unsigned nv = 0, nt = 0;
    int ended = 0;
    while (fgets(line, sizeof line, f)) {
        if (!strncmp(line, "end_header", 10)) { ended = 1; break; }
        if (!strncmp(line, "element vertex ", 15)) sscanf(line + 15, "%u", &nv);
        else if (!strncmp(line, "element face ", 13)) sscanf(line + 13, "%u", &nt);
    }
    int rc = HRT_OK;
    if (!ended || nv == 0 || nt == 0) rc = hrt_fail(HRT_E_INVALID, "%s: PLY header without vertex/face elements", path);
    else if (nv > 1000000 || nt > 1000000) rc = hrt_fail(HRT_E_INVALID, "%s: PLY too big", path);
    if (!rc) {
        m->vs = (Vec3 *)malloc((size_t)nv * sizeof(Vec3));
        m->is = (uint32_t *)malloc((size_t)nt * 3 * sizeof(uint32_t));
        if (!m->vs || !m->is) rc = hrt_fail(HRT_E_NOMEM, "out of host memory");
    }
    for (unsigned i = 0; !rc && i < nv; ++i) {
        float v[5];   /* x y z s t */
        if (fread(v, 4, 5, f) != 5) rc = hrt_fail(HRT_E_INVALID, "%s: truncated vertex data", path);
        else memcpy(&m->vs[i], v, sizeof(Vec3));
    }
    for (unsigned i = 0; !rc && i < nt; ++i) {
        unsigned char n;
        if (fread(&n, 1, 1, f) != 1 || n != 3 || fread(&m->is[3 * i], 4, 3, f) != 3)
            rc = hrt_fail(HRT_E_INVALID, "%s: face %u is not a readable triangle", path, i);
    }
    fclose(f);
    if (rc) { free(m->vs); free(m->is); memset(m, 0, sizeof *m); return rc; }
    m->num_vertices = nv;
    m->num_triangles = nt;
    return HRT_OK;
}

typedef struct { char name[50]; uint32_t material; Vec3 vel; } csv_row;

static int read_csv(const char *path, csv_row **rows, uint32_t *n)
{
    *rows = NULL;
    *n = 0;
    FILE *f = fopen(path, "r");
    if (!f) return HRT_OK;   /* optional */
    char line[256];
    if (!fgets(line, sizeof line, f) ||
        strncmp(line, "name,material_index,velocity_x,velocity_y,velocity_z", 52)) {
        fclose(f);
        return hrt_fail(HRT_E_INVALID, "%s: invalid CSV header", path);
    }
    uint32_t cap = 0;
    int rc = HRT_OK;
    while (!rc && fgets(line, sizeof line, f)) {
        if (line[0] == '\n' || line[0] == '\r' || line[0] == '\0') continue;
        csv_row r;
        memset(&r, 0, sizeof r);
        int got = sscanf(line, "%49[^,],%u,%f,%f,%f", r.name, &r.material, &r.vel.x, &r.vel.y, &r.vel.z);
        if (got < 4) rc = hrt_fail(HRT_E_INVALID, "%s: cannot parse line: %s", path, line);
        else if (r.material >= HRT_NUM_MATERIALS) rc = hrt_fail(HRT_E_INVALID, "%s: material_index %u out of range", path, r.material);
        else {
            if (*n == cap) {
                cap = cap ? 2 * cap : 16;
                csv_row *nr = (csv_row *)realloc(*rows, cap * sizeof(csv_row));
                if (!nr) { rc = hrt_fail(HRT_E_NOMEM, "out of host memory"); break; }
                *rows = nr;
            }
            (*rows)[(*n)++] = r;
        }
    }
    fclose(f);
    if (rc) { free(*rows); *rows = NULL; *n = 0; }
    return rc;
}

/* text between the quote at `p` (pointing just after an opening quote) and the next quote */
static const char *quoted(const char *p, size_t *len)
{
    const char *e = strchr(p, '"');
    if (!e) return NULL;
    *len = (size_t)(e - p);
    return p;
}

int hrt_scene_import_sionna(const char *xml_path, Scene *out)
{
    if (!xml_path || !out) return hrt_fail(HRT_E_INVALID, "hrt_scene_import_sionna: NULL argument");
    memset(out, 0, sizeof *out);
    const char *slash = strrchr(xml_path, '/');
    const char *base = slash ? slash + 1 : xml_path;
    int rc = builtin_scene(base, out);
    if (rc <= 0) return rc;   /* built in (0) or error (<0) */

    const size_t plen = strlen(xml_path);
    if (plen < 5 || strcmp(xml_path + plen - 4, ".xml"))
        return hrt_fail(HRT_E_INVALID, "scene file must end with .xml: %s", xml_path);
    FILE *f = fopen(xml_path, "rb");
    if (!f) return hrt_fail(HRT_E_INVALID, "cannot open %s", xml_path);
    fseek(f, 0, SEEK_END);
    long fsz = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *xml = (char *)malloc((size_t)fsz + 1);
    if (!xml || fread(xml, 1, (size_t)fsz, f) != (size_t)fsz) {
        fclose(f);
        free(xml);
        return hrt_fail(HRT_E_INVALID, "cannot read %s", xml_path);
    }
    fclose(f);
    xml[fsz] = '\0';

    char *csv_path = (char *)malloc(plen + 1);
    csv_row *rows = NULL;
    uint32_t n_rows = 0;
    if (!csv_path) { free(xml); return hrt_fail(HRT_E_NOMEM, "out of host memory"); }
    memcpy(csv_path, xml_path, plen + 1);
    memcpy(csv_path + plen - 4, ".csv", 4);
    rc = read_csv(csv_path, &rows, &n_rows);
    free(csv_path);

    uint32_t n_shapes = 0;
    for (const char *p = xml; !rc && (p = strstr(p, "<shape")); p += 6) ++n_shapes;
    if (!rc && n_shapes == 0) rc = hrt_fail(HRT_E_INVALID, "%s: no <shape> elements", xml_path);
    if (!rc && n_shapes > 1000) rc = hrt_fail(HRT_E_INVALID, "%s: more than 1000 shapes (the .hrt limit)", xml_path);
    if (!rc) {
        out->meshes = (Mesh *)calloc(n_shapes, sizeof(Mesh));
        if (!out->meshes) rc = hrt_fail(HRT_E_NOMEM, "out of host memory");
    }
    const size_t dir_len = slash ? (size_t)(slash - xml_path) + 1 : 0;
    const char *p = xml;
    for (uint32_t i = 0; !rc && i < n_shapes; ++i) {
        p = strstr(p, "<shape") + 6;
        size_t name_len, file_len, mat_len;
        const char *q = strstr(p, "name=\"");
        const char *name = q ? quoted(q + 6, &name_len) : NULL;
        q = name ? strstr(name, "<string name=\"filename\"") : NULL;
        q = q ? strstr(q, "value=\"") : NULL;
        const char *file = q ? quoted(q + 7, &file_len) : NULL;
        q = file ? strstr(file, "id=\"mat-itu_") : NULL;
        const char *mat = q ? quoted(q + 12, &mat_len) : NULL;
        if (!name || !file || !mat) {
            rc = hrt_fail(HRT_E_INVALID, "%s: shape %u lacks a name, a filename or a mat-itu_ material", xml_path, i);
            break;
        }
        char *ply = (char *)malloc(dir_len + file_len + 1);
        if (!ply) { rc = hrt_fail(HRT_E_NOMEM, "out of host memory"); break; }
        memcpy(ply, xml_path, dir_len);
        memcpy(ply + dir_len, file, file_len);
        ply[dir_len + file_len] = '\0';
        rc = read_ply(ply, &out->meshes[i]);
        free(ply);
        if (rc) break;
        out->num_meshes = i + 1;
        out->meshes[i].material_index = material_from_name(mat, mat_len);
        for (uint32_t j = 0; j < n_rows; ++j)
            if (strlen(rows[j].name) == name_len && !memcmp(rows[j].name, name, name_len)) {
                out->meshes[i].material_index = rows[j].material;
                out->meshes[i].velocity = rows[j].vel;
                break;
            }
    }
    free(rows);
    free(xml);
    if (rc) scene_release(out);
    return rc;
}
