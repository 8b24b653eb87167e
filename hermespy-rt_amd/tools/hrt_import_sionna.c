/* hrt_import_sionna -- command-line importer: Sionna / Mitsuba scene -> .hrt
 *
 *     hrt_import_sionna <scene.xml> [out.hrt]
 *
 * Same job and conventions as the reference's `scene_fromSionna` tool
 * (src/scene_fromSionna.c:461-488): writes scene.hrt in the current directory unless an output
 * path is given; the names box.xml and simple_reflector.xml select the two built-in scenes;
 * any failure ends with status 8.  Parsing lives in the library (hrt_scene_import_sionna).
 */
#include <stdio.h>

#include "hrt_device.h"

int main(int argc, char **argv)
{
    if (argc < 2 || argc > 3) {
        fprintf(stderr, "Usage: %s <scene.xml> [out.hrt]\n", argv[0]);
        return 1;
    }
    Scene scene;
    int rc = hrt_scene_import_sionna(argv[1], &scene);
    if (rc != HRT_OK) {
        fprintf(stderr, "hrt_import_sionna: %s\n", hrt_last_error());
        return 8;
    }
    scene_save(&scene, argc == 3 ? argv[2] : "scene.hrt");
    free_scene(&scene);
    return 0;
}
