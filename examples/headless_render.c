/*
 * examples/headless_render.c — the reference's main() without the window: load a scene file, upload it, render one frame
 * through the C ABI and write it as a binary PPM (what saveToPPM, utilities.h:842-856, was meant to dump: the 8-bit
 * framebuffer the viewer shows).  Plain C, no HIP headers: only the headers under include/firefly and libfirefly_hip.so.
 *
 *   gcc -O2 -Iinclude examples/headless_render.c -Lgpupathtracer_amd -lfirefly_hip -Wl,-rpath,$PWD/gpupathtracer_amd -o headless_render
 *   ./headless_render tests/data/box.scene out.ppm 320 240 4 16
 */
#include <stdio.h>
#include <stdlib.h>

#include "firefly/ff_api.h"

static int fail(const char* what)
{
    fprintf(stderr, "%s: %s\n", what, ff_last_error());
    return 1;
}

int main(int argc, char** argv)
{
    if (argc < 3) {
        fprintf(stderr, "usage: %s scene_file out.ppm [width height bounces spp]\n", argv[0]);
        return 2;
    }
    const int width = argc > 3 ? atoi(argv[3]) : 800, height = argc > 4 ? atoi(argv[4]) : 800; /* kernel.cu:262-263 */
    const int bounces = argc > 5 ? atoi(argv[5]) : 1, spp = argc > 6 ? atoi(argv[6]) : 1;

    FfSceneFile* scene = NULL;
    if (ff_scene_file_load(argv[1], &scene) != FF_OK) return fail("scene file");
    int n = 0;
    const FfGeometry* geoms = ff_scene_file_geometries(scene, &n);
    FfCamera camera;
    if (ff_scene_file_camera(scene, width, height, &camera) != FF_OK) return fail("camera");

    FfState* ff = NULL;
    if (ff_create(&ff, 0) != FF_OK) return fail("ff_create");
    if (ff_upload_scene(ff, geoms, n) != FF_OK) return fail("ff_upload_scene");          /* kernel.cu:268-298 */

    FfRenderParams rp;
    rp.width = width;
    rp.height = height;
    rp.bounces = bounces;
    rp.spp = spp;
    rp.seed = 1234;                                                                      /* utilities.h:118 */
    rp.trace_mode = FF_TRACE_BVH;
    rp.shade_mode = (bounces == 1 && spp == 1) ? FF_SHADE_NORMAL_DEBUG : FF_SHADE_DIFFUSE_PATH;
    rp.grid_mode = FF_GRID_FULL;
    rp.spp_per_launch = 0;

    unsigned char* rgb8 = (unsigned char*)malloc((size_t)width * height * 3);
    if (!rgb8) return 1;
    if (ff_render(ff, &camera, &rp, rgb8, 0, NULL, 0) != FF_OK) return fail("ff_render"); /* kernel.cu:335-344, headless */

    FfStats st;
    ff_stats(ff, &st);
    fprintf(stderr, "%d x %d, %d bounces, %d spp: %llu rays, kernel %.3f ms, call %.3f ms\n", width, height, bounces, spp,
            (unsigned long long)st.rays_traced, st.kernel_ms, st.total_ms);

    FILE* f = fopen(argv[2], "wb");
    if (!f) { perror(argv[2]); return 1; }
    fprintf(f, "P6\n%d %d\n255\n", width, height);
    fwrite(rgb8, 3, (size_t)width * height, f);
    fclose(f);

    free(rgb8);
    ff_destroy(ff);
    ff_scene_file_free(scene);
    return 0;
}
