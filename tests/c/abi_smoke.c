/* Plain-C consumer of include/rtfs_amd.h: proves the header is C (not only C++), that the library links, and that the
 * host-side entry points behave without any Python in between.  With a GPU it also renders a tiny frame.
 * Build: gcc -std=c99 -Wall -Werror -I include tests/c/abi_smoke.c -L ray-tracing-fsharp_amd -lrtfs_amd -lm */
#include "rtfs_amd.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(cond)                                                                                   \
    do {                                                                                              \
        if (!(cond)) { fprintf(stderr, "FAILED %s (line %d): %s\n", #cond, __LINE__, rt_last_error()); return 1; } \
    } while (0)

int main(void) {
    CHECK(rt_abi_version() == RT_ABI_VERSION);
    CHECK(rt_abi_sizeof(0) == sizeof(rt_hittable) && rt_abi_sizeof(1) == sizeof(rt_texture) && rt_abi_sizeof(2) == sizeof(rt_camera));
    CHECK(rt_abi_sizeof(3) == sizeof(rt_scene_info) && rt_abi_sizeof(4) == sizeof(rt_stats));

    /* Camera.makeBasic for the final scene */
    const double origin[3] = {13.0, 2.0, -3.0}, up[3] = {0.0, 1.0, 0.0};
    const double n = 1.0 / sqrt(13.0 * 13.0 + 2.0 * 2.0 + 3.0 * 3.0);
    const double dir[3] = {n * -13.0, n * -2.0, n * 3.0};
    rt_camera cam;
    CHECK(rt_camera_make_basic(500, 10.0, 1.5, origin, dir, up, &cam) == RT_OK);
    CHECK(cam.bounce_depth == 150 && cam.viewport_width == 3.0 && cam.viewport_height == 2.0);
    CHECK(fabs(cam.xaxis_dir[0] - 0.22485950669875845) < 1e-15 && cam.xaxis_dir[1] == 0.0);

    /* Scene.make: a Lambert sphere, an unbounded light dome, a mirror plane */
    rt_hittable h[3];
    memset(h, 0, sizeof(h));
    h[0].kind = RT_HITTABLE_SPHERE; h[0].style = RT_SPHERE_LAMBERT_REFLECTION; h[0].point[2] = 3.0; h[0].radius = 1.0; h[0].albedo = 0.8;
    h[0].rgb[0] = 200; h[0].rgb[1] = 100; h[0].rgb[2] = 50; h[0].texture = -1;
    h[1].kind = RT_HITTABLE_UNBOUNDED_SPHERE; h[1].style = RT_SPHERE_LIGHT_SOURCE; h[1].radius = 100.0; h[1].rgb[0] = h[1].rgb[1] = h[1].rgb[2] = 255; h[1].texture = -1;
    h[2].kind = RT_HITTABLE_INFINITE_PLANE; h[2].style = RT_PLANE_PURE_REFLECTION; h[2].point[1] = -1.0; h[2].normal[1] = 1.0; h[2].albedo = 0.5;
    h[2].rgb[0] = h[2].rgb[1] = h[2].rgb[2] = 255; h[2].texture = -1;
    rt_scene *scene = NULL;
    CHECK(rt_scene_create(h, 3, NULL, 0, &scene) == RT_OK);
    rt_scene_info info;
    CHECK(rt_scene_get_info(scene, &info) == RT_OK);
    CHECK(info.n_bounded == 1 && info.n_unbounded == 2 && info.n_nodes == 1 && info.lds_resident == 1);
    h[0].style = 99;
    rt_scene *bad = NULL;
    CHECK(rt_scene_create(h, 3, NULL, 0, &bad) == RT_ERR_INVALID_ARGUMENT && strstr(rt_last_error(), "bad style") != NULL);

    /* ImageOutput.writePpm's bytes (the reference's golden file) and PixelOutput.correct */
    const uint8_t px[18] = {255, 0, 0, 0, 255, 0, 0, 0, 255, 255, 255, 0, 255, 255, 255, 0, 0, 0};
    char buf[128];
    CHECK(rt_format_ppm(px, 2, 3, 0, buf, sizeof(buf)) == 62);
    CHECK(strcmp(buf, "P3\n3 2\n255\n255 0 0 0 255 0 0 0 255\n255 255 0 255 255 255 0 0 0") == 0);
    CHECK(rt_gamma_correct(64) == 128 && rt_gamma_correct(255) == 255 && rt_gamma_correct(1) == 16);

    /* render: loud failure without a GPU, a real frame with one */
    const double o0[3] = {0.0, 0.0, 0.0}, z[3] = {0.0, 0.0, 1.0};
    CHECK(rt_camera_make_basic(20, 1.0, 1.0, o0, z, up, &cam) == RT_OK);
    cam.bounce_depth = 10;
    int32_t accum[7 * 7 * 4];
    uint8_t rgb[7 * 7 * 3];
    rt_stats st;
    const int rc = rt_render(scene, &cam, 3, 3, 42, 0, 0, 1, 7, RT_RENDER_COUNTERS, accum, rgb, &st);
    if (rt_device_count() == 0) {
        CHECK(rc == RT_ERR_NO_DEVICE && strstr(rt_last_error(), "no CPU fallback") != NULL);
        printf("abi_smoke: host checks ok (no GPU: render refused as it must)\n");
    } else {
        CHECK(rc == RT_OK);
        CHECK(st.pixels == 49 && st.samples >= 49 * 11 && st.rays >= st.samples);
        CHECK(accum[0] == 11 || accum[0] == 20);
        printf("abi_smoke: host checks ok, rendered 7x7 px: %llu samples, %llu rays\n", (unsigned long long) st.samples, (unsigned long long) st.rays);
    }
    rt_scene_destroy(scene);
    return 0;
}
