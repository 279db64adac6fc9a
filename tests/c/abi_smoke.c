/* Plain-C consumer of include/rtfs_amd.h: proves the header is C (not only C++), that the library links, and that the
 * host-side entry points behave without any Python in between.  With a GPU it also renders a tiny frame.
 * Build: gcc -std=c99 -Wall -Werror -I include tests/c/abi_smoke.c -L ray-tracing-fsharp_amd -lrtfs_amd -lm */
#include "rtfs_amd.h"

#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(cond)                                                                                   \
    do {                                                                                              \
        if (!(cond)) { fprintf(stderr, "FAILED %s (line %d): %s\n", #cond, __LINE__, rt_last_error()); return 1; } \
    } while (0)

/* The layout a [<StructLayout(LayoutKind.Sequential)>] F# struct gets: fields in order, each at the next multiple
 * of its element's natural alignment (.NET's default Pack = 8 never bites: no field here is wider than 8), size rounded up to the
 * widest alignment.  fields = {element size, element count} per field, in declaration order.  Every offset must equal the one the
 * library was compiled with (rt_abi_offsetof) -- the check the F# shim makes with Marshal.OffsetOf at start-up. */
static int sequential_layout_matches(int which, const int (*fields)[2], int n_fields) {
    size_t off = 0, widest = 1;
    for (int i = 0; i < n_fields; ++i) {
        const size_t al = (size_t) fields[i][0];
        off = (off + al - 1) / al * al;
        if (rt_abi_offsetof(which, i) != off) { fprintf(stderr, "struct %d field %d: sequential layout %zu, library %zu\n", which, i, off, rt_abi_offsetof(which, i)); return 0; }
        off += al * (size_t) fields[i][1];
        if (al > widest) widest = al;
    }
    off = (off + widest - 1) / widest * widest;
    return rt_abi_offsetof(which, n_fields) == (size_t) -1 && rt_abi_sizeof(which) == off;
}

int main(void) {
    CHECK(rt_abi_version() == RT_ABI_VERSION);
    CHECK(rt_abi_sizeof(0) == sizeof(rt_hittable) && rt_abi_sizeof(1) == sizeof(rt_texture) && rt_abi_sizeof(2) == sizeof(rt_camera));
    CHECK(rt_abi_sizeof(3) == sizeof(rt_scene_info) && rt_abi_sizeof(4) == sizeof(rt_stats));
    CHECK(rt_abi_sizeof(5) == sizeof(rt_render_options) && rt_abi_sizeof(6) == sizeof(rt_scene_options));
    {   /* the field lists of INTEGRATION.md's F# mirrors: RtHittable (104 bytes, texture at 100), RtTexture, RtCamera, ... */
        static const int hittable[][2] = {{4, 1}, {4, 1}, {8, 3}, {8, 3}, {8, 1}, {8, 1}, {8, 1}, {8, 1}, {8, 1}, {1, 3}, {1, 1}, {4, 1}};
        static const int texture[][2] = {{4, 1}, {1, 3}, {1, 3}, {1, 2}, {4, 1}, {4, 1}, {8, 1}, {4, 1}, {4, 1}, {8, 1}, {8, 3}, {8, 1}};
        static const int camera[][2] = {{8, 3}, {8, 3}, {8, 3}, {8, 3}, {8, 3}, {8, 3}, {8, 1}, {8, 1}, {8, 1}, {4, 1}, {4, 1}};
        static const int info[][2] = {{4, 1}, {4, 1}, {4, 1}, {4, 1}, {4, 1}, {4, 1}, {4, 1}, {4, 1}, {8, 1}, {8, 1}, {4, 1}, {4, 1}};
        static const int stats[][2] = {{8, 1}, {8, 1}, {8, 1}, {8, 1}, {8, 1}, {8, 1}, {8, 1}, {8, 1}, {8, 1}};
        static const int ropt[][2] = {{4, 1}, {4, 1}, {4, 1}, {4, 1}, {4, 1}, {4, 1}, {4, 1}, {4, 1}};
        static const int sopt[][2] = {{4, 1}, {4, 1}};
        static const int tune[][2] = {{4, 1}, {4, 1}, {4, 1}, {4, 1}, {4, 1}, {4, 1}, {8, 1}, {8, 1}, {8, 1}, {8, 1}};
        CHECK(sequential_layout_matches(0, hittable, 12) && sizeof(rt_hittable) == 104 && rt_abi_offsetof(0, 11) == 100);
        CHECK(sequential_layout_matches(1, texture, 12));
        CHECK(sequential_layout_matches(2, camera, 11));
        CHECK(sequential_layout_matches(3, info, 12) && sequential_layout_matches(4, stats, 9));
        CHECK(sequential_layout_matches(5, ropt, 8) && sequential_layout_matches(6, sopt, 2));
        CHECK(rt_abi_offsetof(0, 3) == offsetof(rt_hittable, normal) && rt_abi_offsetof(1, 9) == offsetof(rt_texture, texels));
        CHECK(rt_abi_sizeof(7) == sizeof(rt_tune_info) && sequential_layout_matches(7, tune, 10));
        CHECK(rt_abi_offsetof(8, 0) == (size_t) -1);
    }

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
    {   /* rt_scene_tune_rays: the host half of tuning (no GPU involved); one bounded sphere walks the reference's tree: left alone */
        const double probe[12] = {0.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.5, 0.5, -2.0, 0.0, 0.0, 1.0};
        rt_tune_info ti;
        memset(&ti, 0, sizeof(ti));
        ti.struct_size = (uint32_t) sizeof(ti);
        CHECK(rt_scene_tune_rays(scene, probe, 2, &ti) == RT_OK && ti.tuned == 0 && ti.struct_size == sizeof(ti) && ti.nodes_after == 1);
        ti.struct_size = 0;
        CHECK(rt_scene_tune_rays(scene, probe, 2, &ti) == RT_ERR_INVALID_ARGUMENT);
        /* five spheres in a row: tuned, and the tree stays a tree over the same five leaves */
        rt_hittable row[5];
        for (int i = 0; i < 5; ++i) { row[i] = h[0]; row[i].point[0] = 3.0 * i; }
        rt_scene *five = NULL;
        CHECK(rt_scene_create(row, 5, NULL, 0, &five) == RT_OK);
        ti.struct_size = (uint32_t) sizeof(ti);
        CHECK(rt_scene_tune_rays(five, probe, 2, &ti) == RT_OK && ti.tuned == 1 && ti.probe_rays == 2 && ti.nodes_before == 9);
        CHECK(rt_scene_get_info(five, &info) == RT_OK && info.walk_tree == RT_WALK_TREE_TUNED && info.walk_tree_nodes == ti.nodes_after && info.n_nodes == 9);
        int32_t prim[9], leaves = 0;
        CHECK(rt_scene_get_walk_tree(five, NULL, prim, NULL) == RT_OK);
        for (int i = 0; i < info.walk_tree_nodes; ++i) leaves += prim[i] >= 0;
        CHECK(leaves == 5);
        rt_scene_destroy(five);
        CHECK(rt_scene_get_info(scene, &info) == RT_OK);
    }
    h[0].style = 99;
    rt_scene *bad = NULL;
    CHECK(rt_scene_create(h, 3, NULL, 0, &bad) == RT_ERR_INVALID_ARGUMENT && strstr(rt_last_error(), "bad style") != NULL);

    /* ImageOutput.writePpm's bytes (the reference's golden file) and PixelOutput.correct */
    const uint8_t px[18] = {255, 0, 0, 0, 255, 0, 0, 0, 255, 255, 255, 0, 255, 255, 255, 0, 0, 0};
    char buf[128];
    CHECK(rt_format_ppm(px, 2, 3, 0, buf, sizeof(buf)) == 62);
    CHECK(strcmp(buf, "P3\n3 2\n255\n255 0 0 0 255 0 0 0 255\n255 255 0 255 255 255 0 0 0") == 0);
    CHECK(rt_gamma_correct(64) == 128 && rt_gamma_correct(255) == 255 && rt_gamma_correct(1) == 16);
    /* TestPpmOutput.fs:12-46 end to end: the 3x2 image spilled in ImageOutput.resume's temp-file format (ImageOutput.fs:131-161:
     * `<row>,<col>\n` in ASCII with 0 written as NO digits, then three raw bytes), read back (readPixelMap, ImageOutput.fs:46-113),
     * and written as P3: the reference's golden text again */
    {
        uint8_t map[128], back[18], present[6];
        const int64_t len = rt_format_pixel_map(px, 2, 3, map, sizeof(map));
        /* rows 0,1 x cols 0,1,2: "," "\n" + 3 bytes = 5 for (0,0); one digit more per non-zero coordinate */
        CHECK(len == 5 + 6 + 6 + 6 + 7 + 7);
        CHECK(memcmp(map, ",\n\xff\x00\x00,1\n\x00\xff\x00", 11) == 0);
        memset(back, 7, sizeof(back));
        CHECK(rt_parse_pixel_map(map, (size_t) len, 2, 3, back, present) == 6);
        for (int i = 0; i < 6; ++i) CHECK(present[i] == 1);
        CHECK(memcmp(back, px, sizeof(px)) == 0);
        CHECK(rt_format_ppm(back, 2, 3, 0, buf, sizeof(buf)) == 62);
        CHECK(strcmp(buf, "P3\n3 2\n255\n255 0 0 0 255 0 0 0 255\n255 255 0 255 255 255 0 0 0") == 0);
        /* a file cut off in the middle of a pixel: the pixels before the cut are kept, nothing else is touched (ImageOutput.fs:69-106) */
        memset(back, 7, sizeof(back));
        CHECK(rt_parse_pixel_map(map, (size_t) len - 2, 2, 3, back, present) == 5 && present[5] == 0 && back[15] == 7);
        /* a pixel outside the image is an error (the reference throws) */
        CHECK(rt_parse_pixel_map(map, (size_t) len, 1, 3, back, present) == -RT_ERR_INVALID_ARGUMENT);
    }

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
        /* Scene.render on "several" devices from this one process (the list repeats the one GPU of the box), with per-call
         * launch options: the very same PixelStats */
        const int32_t devices[3] = {0, 0, 0};
        rt_render_options opt;
        memset(&opt, 0, sizeof(opt));
        opt.struct_size = (uint32_t) sizeof(opt); opt.block_threads = 256; opt.passes = 2; opt.park_lanes = 8;
        int32_t accum2[7 * 7 * 4];
        uint8_t rgb2[7 * 7 * 3];
        rt_stats st3[3];
        CHECK(rt_render_frame(scene, &cam, 3, 3, 42, devices, 3, RT_RENDER_COUNTERS, RT_GATHER_PEER, &opt, accum2, rgb2, st3) == RT_OK);
        CHECK(memcmp(accum, accum2, sizeof(accum)) == 0 && memcmp(rgb, rgb2, sizeof(rgb)) == 0);
        CHECK(st3[0].rays + st3[1].rays + st3[2].rays == st.rays && st3[0].pixels == 21 && st3[2].pixels == 14);
        /* rt_scene_tune from C: a probe render on the GPU; this scene (one bounded sphere) has nothing to tune */
        rt_tune_info ti;
        memset(&ti, 0, sizeof(ti));
        ti.struct_size = (uint32_t) sizeof(ti);
        CHECK(rt_scene_tune(scene, &cam, 3, 3, 42, 0, &ti) == RT_OK && ti.tuned == 0);
        CHECK(rt_scene_tune(scene, &cam, 0, 3, 42, 0, &ti) == RT_ERR_INVALID_ARGUMENT);
        printf("abi_smoke: host checks ok, rendered 7x7 px: %llu samples, %llu rays\n", (unsigned long long) st.samples, (unsigned long long) st.rays);
    }
    rt_scene_destroy(scene);
    return 0;
}
