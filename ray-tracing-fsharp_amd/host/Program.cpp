// Program.cpp -- the console driver, after RayTracing.App/Program.fs:58-86: `rtfs_render <sample-name> [output.ppm]`.
// Differences that follow from the scope: the output is the gamma-corrected P3 PPM of ImageOutput.writePpm
// (the reference writes the same pixels as PNG through SkiaSharp), and the randomness comes from --seed.
// Options: --seed N  --spp N  --depth N  --scale K (divide maxWidthCoord/maxHeightCoord by K)  --device D  --no-gamma  --list
#include "SampleImages.hpp"

#include <cstdio>
#include <cstdlib>
#include <iostream>

using namespace RayTracing;

int main(int argc, char **argv) {
    try {
        std::vector<std::string> pos;
        uint64_t seed = 2024;
        int spp = 0, depth = -1, scale = 1, device = 0;
        bool gamma = true, info = false;
        for (int i = 1; i < argc; ++i) {
            const std::string a = argv[i];
            auto val = [&]() -> std::string { if (i + 1 >= argc) throw std::runtime_error("missing value for " + a); return argv[++i]; };
            if (a == "--list") { for (auto &kv : SampleImages::catalogue()) std::cout << kv.first << "\n"; std::cout << "gradient\n"; return 0; }
            else if (a == "--seed") seed = std::stoull(val());
            else if (a == "--spp") spp = std::stoi(val());
            else if (a == "--depth") depth = std::stoi(val());
            else if (a == "--scale") scale = std::stoi(val());
            else if (a == "--device") device = std::stoi(val());
            else if (a == "--no-gamma") gamma = false;
            else if (a == "--info") info = true; // print the flattened scene's shape and a hash of its tree, do not render
            else pos.push_back(a);
        }
        if (pos.empty() || pos.size() > 2) // failwithf "Expected two args 'sample name' 'output file', got %+A" (Program.fs:69)
            throw std::runtime_error("Expected two args 'sample name' 'output file', got " + std::to_string(pos.size()));
        const std::string name = pos[0], output = pos.size() == 2 ? pos[1] : std::string("/tmp/") + name + ".ppm";
        auto tick = [](double) {};
        if (name == "gradient") { // SampleImages.fs:37-57
            ImageOutput::writePpm(gamma, tick, SampleImages::gradient(), 256, 256, output);
            std::cout << output << "\n";
            return 0;
        }
        SampleImages::SceneDef def = name == "random-spheres" ? SampleImages::randomSpheres(seed) : SampleImages::get(name);
        if (spp > 0) def.camera.SamplesPerPixel = spp;
        if (depth >= 0) def.camera.BounceDepth = depth;
        if (scale > 1) { def.maxWidthCoord = std::max(1, def.maxWidthCoord / scale); def.maxHeightCoord = std::max(1, def.maxHeightCoord / scale); }
        auto scene = Scene::make(def.objects);
        if (info) {
            rt_scene_info si;
            check(rt_scene_get_info(scene->handle(), &si));
            std::vector<int32_t> skip((size_t) si.n_nodes), prim((size_t) si.n_nodes);
            std::vector<double> boxes((size_t) si.n_nodes * 6);
            check(rt_scene_get_tree(scene->handle(), skip.data(), prim.data(), boxes.data()));
            uint64_t h = 1469598103934665603ull; // FNV-1a over skip, prim, boxes
            auto mix = [&](const void *p, size_t n) { const unsigned char *b = (const unsigned char *) p; for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; } };
            mix(skip.data(), skip.size() * 4); mix(prim.data(), prim.size() * 4); mix(boxes.data(), boxes.size() * 8);
            std::cout << "bounded=" << si.n_bounded << " unbounded=" << si.n_unbounded << " nodes=" << si.n_nodes << " depth=" << si.tree_depth
                      << " maxW=" << def.maxWidthCoord << " maxH=" << def.maxHeightCoord << " spp=" << def.camera.SamplesPerPixel
                      << " bounce=" << def.camera.BounceDepth << " tree=" << h << " walk=" << (si.walk_tree == RT_WALK_TREE_SAH ? "sah" : "reference")
                      << " walkDepth=" << si.walk_tree_depth << "\n";
            return 0;
        }
        // one frame per scene here: rt_scene_tune (~20 ms for a probe and a tree build; same pixels, ~12 % off the frame time) repays
        // itself on frames of more than about 0.15 s; from ~3e9 pixel-samples on it is a clear gain
        const double samples = (double) (2 * def.maxWidthCoord + 1) * (double) (2 * def.maxHeightCoord + 1) * (double) def.camera.SamplesPerPixel;
        if (samples >= 3e9) {
            const rt_tune_info ti = scene->tune(def.maxWidthCoord, def.maxHeightCoord, def.camera, seed, device);
            if (ti.tuned) std::fprintf(stderr, "walk tree tuned: %.1f -> %.1f box tests per probe ray, %d -> %d nodes, %.0f ms\n", ti.box_tests_before,
                                       ti.box_tests_after, ti.nodes_before, ti.nodes_after, ti.probe_ms + ti.build_ms);
        }
        auto res = Scene::render(tick, [](const std::string &) {}, def.maxWidthCoord, def.maxHeightCoord, def.camera, scene, seed, device);
        Image &image = res.second;
        const std::vector<uint8_t> &rows = image.render(); // forces the render on the GPU
        ImageOutput::writePpm(gamma, tick, rows, image.RowCount, image.ColCount, output);
        std::fprintf(stderr, "%dx%d px, %llu samples, kernel %.2f ms\n", image.ColCount, image.RowCount, (unsigned long long) scene->lastStats.samples,
                     scene->lastStats.kernel_ms);
        std::cout << output << "\n"; // printfn "%s" pngOutput.FullName (Program.fs:56)
        return 0;
    } catch (const std::exception &e) {
        std::cerr << "error: " << e.what() << "\n";
        return 1;
    }
}
