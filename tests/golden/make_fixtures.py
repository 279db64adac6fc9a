"""Regenerates the committed fixtures under tests/golden/ (run in the build container, where /root/reference exists).

  PpmOutputExample.txt   byte copy of the golden file of the reference's only golden test
                         (RayTracing.Test/PpmOutputExample.txt, used by TestPpmOutput.fs:12-46)
  earthmap_rgb.npz       the reference's texture image RayTracing.App/earthmap.jpg decoded to RGB8 with PIL
                         (the reference decodes with SkiaSharp; decoders may differ by 1-2 LSB, so the DECODED texels are
                         the fixture and nothing decodes a JPEG at test time).  rows are top-first, as SKBitmap.GetPixel.
  oracle_*.npz           renders of small seeded scenes by the CPU oracle (oracle/oracle.cpp): PixelStats accumulators,
                         mean RGB and counters.  The oracle cannot be checked against a run of the F# reference (no .NET
                         here, and the reference is unseeded), so these pin the oracle against regressions and give the
                         GPU tests a second, committed target.
"""
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference"


def reference_data():
    shutil.copyfile(os.path.join(REF, "RayTracing.Test", "PpmOutputExample.txt"), os.path.join(HERE, "PpmOutputExample.txt"))
    from PIL import Image

    img = np.asarray(Image.open(os.path.join(REF, "RayTracing.App", "earthmap.jpg")).convert("RGB"), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "earthmap_rgb.npz"), rgb=img)


def oracle_renders():
    import oracle as orc
    import ray_tracing_fsharp_amd as rt
    import scenes

    earth = np.load(os.path.join(HERE, "earthmap_rgb.npz"))["rgb"]
    cases = {
        "oracle_all_materials_seed0": (scenes.all_materials(), 0),
        "oracle_final_thumb_seed7": (scenes.small_final(), 7),
        "oracle_config2_small_seed2": (rt.sample_images.config2_three_lambert(spp=30, pixels=27), 2),
        "oracle_earth_thumb_seed3": (scenes.earth_thumb(earth), 3),
    }
    for name, ((objs, cam, w, h), seed) in cases.items():
        acc, rgb, st = orc.OracleScene(objs).render_rows(w, h, cam.to_abi(), seed=seed, threads=8)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), accum=acc, rgb=rgb, seed=seed, max_w=w, max_h=h,
                            stats=np.array([st[k] for k in ("rays", "aabb_tests", "prim_tests", "reflections", "samples", "pixels_early")], np.uint64))
        print(name, acc.shape, st)


if __name__ == "__main__":
    if os.path.isdir(REF):
        reference_data()
    oracle_renders()
