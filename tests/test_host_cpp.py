"""The C++ host mirror (ray-tracing-fsharp_amd/host/RayTracing.hpp, SampleImages.hpp) and its console driver
(Program.cpp, after RayTracing.App/Program.fs): builds with g++, lists the catalogue, fails like the reference on a bad
name, flattens every scene to the same tree as the Python mirror, and -- on the GPU -- renders byte-identical PPMs."""
import dataclasses
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "ray-tracing-fsharp_amd", "host")
EXE = os.path.join(HOST, "rtfs_render")
NAMES = ["spheres", "shiny-floor", "fuzzy-floor", "inside-sphere", "total-refraction", "moved-camera", "glass", "random-spheres", "textured-sphere"]


@pytest.fixture(scope="module")
def exe():
    subprocess.check_call(["make", "-C", HOST, "-s"])
    return EXE


def run(exe, *args):
    return subprocess.run([exe, *args], capture_output=True, text=True, timeout=600)


def fnv1a(*arrays):
    h = 1469598103934665603
    for a in arrays:
        for b in np.ascontiguousarray(a).tobytes():
            h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def test_catalogue_and_reference_error_behaviour(exe):
    out = run(exe, "--list")
    assert out.returncode == 0 and sorted(out.stdout.split()) == sorted(NAMES + ["gradient"])
    bad = run(exe, "no-such-scene")
    assert bad.returncode == 1 and "Unrecognised arg: no-such-scene" in bad.stderr  # SampleImages.fs:32
    none = run(exe)
    assert none.returncode == 1 and "Expected two args" in none.stderr  # Program.fs:69


@pytest.mark.parametrize("name", NAMES)
def test_cpp_scene_flattens_like_the_python_mirror(rt, exe, name):
    out = run(exe, name, "--info", "--seed", "2024")
    assert out.returncode == 0, out.stderr
    fields = dict(kv.split("=") for kv in out.stdout.split())
    objs, cam, w, h = rt.sample_images.get(name)()
    s = rt.Scene.make(objs)
    info = s.info()
    skip, prim, boxes = s.tree()
    assert (int(fields["bounded"]), int(fields["unbounded"]), int(fields["nodes"]), int(fields["depth"])) == (
        info["n_bounded"], info["n_unbounded"], info["n_nodes"], info["tree_depth"])
    assert (int(fields["maxW"]), int(fields["maxH"]), int(fields["spp"]), int(fields["bounce"])) == (w, h, cam.SamplesPerPixel, cam.BounceDepth)
    assert int(fields["tree"]) == fnv1a(skip, prim, boxes)


def test_gradient_ppm_matches_the_python_writer(rt, exe, tmp_path):
    out = tmp_path / "g.ppm"
    assert run(exe, "gradient", str(out), "--no-gamma").returncode == 0
    assert out.read_bytes() == rt.ImageOutput.formatPpm(False, rt.sample_images.gradient())


def test_no_gpu_is_a_loud_error(rt, exe, tmp_path):
    if rt.device_count() > 0:
        pytest.skip("a GPU is visible")
    r = run(exe, "glass", str(tmp_path / "x.ppm"), "--scale", "20")
    assert r.returncode == 1 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_cpp_driver_renders_the_same_bytes_as_the_python_mirror(rt, exe, tmp_path, name):
    out = tmp_path / "o.ppm"
    r = run(exe, name, str(out), "--scale", "20", "--spp", "20", "--seed", "8")
    assert r.returncode == 0, r.stderr
    objs, cam, w, h = rt.sample_images.get(name)() if name != "random-spheres" else rt.sample_images.randomSpheres(seed=8)
    cam = dataclasses.replace(cam, SamplesPerPixel=20)
    total, image = rt.Scene.render(lambda _: None, lambda _: None, max(1, w // 20), max(1, h // 20), cam, rt.Scene.make(objs), seed=8)
    assert out.read_bytes() == rt.ImageOutput.formatPpm(True, rt.Image.render(image))
