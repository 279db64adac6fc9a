"""rt_scene_tune on the GPU: the probe render, the rebuilt (n-ary, thinned) walk tree, and -- the point -- that no pixel and no
counter other than the box tests changes.  (Scenes are made with walk_tree="sah" explicitly: a scene that walks the reference's own tree,
the process default under the RTFS_TREE=reference knob, is left alone by the tune.)  The whole suite also runs with every scene tuned (RTFS_TUNE=1, conftest.py)."""
import dataclasses
import os

import numpy as np
import pytest

import scenes
from test_tune_tree import check_nary

pytestmark = pytest.mark.gpu
KNOB = os.environ.get("RTFS_TUNE") == "1"  # conftest.py: every scene is tuned at its first render, the "untuned" ones here too


def _same_but_for_box_tests(a, b):
    assert np.array_equal(a.accum, b.accum) and np.array_equal(a.rgb, b.rgb)
    for k in ("rays", "prim_tests", "reflections", "samples", "pixels", "pixels_early"):
        assert a.stats[k] == b.stats[k], k


def test_tuned_final_scene_thumbnail_equals_untuned_and_oracle(rt, orc):
    objs, cam, w, h = scenes.small_final(spp=60, pixels=40)
    plain = rt.Scene.make(objs, walk_tree="sah")
    base = plain.render_rows(w, h, cam, seed=5, counters=True)
    s = rt.Scene.make(objs, walk_tree="sah")
    first = s.render_rows(w, h, cam, seed=5, counters=True)  # a device copy of the image exists before the tune replaces it
    info = s.tune(w, h, cam, seed=5)
    assert info["tuned"] == 1 and info["probe_rows"] == 16 and info["probe_rays"] >= 1000 and info["probe_ms"] > 0
    assert info["box_tests_after"] < info["box_tests_before"]
    si = s.info()
    assert si["walk_tree"] == rt._abi.RT_WALK_TREE_TUNED and si["walk_tree_nodes"] == info["nodes_after"] and (si["n_nodes"] == info["nodes_before"] or KNOB)
    leaves = check_nary(*s.walk_tree())
    assert sorted(leaves) == sorted(int(p) for p in s.tree()[1] if p >= 0)
    tuned = s.render_rows(w, h, cam, seed=5, counters=True)
    _same_but_for_box_tests(first, base)
    _same_but_for_box_tests(tuned, base)
    assert tuned.stats["aabb_tests"] < base.stats["aabb_tests"] or KNOB
    timed = s.render_rows(w, h, cam, seed=5)  # the hand-written node loop over the thinned tree
    assert np.array_equal(timed.accum, base.accum)
    acc, rgb, st = orc.OracleScene(objs).render_rows(w, h, cam.to_abi(), seed=5, threads=8)
    assert np.array_equal(tuned.accum, acc) and np.array_equal(tuned.rgb, rgb) and tuned.stats["rays"] == st["rays"]
    # a fresh scene tuned before its first render, and the same probe, give the same tree
    t = rt.Scene.make(objs, walk_tree="sah")
    t.tune(w, h, cam, seed=5)
    assert all(np.array_equal(a, b) for a, b in zip(t.walk_tree(), s.walk_tree()))
    assert np.array_equal(t.render_rows(w, h, cam, seed=5).accum, base.accum)
    # another camera / seed for the probe: another tree, the same pixels
    u = rt.Scene.make(objs, walk_tree="sah")
    u.tune(w // 2, h // 2, cam, seed=99)
    assert np.array_equal(u.render_rows(w, h, cam, seed=5).accum, base.accum)


@pytest.mark.parametrize("name", ["spheres", "inside-sphere", "glass", "moved-camera", "textured-sphere"])
def test_catalogue_scenes_tuned(rt, orc, name):
    objs, cam, w, h = rt.sample_images.get(name)()
    cam = dataclasses.replace(cam, SamplesPerPixel=24)
    w, h = max(1, w // 20), max(1, h // 20)
    s = rt.Scene.make(objs, walk_tree="sah")
    base = s.render_rows(w, h, cam, seed=3, counters=True)
    s.tune(w, h, cam, seed=3)
    _same_but_for_box_tests(s.render_rows(w, h, cam, seed=3, counters=True), base)
    acc, rgb, st = orc.OracleScene(objs).render_rows(w, h, cam.to_abi(), seed=3, threads=8)
    assert np.array_equal(base.accum, acc)


@pytest.mark.parametrize("seed", range(16))
def test_random_scenes_tuned(rt, orc, seed):
    """The fuzz scenes of test_gpu_parity (every style, negative radii, duplicates, planes, textures, cameras inside objects), tuned:
    equal to the oracle."""
    objs, cam, w, h = scenes.random_scene(3000 + seed)
    s = rt.Scene.make(objs, walk_tree="sah")
    info = s.tune(w, h, cam, seed=seed)
    res = s.render_rows(w, h, cam, seed=seed, counters=True)
    plain = s.render_rows(w, h, cam, seed=seed)
    acc, rgb, st = orc.OracleScene(objs).render_rows(w, h, cam.to_abi(), seed=seed, threads=8)
    assert np.array_equal(res.accum, acc) and np.array_equal(res.rgb, rgb) and np.array_equal(plain.accum, acc)
    for k in ("rays", "prim_tests", "reflections", "samples", "pixels", "pixels_early"):
        assert res.stats[k] == st[k], k
    assert res.stats["aabb_tests"] <= st["aabb_tests"] or info["tuned"] == 0 or s.info()["n_bounded"] < 8


def test_tune_leaves_reference_tree_scenes_and_the_callers_device_alone(rt):
    import torch
    objs, cam, w, h = scenes.small_final()
    s = rt.Scene.make(objs, walk_tree="reference")
    before = s.walk_tree()
    assert s.tune(w, h, cam)["tuned"] == 0 and s.info()["walk_tree"] == 1
    assert all(np.array_equal(a, b) for a, b in zip(s.walk_tree(), before))
    dev = torch.cuda.current_device()
    rt.Scene.make(objs, walk_tree="sah").tune(w, h, cam)
    assert torch.cuda.current_device() == dev
    with pytest.raises(Exception):
        rt.Scene.make(objs, walk_tree="sah").tune(0, h, cam)
    with pytest.raises(Exception):
        rt.Scene.make(objs, walk_tree="sah").tune(w, h, cam, device=99)


def test_render_frame_over_a_tuned_scene(rt):
    """rt_render_frame (one process, several entries of the device list) after a tune: every device copy holds the new image."""
    objs, cam, w, h = scenes.small_final(spp=30, pixels=20)
    s = rt.Scene.make(objs, walk_tree="sah")
    want = s.render_rows(w, h, cam, seed=8)
    got0 = s.render_frame(w, h, cam, seed=8, devices=[0, 0])
    s.tune(w, h, cam, seed=8)
    got1 = s.render_frame(w, h, cam, seed=8, devices=[0, 0, 0])
    assert np.array_equal(got0.accum, want.accum) and np.array_equal(got1.accum, want.accum)


def test_full_size_config3_tuned(rt):
    """BASELINE config 3 whole, tuned as bench.py does: the frame and every counter but the box tests equal the untuned run's (which
    test_full_size_config3_equals_the_oracle holds to the oracle), at about 17 box tests per ray instead of 23.8."""
    objs, cam, w, h = rt.sample_images.config3_final()
    a = rt.Scene.make(objs, walk_tree="sah").render_rows(w, h, cam, seed=2024, counters=True)
    s = rt.Scene.make(objs, walk_tree="sah")
    info = s.tune(w, h, cam, seed=2024)
    b = s.render_rows(w, h, cam, seed=2024, counters=True)
    c = s.render_rows(w, h, cam, seed=2024)
    _same_but_for_box_tests(b, a)
    assert np.array_equal(c.accum, a.accum)
    assert info["tuned"] == 1 and (b.stats["aabb_tests"] < 0.8 * a.stats["aabb_tests"] or KNOB)
    assert abs(info["box_tests_after"] - b.stats["aabb_tests"] / b.stats["rays"]) < 1.0  # the probe's estimate against the frame's count
