"""The C-ABI library loads and exports every symbol include/rtfs_amd.h declares; struct layouts match the ctypes mirror;
argument errors are reported the way the header promises.  No GPU compute is attempted here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "rtfs_amd.h")).read()


def declared_functions():
    text = re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(rt):
    from ray_tracing_fsharp_amd import _lib

    names = declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(_lib.lib, n), f"{n} is declared in include/rtfs_amd.h but not exported by librtfs_amd.so"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert set(_lib.SIGNATURES) <= set(names)


def test_the_library_is_the_in_tree_hip_build(rt):
    assert os.path.samefile(os.path.dirname(rt.LIB_PATH), os.path.join(ROOT, "ray-tracing-fsharp_amd"))
    blob = open(rt.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"render_kernel" in blob  # carries a gfx950 code object with the render kernel


def test_struct_sizes_and_version(rt):
    A = rt._abi
    for i, t in enumerate((A.rt_hittable, A.rt_texture, A.rt_camera, A.rt_scene_info, A.rt_stats)):
        assert rt.lib.rt_abi_sizeof(i) == C.sizeof(t)
    assert rt.lib.rt_abi_version() == A.RT_ABI_VERSION


def test_no_gpu_means_a_loud_error_not_a_fallback(rt):
    if rt.device_count() > 0:
        pytest.skip("a GPU is visible")
    objs, cam, w, h = rt.sample_images.config1_empty()
    with pytest.raises(rt.RtError) as e:
        rt.Scene.make(objs).render_rows(w, h, cam)
    assert e.value.code == rt._abi.RT_ERR_NO_DEVICE and "no CPU fallback" in e.value.message
    with pytest.raises(rt.RtError):
        rt.hooks.float_producer((1, 2, 3, 4), 4)


def test_argument_errors(rt):
    A = rt._abi
    objs, cam, w, h = rt.sample_images.config1_empty()
    s = rt.Scene.make(objs)
    for kw in (dict(maxWidthCoord=0, maxHeightCoord=5), dict(maxWidthCoord=5, maxHeightCoord=-1)):
        with pytest.raises(rt.RtError) as e:
            s.render_rows(camera=cam, **kw)
        assert e.value.code == A.RT_ERR_INVALID_ARGUMENT
    with pytest.raises(rt.RtError) as e:
        s.render_rows(5, 5, cam, row_first=11, n_rows=1)  # rows = 11: row 11 does not exist
    assert e.value.code == A.RT_ERR_INVALID_ARGUMENT
    with pytest.raises(rt.RtError) as e:
        rt.set_launch_config(300, 0)
    assert e.value.code == A.RT_ERR_INVALID_ARGUMENT
    # degenerate camera: view direction parallel to view-up (the reference's ValueOption.get throws, Plane.fs:87-96)
    with pytest.raises(rt.RtError):
        rt.Camera.makeBasic(1, 1.0, 1.0, rt.Point.make(0, 0, 0), rt.Vector.make(0.0, 1.0, 0.0), rt.Vector.make(0.0, 1.0, 0.0))


def test_scene_validation(rt):
    A = rt._abi
    h = (A.rt_hittable * 1)()
    h[0].kind, h[0].style, h[0].texture = 7, 0, -1
    out = C.c_void_p()
    assert rt.lib.rt_scene_create(h, 1, None, 0, C.byref(out)) == A.RT_ERR_INVALID_ARGUMENT
    h[0].kind, h[0].style = A.RT_HITTABLE_SPHERE, 9
    assert rt.lib.rt_scene_create(h, 1, None, 0, C.byref(out)) == A.RT_ERR_INVALID_ARGUMENT
    h[0].style, h[0].texture = A.RT_SPHERE_GLASS, 3
    assert rt.lib.rt_scene_create(h, 1, None, 0, C.byref(out)) == A.RT_ERR_INVALID_ARGUMENT
    assert b"texture index" in rt.lib.rt_last_error()
    t = (A.rt_texture * 1)()
    t[0].kind = 17  # a closure kind that cannot cross the ABI
    h[0].texture = 0
    assert rt.lib.rt_scene_create(h, 1, t, 1, C.byref(out)) == A.RT_ERR_UNSUPPORTED
    with pytest.raises(rt.RtError):
        rt.ParameterisedTexture.Arbitrary(lambda x, y: None)
    # a plane style that carries a Pixel cannot take a texture
    h[0].kind, h[0].style, h[0].texture = A.RT_HITTABLE_INFINITE_PLANE, A.RT_PLANE_PURE_REFLECTION, 0
    t[0].kind = A.RT_TEXTURE_COLOUR
    assert rt.lib.rt_scene_create(h, 1, t, 1, C.byref(out)) == A.RT_ERR_INVALID_ARGUMENT


def test_empty_scene_is_allowed(rt):
    s = rt.Scene.make([])
    info = s.info()
    assert info["n_bounded"] == 0 and info["n_nodes"] == 0 and info["n_unbounded"] == 0


def _build_c_smoke(tmp_path):
    import subprocess
    exe = str(tmp_path / "abi_smoke")
    libdir = os.path.join(ROOT, "ray-tracing-fsharp_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "abi_smoke.c"),
                           "-L", libdir, "-lrtfs_amd", f"-Wl,-rpath,{libdir}", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-lm", "-o", exe])
    return exe


def test_header_is_plain_c99_and_a_c_program_can_use_the_abi(tmp_path):
    """include/rtfs_amd.h must be consumable from C (a P/Invoke or cgo maintainer reads it as C): syntax-check it as C99 with
    -pedantic, then build and run tests/c/abi_smoke.c (host-side entry points; the render call must refuse without a GPU)."""
    import subprocess
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-fsyntax-only", "-x", "c", os.path.join(ROOT, "include", "rtfs_amd.h")])
    out = subprocess.run([_build_c_smoke(tmp_path)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "host checks ok" in out.stdout


@pytest.mark.gpu
def test_c_program_renders_on_the_gpu(tmp_path):
    import subprocess
    out = subprocess.run([_build_c_smoke(tmp_path)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "rendered 7x7 px" in out.stdout
