"""ImageOutput.resume's temp-file format and readPixelMap (ImageOutput.fs:46-161): product writer/reader against the
oracle's literal restatement, the zero-as-no-digits quirk, truncated tails, and the reference's own golden-file flow."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _fmt(rt, px):
    import ctypes as C
    px = np.ascontiguousarray(px, np.uint8)
    n = rt.lib.rt_format_pixel_map(px.ctypes.data_as(C.POINTER(C.c_uint8)), px.shape[0], px.shape[1], None, 0)
    buf = C.create_string_buffer(int(n))
    rt.lib.rt_format_pixel_map(px.ctypes.data_as(C.POINTER(C.c_uint8)), px.shape[0], px.shape[1], buf, int(n))
    return buf.raw[: int(n)]


def test_writer_matches_the_oracle_and_zero_has_no_digits(rt, orc):
    rng = np.random.default_rng(1)
    px = rng.integers(0, 256, size=(13, 11, 3), dtype=np.uint8)
    px[0, 0] = [10, 44, 48]  # colour bytes that look like '\\n', ',' and '0' are raw data, not syntax
    data = _fmt(rt, px)
    assert data == orc.format_pixel_map(px)
    assert data.startswith(b",\n" + bytes([10, 44, 48]) + b",1\n")  # pixel (0,0) then (0,1): zeros written as nothing (ImageOutput.fs:115-129)
    assert b"12,10\n" in data


def test_reader_round_trip_truncation_and_errors(rt, orc, tmp_path):
    rng = np.random.default_rng(2)
    px = rng.integers(0, 256, size=(9, 7, 3), dtype=np.uint8)
    data = orc.format_pixel_map(px)
    path = tmp_path / "spill.bin"
    path.write_bytes(data)
    rgb, present = rt.ImageOutput.readPixelMap(str(path), 9, 7)
    assert present.all() and np.array_equal(rgb, px)
    assert np.array_equal(rt.ImageOutput.assertComplete((rgb, present)), px)
    for cut in (0, 1, 5, len(data) // 2, len(data) - 1):  # a truncated tail is ignored, as the reference's `go` does
        path.write_bytes(data[:cut])
        rgb2, present2 = rt.ImageOutput.readPixelMap(str(path), 9, 7)
        n, orgb, opresent = orc.parse_pixel_map(data[:cut], 9, 7)
        assert np.array_equal(present2, opresent) and np.array_equal(rgb2, orgb) and int(present2.sum()) == n
        if cut < len(data):
            with pytest.raises(ValueError):
                rt.ImageOutput.assertComplete((rgb2, present2))
    path.write_bytes(b"99,0\nabc")
    with pytest.raises(rt.RtError):
        rt.ImageOutput.readPixelMap(str(path), 9, 7)


def test_reference_golden_flow(rt, tmp_path):
    """TestPpmOutput.fs:12-46 end to end: Image -> toPpm (spill) -> readPixelMap -> assertComplete -> writePpm false."""
    expected = open(os.path.join(HERE, "golden", "PpmOutputExample.txt"), "rb").read().replace(b"\r\n", b"\n")
    image = rt.Image.make(2, 3, np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255]], [[255, 255, 0], [255, 255, 255], [0, 0, 0]]], np.uint8))
    temp = rt.ImageOutput.toPpm(lambda _: None, image, str(tmp_path / "temp"))
    arr = rt.ImageOutput.assertComplete(rt.ImageOutput.readPixelMap(temp, rt.Image.rowCount(image), rt.Image.colCount(image)))
    out = tmp_path / "out.ppm"
    rt.ImageOutput.writePpm(False, lambda _: None, arr, str(out))
    assert out.read_bytes() == expected
