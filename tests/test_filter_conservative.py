"""The timed kernel's node loop replaces BoundingBox.hits above the leaves by a single-precision filter (csrc/rt_device.h).  What
makes that legitimate is one implication -- exact hit => filter hit -- for every ray and every box.  Here: a numpy model of the
filter against the oracle's literal BoundingBox.hits (BoundingBox.fs:30-94) on cases built for the purpose (tests/filter_cases.py),
with the hardware reciprocal's one-ulp freedom bracketed.  The device's own filter is held to the device's exact test over the same
cases in tests/test_gpu_parity.py::test_single_precision_filter_never_loses_a_hit."""
import numpy as np

import filter_cases as fc


def test_filter_model_is_conservative(orc):
    n = 60_000
    total = hits = false_pos = 0
    for name, rays, boxes in fc.classes(n, seed=20261004):
        exact = orc.bbox_hits(rays, boxes).astype(bool)
        for ulps in (-1, 0, 1):
            f = fc.model(rays, boxes, rcp_ulps=ulps)
            lost = exact & ~f
            assert not lost.any(), f"{name}: the filter lost {int(lost.sum())} hits (rcp {ulps:+d} ulp), first at {int(np.flatnonzero(lost)[0])}"
        total += n
        hits += int(exact.sum())
        false_pos += int((fc.model(rays, boxes) & ~exact).sum())
    assert hits > total // 10  # the generators do aim at the boxes
    # and the filter is a filter: on ordinary scales it passes very few boxes the exact test rejects
    for name, rays, boxes in fc.classes(n, seed=7):
        if name.startswith("random_scale2_") or name.startswith("random_scale10_"):
            exact = orc.bbox_hits(rays, boxes).astype(bool)
            extra = int((fc.model(rays, boxes) & ~exact).sum())
            assert extra <= n // 2000, f"{name}: {extra} false positives of {n}"


def test_filter_model_with_a_larger_margin_scale_is_still_conservative(orc):
    # the scene image takes the largest |coordinate| of the whole tree as the margin's scale: larger than a single box needs
    for name, rays, boxes in fc.classes(20_000, seed=99):
        exact = orc.bbox_hits(rays, boxes).astype(bool)
        for bmax in (10.0, 2000.0, 1e9):
            assert not (exact & ~fc.model(rays, boxes, bmax=bmax)).any(), (name, bmax)
