#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's config, on N GPUs of one node.

Metric: Mray/s (primary + secondary rays = Scene.hitObject calls per second), whole job.
Workload at N=1 (and, row-sharded, at N>1): config 3 = the RTIOW final scene (SampleImages.randomSpheres), the reference's
"1200x800" half-extents = 2401x1601 pixels, 500 spp, 50 bounces, adaptive sampling as in Scene.renderPixel.
A "step" is one full frame: every rank renders its interleaved rows, then one gather of the accumulators to rank 0.

Launch: `python bench.py --gpus 1 --steps K --warmup W`, or for N>1
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VECTOR_PEAK_TF = 78.6  # MI355X vector FP64 spec: 256 CUs x 4 SIMDs x 16 lanes x 2 (FMA) x 2.4 GHz


def algorithmic_bytes(st) -> int:
    """56 B per ray-AABB test (6 doubles + 2 links), 32 B per ray-primitive test (centre + r^2),
    40 B material record per shaded vertex, 4 B packed RGB per finished sample."""
    return 56 * st["aabb_tests"] + 32 * st["prim_tests"] + 40 * st["reflections"] + 4 * st["samples"]


def algorithmic_flops(st) -> int:
    """18 flops per ray-AABB test (6 sub, 6 mul, 6 compare-select), 17 per ray-primitive test, 3 per ray
    (inverseDirections).  The exactness contract forbids contraction, so none of them can be half of an FMA."""
    return 18 * st["aabb_tests"] + 17 * st["prim_tests"] + 3 * st["rays"]


def measured_issue_ceiling():
    """TFLOP/s of non-FMA FP64 (one flop per lane per instruction) at the issue cost MEASURED on this chip: the newest
    profiles/r*/valu_rates.txt (output of scripts/ubench/valu_rates.hip) -- cycles per wave64 v_add_f64 / v_mul_f64 / v_max_f64 per
    SIMD at 4 waves per SIMD, at the clock that file reports.  (None, None) when no such file is committed."""
    import glob
    import re

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "valu_rates.txt")))
    if not files:
        return None, None
    txt = open(files[-1]).read()
    clk = re.search(r"clock (\d+) kHz", txt)
    row = [ln for ln in txt.splitlines() if ln.startswith("waves/SIMD 4:")]
    if not clk or not row:
        return None, None
    cyc = [float(re.search(name + r"=([0-9.]+)", row[0]).group(1)) for name in ("v_add_f64", "v_mul_f64", "v_max_f64")]
    mean = sum(cyc) / len(cyc)
    return 1024 * 64 * (float(clk.group(1)) * 1e3) / mean / 1e12, {"file": os.path.relpath(files[-1], ROOT), "cycles_per_wave64_f64_op": round(mean, 2),
                                                                   "clock_khz": int(clk.group(1))}


def profiled_traffic():
    """HBM bytes per launch of the render kernel from the newest committed PMC summary under profiles/ (separate
    rocprofv3 --pmc passes, scripts/pmc.sh): WRITE_SIZE + 2*FETCH_SIZE KiB (gfx950 reports half the fetched bytes,
    MI355X_MICROARCH.md "HBM").  None when no summary is committed."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_bench_config_summary.csv")))
    if not files:
        return None, None
    vals = dict(line.strip().split(",") for line in open(files[-1]) if "," in line and not line.startswith("#"))
    if "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals:
        return None, None
    phys = None
    try:  # the physical reading of the same profile: what actually limits the kernel
        v = {k: float(x) for k, x in vals.items()}
        # SQ_ACTIVE_INST_VALU is in quad-cycles summed over waves; GRBM_GUI_ACTIVE is cycles summed over the 8 XCDs; 1024 SIMDs
        simd_cycles = 1024.0 * v["GRBM_GUI_ACTIVE"] / 8.0
        phys = {"valu_issue_busy_frac": round(4.0 * v["SQ_ACTIVE_INST_VALU"] / simd_cycles, 3),
                "resident_waves_per_simd": round(4.0 * v["SQ_WAVE_CYCLES"] / simd_cycles, 2),
                "valu_lane_utilisation": round(v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_INSTS_VALU"] * 64.0), 3),
                "valu_wave_instructions": int(v["SQ_INSTS_VALU"]), "lds_wave_instructions": int(v["SQ_INSTS_LDS"])}
    except (KeyError, ZeroDivisionError):
        pass
    return int((2.0 * float(vals["FETCH_SIZE"]) + float(vals["WRITE_SIZE"])) * 1024), (os.path.relpath(files[-1], ROOT), phys)


def effective_cpus() -> int:
    """Host cores this process may actually use: the cgroup CPU quota when there is one (the GPU box shows 256 hardware
    threads but grants 16 CPUs; oversubscribing the quota throttles and makes the baseline look 1.5x slower), else the
    scheduler affinity."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]  # cgroup v2
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:  # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return max(1, n)


def cpu_baseline(objs, cam, w, h, seed, target_seconds):
    """The oracle (kind "port": a C++ restatement; the F#/.NET reference cannot run in this image) on all host cores, on a
    bounded sample of the SAME workload: every `stride`-th image row of the frame, sized by a short calibration pass so the
    timed pass is about `target_seconds` of CPU wall time."""
    import oracle as orc

    rows = 2 * h + 1
    threads = effective_cpus()
    scene = orc.OracleScene(objs)

    def run(stride):
        first = stride // 2
        n = len(range(first, rows, stride))
        t0 = time.perf_counter()
        _, _, st = scene.render_rows(w, h, cam.to_abi(), seed=seed, row_first=first, row_stride=stride, n_rows=n, threads=threads)
        return st, time.perf_counter() - t0, first, n

    # one host thread on a small sample (BASELINE.md 3.2 asks for the 1-thread figure beside the all-cores one)
    t0 = time.perf_counter()
    _, _, st1 = scene.render_rows(w, h, cam.to_abi(), seed=seed, row_first=rows // 2, row_stride=max(1, rows // 4), n_rows=2, threads=1)
    one_thread = st1["rays"] / (time.perf_counter() - t0) / 1e6

    cal_stride = max(1, rows // 32)
    st, dt, first, n = run(cal_stride)  # calibration: ~32 rows spread over the frame
    rate = st["rays"] / dt
    full_rays = st["rays"] * cal_stride
    stride = max(1, min(cal_stride, int(round(full_rays / max(rate * target_seconds, 1.0)))))
    if stride < cal_stride:
        st, dt, first, n = run(stride)
    return {"value": round(st["rays"] / dt / 1e6, 4), "unit": "Mray/s", "cores": threads, "hardware_threads": orc.hardware_threads(), "value_1_thread": round(one_thread, 4), "kind": "port",
            "sample": f"{n} of {rows} image rows (every {stride}th from row {first}) of the same frame: {st['rays']} rays in {dt:.1f} s; "
                      f"oracle = C++ restatement of the F# path (the .NET reference cannot run in this image)",
            "rays": st["rays"], "seconds": round(dt, 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=500)
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--pixels", type=int, default=800, help="the reference's `pixels` (maxHeightCoord); image is (3*pixels+1)x(2*pixels+1)")
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--workload", default="c3", choices=["c2", "c3", "c4", "c5"],
                    help="BASELINE.json config: c3 (default, the metric's config) final scene 2401x1601x500spp; c2 three Lambert spheres "
                         "801x451x100spp; c4 final scene 7681x4321x1000spp; c5 final scene + earth texture + plane + Dielectric, 2000spp")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target wall time of the CPU baseline sample (0 = skip)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL) is the product path; gloo stages the gather through host memory and lets several ranks share one GPU (rehearsal only)")
    ap.add_argument("--block", type=int, default=0)
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--no-untuned-leg", action="store_true", help="skip the legs that time the same frame WITHOUT rt_scene_tune and a scene's FIRST frame (tune + frame): "
                                                                  "for kernel traces, which should hold one configuration of the timed kernel")
    ap.add_argument("--untuned-leg", action="store_true", help=argparse.SUPPRESS)  # the default now; accepted for older command lines
    ap.add_argument("--no-tune", action="store_true", help="walk the surface-area tree as built, without rt_scene_tune's probe")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import ray_tracing_fsharp_amd as rt
    from ray_tracing_fsharp_amd import distributed as rtd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available() or rt.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    if args.backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()  # rehearsal: ranks may share a GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    if args.block or args.chunk:
        rt.set_launch_config(args.block, args.chunk)

    if args.workload == "c3":
        objs, cam, w, h = rt.sample_images.config3_final(seed=args.seed, spp=args.spp, depth=args.depth, pixels=args.pixels)
        label = "RTIOW final random-spheres scene (SampleImages.randomSpheres recipe"
    elif args.workload == "c2":
        objs, cam, w, h = rt.sample_images.config2_three_lambert()
        label = "config 2: three Lambert spheres + floor + dome (SURVEY.md 8d"
    elif args.workload == "c4":
        objs, cam, _, _ = rt.sample_images.config3_final(seed=args.seed, spp=1000, depth=args.depth)
        w, h = 3840, 2160
        label = "config 4: final scene at maxW=3840 maxH=2160 (SampleImages.randomSpheres recipe"
    else:
        import numpy as np
        earth = np.load(os.path.join(ROOT, "tests", "golden", "earthmap_rgb.npz"))["rgb"]
        objs, cam, w, h = rt.sample_images.config5_mixed(earth, seed=args.seed, depth=args.depth)
        label = "config 5: final scene + earth-textured sphere + Dielectric sphere + mirror InfinitePlane (recipe"
    scene = rt.Scene.make(objs)
    # Scene preparation, once per scene and camera, outside the timed steps like Scene.make itself: the walk tree rebuilt from the
    # rays of a 16-row probe render (rt_scene_tune).  Every rank probes the same rows of the whole frame, so all walk the same tree.
    # The probe has a seed of its own: the timed frames' rays are not the rays the tree was tuned on.
    tune = None if args.no_tune else scene.tune(w, h, cam, seed=args.seed ^ 0x5EED, device=local_rank)
    rows, cols = 2 * h + 1, 2 * w + 1
    first, stride, n = rtd.shard_rows(rows, rank, world)
    n_pad = (rows + world - 1) // world
    local = torch.zeros((n_pad, cols, 4), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev)

    def step():
        rtd.render_shard_device(scene, cam, w, h, args.seed, local_rank, first, stride, n, local, stream=stream.cuda_stream)
        return rtd.gather_frame(local, rows, cols, rank, world)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # the counters of one launch (deterministic for a seed): an untimed replica with the counting kernel variant
    st = rtd.render_shard_device(scene, cam, w, h, args.seed, local_rank, first, stride, n, local, stream=stream.cuda_stream,
                                 counters=True, want_stats=True)
    # the same launch over BoundingBoxTree.make's own tree: its box-test count is the reference's (and the oracle's); every other
    # counter and every pixel must be the walked (surface-area) tree's -- checked here on the real workload, untimed
    rt.set_walk_tree("reference")
    try:
        ref_scene = rt.Scene.make(objs)
    finally:
        rt.set_walk_tree("sah")
    check = torch.zeros_like(local)
    st_ref = rtd.render_shard_device(ref_scene, cam, w, h, args.seed, local_rank, first, stride, n, check, stream=stream.cuda_stream,
                                     counters=True, want_stats=True)
    torch.cuda.synchronize(dev)
    if not torch.equal(check, local) or any(st_ref[k] != st[k] for k in ("rays", "prim_tests", "reflections", "samples", "pixels_early")):
        raise SystemExit("walking the reference's own tree and the tree of this run gave different results")
    del check, ref_scene
    step()  # untimed, whatever --warmup says: first launch of the timed kernel variant, stream-ordered pool set-up, and (N>1)
            # the first gather, which sets up RCCL's peer-to-peer connections over xGMI
    for _ in range(args.warmup):
        step()
    fence()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    k_ms = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ev0.record(stream)  # HIP events on the stream the kernel is launched on
        rtd.render_shard_device(scene, cam, w, h, args.seed, local_rank, first, stride, n, local, stream=stream.cuda_stream)
        ev1.record(stream)
        frame = rtd.gather_frame(local, rows, cols, rank, world)
        ev1.synchronize()
        k_ms += ev0.elapsed_time(ev1)
    fence()
    dt = time.perf_counter() - t0

    # the same frame over the surface-area tree as built (no rt_scene_tune), timed the same way: both numbers from one run
    untuned = None
    first_frame = None
    if world == 1 and tune is not None and tune["tuned"] and not args.no_untuned_leg:
        plain = rt.Scene.make(objs)
        st_plain = rtd.render_shard_device(plain, cam, w, h, args.seed, local_rank, first, stride, n, local, stream=stream.cuda_stream,
                                           counters=True, want_stats=True)
        rtd.render_shard_device(plain, cam, w, h, args.seed, local_rank, first, stride, n, local, stream=stream.cuda_stream)
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            rtd.render_shard_device(plain, cam, w, h, args.seed, local_rank, first, stride, n, local, stream=stream.cuda_stream)
            rtd.gather_frame(local, rows, cols, rank, world)
        fence()
        ms_plain = (time.perf_counter() - t1) * 1e3 / args.steps
        untuned = {"ms_per_step": round(ms_plain, 3), "value": round(st_plain["rays"] / (ms_plain * 1e-3) / 1e6, 3),
                   "aabb_tests": int(st_plain["aabb_tests"]), "note": "the same frame without rt_scene_tune (surface-area walk tree as built)"}
        del plain
        # A scene's FIRST frame, as the reference renders it (ONE frame per scene, SampleImages.fs:958-960): rt_scene_tune and the frame,
        # wall clock, on a fresh scene (kernels already loaded by the legs above).  The headline `value` is the steady state of a scene
        # that is rendered again and again; this is the other number, from the same run.
        ff = []
        for _ in range(3):
            fresh = rt.Scene.make(objs)
            fence()
            t2 = time.perf_counter()
            info_ff = fresh.tune(w, h, cam, seed=args.seed ^ 0x5EED, device=local_rank)
            rtd.render_shard_device(fresh, cam, w, h, args.seed, local_rank, first, stride, n, local, stream=stream.cuda_stream)
            rtd.gather_frame(local, rows, cols, rank, world)
            fence()
            ff.append(((time.perf_counter() - t2) * 1e3, info_ff))
            del fresh
        ff_ms, ff_info = min(ff, key=lambda x: x[0])
        first_frame = {"ms": round(ff_ms, 3), "tune_probe_ms": round(ff_info["probe_ms"], 3), "tune_build_ms": round(ff_info["build_ms"], 3),
                       "untuned_frame_ms": round(ms_plain, 3),
                       "note": "rt_scene_tune + one frame on a fresh scene, wall clock, best of 3; `untuned_frame_ms` is the same frame with no tune at all: "
                               "whichever is smaller is what ONE frame of this scene costs"}

    tot = torch.tensor([dt, float(st["rays"]), float(st["aabb_tests"]), float(st["prim_tests"]), float(st["reflections"]),
                        float(st["samples"]), float(st["pixels_early"]), k_ms / max(1, args.steps), float(st_ref["aabb_tests"])],
                       dtype=torch.float64, device=dev)
    if world > 1:
        if args.backend == "gloo":
            tot = tot.cpu()
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = tot.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        dt = float(mx[0])
        job = {k: int(sm[i + 1]) for i, k in enumerate(("rays", "aabb_tests", "prim_tests", "reflections", "samples", "pixels_early"))}
        kernel_ms = float(mx[7])
        job["aabb_tests_reference_tree"] = int(sm[8])
    else:
        job = {k: int(st[k]) for k in ("rays", "aabb_tests", "prim_tests", "reflections", "samples", "pixels_early")}
        job["aabb_tests_reference_tree"] = int(st_ref["aabb_tests"])
        kernel_ms = float(tot[7])

    if rank == 0:
        ms_per_step = dt * 1e3 / args.steps
        value = job["rays"] / (ms_per_step * 1e-3) / 1e6
        # dominant (only) kernel: rtd::render_kernel.  What binds it is VALU issue of FP64 work (no MFMA: nothing here is a
        # contraction; no HBM: the scene lives in LDS), so the roofline is the FP64 vector one: this rank's algorithmic flops per
        # launch / its mean launch time, against the 78.6 TF spec, and beside it against what this chip was MEASURED to issue when
        # every instruction is a plain add/mul/max (no FMA: contraction is off by contract).
        rank_bytes = algorithmic_bytes(st)
        rank_flops = algorithmic_flops(st)
        achieved_tf = rank_flops / (kernel_ms * 1e-3) / 1e12
        # the same with the box tests of the reference's own tree ("N_* must equal the oracle's counters"): the work
        # the reference's algorithm does for this frame, of which the walked tree skips a part
        ref_flops = algorithmic_flops({**st, "aabb_tests": st_ref["aabb_tests"]})
        ceiling_tf, ceiling_src = measured_issue_ceiling()
        info = scene.info()
        traffic, src = profiled_traffic() if (world == 1 and args.workload == "c3") else (None, None)  # the committed PMC passes are of config 3
        traffic_src, physical = src if src else (None, None)
        limited = None
        if physical:
            limited = (f"VALU issue: a vector instruction in flight {100 * physical['valu_issue_busy_frac']:.0f} % of all SIMD cycles at "
                       f"{100 * physical['valu_lane_utilisation']:.0f} % lane occupancy, {physical['valu_wave_instructions']:.3g} wave instructions per frame "
                       f"(PMC, {traffic_src}); neither HBM nor MFMA")
        out = {
            "metric": "Mray/s (primary+secondary)", "value": round(value, 3), "unit": "Mray/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "backend": args.backend if world > 1 else None,
            "config": {"workload": f"{label}, seed {args.seed}), "
                                   f"maxW={w} maxH={h} -> {cols}x{rows} px, {cam.SamplesPerPixel} spp adaptive, {cam.BounceDepth} bounces",
                       "spheres_bounded": info["n_bounded"], "unbounded": info["n_unbounded"], "tree_nodes": info["n_nodes"], "walk_tree_nodes": info["walk_tree_nodes"],
                       "lds_resident_scene": bool(info["lds_resident"]),
                       "walk_tree": ("surface-area build over the reference's leaf boxes (same hits; DESIGN.md 'Walk tree')",
                                     "BoundingBoxTree.make's own",
                                     "rebuilt from the rays of a probe render and thinned (rt_scene_tune; same hits; DESIGN.md 'Walk tree')")[info["walk_tree"]],
                       "tune": tune, "without_tune": untuned, "sharding": f"rows interleaved over {world} rank(s), one gather"},
            "job": {**job, "pixels": rows * cols, "wall_s_per_frame": round(ms_per_step / 1e3, 4),
                    "first_frame_ms": first_frame["ms"] if first_frame else None, "first_frame": first_frame,
                    "ray_sphere_tests_per_s": round(job["prim_tests"] / (ms_per_step * 1e-3), 1),
                    "aabb_tests_per_s": round(job["aabb_tests"] / (ms_per_step * 1e-3), 1),
                    "samples_per_s": round(job["samples"] / (ms_per_step * 1e-3), 1),
                    "primary_rays_per_s": round(job["samples"] / (ms_per_step * 1e-3), 1),
                    "mean_hits_per_path": round(job["reflections"] / max(1, job["samples"]), 4),
                    "early_exit_fraction": round(job["pixels_early"] / (rows * cols), 4)},
            "roofline": {"bound": "fp64_vector", "achieved": round(achieved_tf, 3), "peak": FP64_VECTOR_PEAK_TF, "unit": "TFLOP/s",
                         "frac": round(achieved_tf / FP64_VECTOR_PEAK_TF, 5),
                         "peak_measured_non_fma_issue": round(ceiling_tf, 2) if ceiling_tf else None,
                         "frac_of_measured_non_fma_issue": round(achieved_tf / ceiling_tf, 4) if ceiling_tf else None,
                         "measured_issue_source": ceiling_src,
                         "traffic": traffic, "traffic_source": traffic_src, "physical_from_same_profile": physical,
                         "kernel": "rtd::render_kernel", "kernel_ms": round(kernel_ms, 3), "algorithmic_flops_per_launch": rank_flops,
                         "reference_tree_equiv": {"flops_per_launch": ref_flops, "tflops": round(ref_flops / (kernel_ms * 1e-3) / 1e12, 3),
                                                  "note": "the box tests BoundingBoxTree.make's own tree would have cost (the oracle's count), / the same "
                                                          "kernel time: work the reference's algorithm does per frame -- NOT executed here, not a roofline fraction"},
                         "limited_by": limited,
                         "algorithmic_hbm_equiv": {"bytes_per_launch": rank_bytes, "gb_per_s": round(rank_bytes / (kernel_ms * 1e-3) / 1e9, 1),
                                                   "of_hbm_peak": round(rank_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 3),
                                                   "note": "SURVEY.md 8(d)'s scene bytes the algorithm touches (56 B per box test, ...), served by LDS -- "
                                                           "not HBM traffic and not a roofline fraction; `traffic` is the physical HBM byte count per launch"},
                         "note": "achieved = algorithmic flops of the tests made (SURVEY.md 8(d): 18 per box test of the walked tree, 17 per sphere test, 3 per ray) / mean "
                                 "kernel time.  Since round 3 the box tests ABOVE the leaves run as a conservative single-precision filter (same visits, "
                                 "same hits; DESIGN.md section 4) -- the flops are counted at SURVEY's double-precision figure, the peak is the FP64 vector one"},
        }
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(objs, cam, w, h, args.seed, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
