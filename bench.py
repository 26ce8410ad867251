#!/usr/bin/env python3
"""Headline benchmark: Mrays/s and s/frame, teapot_tri.obj 2048x2048 @ 64 spp (BASELINE.json config 3).

A step = one frame of the canonical scene (raytrace/src/main.rs:116-173: teapot_tri.obj + two mirror
disks, octree (maxdepth 10, minobjs 19), depth 5) through HipRayCaster.  With N > 1 ranks the frame is
tiled by interleaved row stripes, one rank per GPU, and collected by ONE gather (RCCL) on rank 0 inside
the timed region; the total work is fixed, so scaling is "strong".  Rays = project_ray calls with
depth > 0 (raytrace.rs:1278), the reference's "Rays" statistic.

Launch: `python bench.py --gpus N` starts the N ranks itself (torch.distributed.run, before this process
touches a GPU) when it is not already running under a launcher; under `python -m torch.distributed.run
--nproc-per-node N bench.py --gpus N` it is one of the ranks.

Prints one JSON line (rank 0).  Extra legs outside the timed region (rank 0 of a 1-GPU run only): a counting
pass (device work counters -> algorithmic flops and bytes for the roofline object), a second timed loop
that includes the frame's device-to-host copy, and the CPU baseline (the oracle on a bounded sample).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X_MICROARCH.md: 8.0 TB/s HBM spec; FP32 vector peak 157.3 TFLOP/s counts an FMA as 2 flops.  This path may
# not contract a*b+c (bit parity with the reference), so every VALU lane-operation is ONE flop:
# the bound that applies is 157.3 / 2 = 78.65 TFLOP/s.
HBM_PEAK_GBS = 8000.0
FP32_FMA_PEAK_TF = 157.3
FP32_NOFMA_PEAK_TF = FP32_FMA_PEAK_TF / 2.0

CONFIGS = {  # BASELINE.json configs that run on the GPU
    2: dict(scene="linear", width=512, height=512, spp=16),
    3: dict(scene="canonical", width=2048, height=2048, spp=64),
    4: dict(scene="canonical", width=4096, height=4096, spp=256),
    5: dict(scene="grid", width=2048, height=2048, spp=64),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS),
                    help="BASELINE.json config: 3 = headline (default), 2 linear list, 4 = 4096x4096 @ 256 spp, 5 = 8-teapot grid")
    ap.add_argument("--width", type=int)
    ap.add_argument("--height", type=int)
    ap.add_argument("--spp", type=int)
    ap.add_argument("--scene", choices=["canonical", "grid", "linear"])
    ap.add_argument("--maxdepth", type=int, default=5)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--stripe-rows", type=int, default=16)
    ap.add_argument("--u8", action="store_true",
                    help="quantise every band on its GPU ((c*255.) as u8, raytrace.rs:1468-1473) and gather 3 B/pixel instead of 16")
    ap.add_argument("--fast", action="store_true", help="RTMI_OPT_FAST (not bit-exact, NOT the headline): skip boxes behind the ray origin")
    ap.add_argument("--bvh", action="store_true",
                    help="RTMI_OPT_BVH fast mode (NOT the headline): linear-list semantics through the library's SAH BVH; "
                         "prints the differing-pixel count against the exact octree render")
    ap.add_argument("--backend", default="nccl", help="nccl (RCCL, one rank per GPU) or gloo (rehearsal: ranks may share a GPU)")
    ap.add_argument("--check", action="store_true", help="rank 0 verifies the gathered frame against a single-tile render")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-counters", action="store_true")
    ap.add_argument("--no-d2h-leg", action="store_true", help="skip the extra timed loop that includes the frame's device-to-host copy")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="target wall time of the CPU baseline sample")
    args = ap.parse_args()
    preset = CONFIGS[args.config]
    for k, v in preset.items():
        if getattr(args, k) is None:
            setattr(args, k, v)
    return args


def self_launch(args):
    """`python bench.py --gpus N` from a bare shell: start the N ranks as children BEFORE this process touches a GPU
    (a process that has initialised HIP must not exec or fork GPU work) and exit with their status."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def oracle_flags():
    try:
        for line in open(os.path.join(ROOT, "oracle", "Makefile")):
            if line.startswith("CXXFLAGS"):
                return "g++ " + line.split("=", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    args = parse()
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        raise SystemExit(self_launch(args))

    import numpy as np
    import torch
    import torch.distributed as dist

    from rust_raytrace_amd import dist as rdist
    from rust_raytrace_amd import raytrace as R

    rank = int(os.environ.get("RANK", "0"))
    world = int(world_env or "1")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {world}")
    ndev = torch.cuda.device_count()
    if args.backend == "gloo":
        local_rank = local_rank % max(ndev, 1)  # rehearsal: several ranks on one GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    obj = os.path.join(ROOT, "tests", "golden", "teapot_tri.obj")
    t0 = time.time()
    # the octree is built with the overlap tests on this rank's GPU (rtmi_builder_*; bit-equal to the host build), so
    # the ranks of a node do not compete for host cores
    if args.scene == "grid":
        scene = R.grid_scene(obj, gpu_build=local_rank)
    elif args.scene == "linear":
        scene = R.canonical_scene(os.path.join(ROOT, "tests", "golden", "teapot.obj"), accel="trivial")
    else:
        scene = R.canonical_scene(obj, gpu_build=local_rank)  # octree (10, 19)
    t_build = time.time() - t0
    W, H, spp = args.width, args.height, args.spp
    vp = R.canonical_viewport(W, H, args.maxdepth, spp)
    caster = R.HipRayCaster(seed=args.seed, device=local_rank, options=(R.OPT_FAST if args.fast else 0) | (R.OPT_BVH if args.bvh else 0))
    t0 = time.time()
    caster.upload(scene)
    t_upload = time.time() - t0

    tile = rdist.rank_tile(rank, world, H, args.stripe_rows)
    local = torch.zeros((tile[1], W, 4), dtype=torch.float32, device=dev)
    local_u8 = torch.zeros((tile[1], W, 3), dtype=torch.uint8, device=dev) if args.u8 else None
    stream = torch.cuda.current_stream(dev)
    gather = rdist.FrameGather(rank, world, H, W, args.stripe_rows, channels=3 if args.u8 else 4,
                               dtype=torch.uint8 if args.u8 else torch.float32, device=dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]

    def step():
        ev[0].record(stream)
        ctx = caster.walk_tile_device(vp, scene, tile, local.data_ptr(), stream.cuda_stream)
        band = local
        if args.u8:
            caster.quantize_device(scene, local.data_ptr(), tile[1] * W, local_u8.data_ptr(), stream.cuda_stream)
            band = local_u8
        ev[1].record(stream)
        frame = gather(band)
        ev[2].record(stream)
        return ctx, frame

    first_gather_ms = None  # the first gather of the process: communicator set-up (RCCL ring/tree construction) included
    for _ in range(args.warmup):
        step()
        if first_gather_ms is None:
            ev[2].synchronize()
            first_gather_ms = ev[1].elapsed_time(ev[2])
    barrier()
    t0 = time.perf_counter()
    rays = 0
    trace_ms = kernel_ms = render_ms = gather_ms = 0.0
    launches = 0
    frame = None
    for _ in range(args.steps):
        ctx, frame = step()
        rays += ctx.total_rays
        trace_ms += ctx.stats["trace_ms"]
        kernel_ms += ctx.stats["kernel_ms"]
        launches += ctx.stats["trace_launches"]
        ev[2].synchronize()
        render_ms += ev[0].elapsed_time(ev[1])
        gather_ms += ev[1].elapsed_time(ev[2])
    barrier()
    dt = time.perf_counter() - t0

    mine = torch.tensor([float(rays), dt, render_ms / args.steps, gather_ms / args.steps, t_build, t_upload,
                         first_gather_ms if first_gather_ms is not None else -1.0], dtype=torch.float64, device=dev)
    per_rank = [mine]
    if world > 1:
        per_rank = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(per_rank, mine)
        dt = max(float(p[1]) for p in per_rank)
        rays = sum(float(p[0]) for p in per_rank)
    n_ranks_seen = len(per_rank)
    rank_render_ms = [round(float(p[2]), 3) for p in per_rank]
    rank_gather_ms = [round(float(p[3]), 3) for p in per_rank]
    rank_build_s = [round(float(p[4]), 3) for p in per_rank]
    rank_upload_s = [round(float(p[5]), 3) for p in per_rank]
    rank_first_gather_ms = [round(float(p[6]), 3) if float(p[6]) >= 0 else None for p in per_rank]

    roofline = None
    cpu_baseline = None
    incl_d2h = None
    if rank == 0:
        if world == 1 and not args.no_d2h_leg:
            # ---- frame device-to-host copy inside the window (SURVEY 8d's GPU window); never the headline `value`
            host = torch.empty(frame.shape, dtype=frame.dtype, pin_memory=True)
            n2 = min(args.steps, 3)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(n2):
                _, fr = step()
                host.copy_(fr, non_blocking=True)
                torch.cuda.synchronize(dev)
            d2 = time.perf_counter() - t1
            incl_d2h = {"value": round(rays / args.steps * n2 / d2 / 1e6, 2), "unit": "Mrays/s", "ms_per_step": round(d2 / n2 * 1e3, 3),
                        "note": f"timed window also contains the {host.numel() * host.element_size() >> 20} MiB frame copy to pinned host memory"}
        if world == 1 and not args.no_counters and not args.bvh:
            roofline = roofline_object(args, caster, scene, vp, tile, local, stream, rays / args.steps, trace_ms, kernel_ms, launches, dt)
        if world == 1 and not args.no_cpu_baseline and args.scene == "canonical":
            cpu_baseline = cpu_baseline_object(args, obj)
        fast_vs_exact = None
        if (args.bvh or args.fast) and world == 1:
            # how far the opt-in mode is from the exact (default) render of the same frame
            ex = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
            ectx = R.HipRayCaster(seed=args.seed, device=local_rank).walk_tile_device(vp, scene, (0, H, H, 0), ex.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize(dev)
            diff = (ex.view(torch.int32) != frame.view(torch.int32)).any(dim=2)
            fast_vs_exact = {"differing_pixels": int(diff.sum()), "pixels": H * W, "max_abs_diff": float((ex - frame).abs().max()),
                             "rays_exact": int(ectx.total_rays), "rays_this_mode": int(rays / args.steps)}
        if args.check:
            ref = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
            caster.walk_tile_device(vp, scene, (0, H, H, 0), ref.data_ptr(), stream.cuda_stream)
            if args.u8:
                q = torch.zeros((H, W, 3), dtype=torch.uint8, device=dev)
                caster.quantize_device(scene, ref.data_ptr(), H * W, q.data_ptr(), stream.cuda_stream)
                torch.cuda.synchronize(dev)
                from oracle import orc  # checker only (--check): the oracle's quantisation of the single-tile frame
                same = bool(torch.equal(q, frame)) and bool(np.array_equal(orc.quantize(ref.cpu().numpy()).reshape(H, W, 3), frame.cpu().numpy()))
            else:
                torch.cuda.synchronize(dev)
                same = bool(torch.equal(ref.view(torch.int32), frame.view(torch.int32)))
            print(f"[bench] gathered frame == single-tile render: {same}", file=sys.stderr)
            if not same:
                raise SystemExit("gathered frame differs from the single-tile render")
        value = rays / dt / 1e6
        workload = {"canonical": "canonical main.rs scene (teapot_tri.obj + 2 mirror disks, 6721 triangles), octree (10,19), ",
                    "grid": "config 5: 8 x teapot_tri.obj grid (50561 triangles), octree (10,19), ",
                    "linear": "config 2: canonical scene from teapot.obj, trivial bounding box (linear list of 6720 triangles), "}[args.scene]
        workload += f"{W}x{H} @ {spp} spp, depth {args.maxdepth}, seed {args.seed}" + (" [RTMI_OPT_FAST: not bit-exact]" if args.fast else "")
        workload += " [RTMI_OPT_BVH fast mode: linear-list semantics, NOT the octree traversal]" if args.bvh else ""
        headline = args.scene == "canonical" and (W, H, spp) == (2048, 2048, 64) and not args.fast and not args.bvh
        out = {
            "metric": "Mrays/s (primary + bounce rays per second of frame time)" + (", teapot_tri.obj 2048x2048 @64spp" if headline else f", BASELINE config {args.config} workload, see config.workload"),
            "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "s_per_frame": round(dt / args.steps, 5),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload,
                       "tiling": f"{world} x interleaved {args.stripe_rows}-row stripes + one gather of " + ("3 B/pixel (u8 RGB, quantised on each GPU)" if args.u8 else "16 B/pixel (f32 x 4)"),
                       "rays_per_frame": int(rays / args.steps)},
            "ranks": {"n_ranks_seen": n_ranks_seen, "backend": args.backend if world > 1 else None,
                      "render_ms_per_rank": rank_render_ms, "gather_ms_per_rank": rank_gather_ms,
                      "first_gather_ms_per_rank": rank_first_gather_ms,  # warm-up step 1: communicator set-up included (null without warm-up)
                      "octree_build_s_per_rank": rank_build_s, "scene_upload_s_per_rank": rank_upload_s},
            "value_incl_frame_d2h": incl_d2h, "opt_in_mode_vs_exact": fast_vs_exact,
            "roofline": roofline, "cpu_baseline": cpu_baseline,
            "setup": {"octree_build_s": round(t_build, 2), "scene_upload_s": round(t_upload, 3)},
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def roofline_object(args, caster, scene, vp, tile, local, stream, rays_per_frame, trace_ms, kernel_ms, launches, dt):
    """Dominant kernel = the closest-hit kernel of the bounce passes (k_trace_oct, 4 launches per single-stream frame, ~2/3
    of the device time; k_trace_linear for config 2); k_path_primary (pixel_ray + closest hit + color_ray of the primary
    rays, one launch) is reported beside it.  Both are bound by FP32 VALU issue, not by HBM: the ~19 MB scene is served
    from L1/L2/Infinity Cache.  Everything is recomputable from the printed raw counters: flops = 15*box + 28*tri + 21*full,
    bytes = 16*box + 32*tri + 52*full.

    ONE launch set for every per-launch figure: the frame rendered on a single internal stream (tuning streams = 1), where
    each launch has the GPU to itself -- `achieved` (HIP events around the launches, measured here), `traffic` and the
    VALU / clock figures (rocprofv3 --pmc of the same single-stream frame, profiles/pmc_latest.json)."""
    from rust_raytrace_amd import raytrace as R
    base_opts = caster.options
    solo_tune = {"streams": 1}

    def counted(v):
        c = R.HipRayCaster(seed=caster.seed, device=caster.device, options=base_opts | R.OPT_COUNTERS, tuning=solo_tune)
        return c.walk_tile_device(v, scene, tile, local.data_ptr(), stream.cuda_stream).stats

    st = counted(vp)                                                # the whole frame
    work = lambda q: (15 * q["box_tests"] + 28 * q["tri_tests"] + 21 * q["full_tests"],          # flops  
                      16 * q["box_tests"] + (4 + 28) * q["tri_tests"] + 52 * q["full_tests"])   # bytes  
    flops, alg_bytes = work(st)
    # the primary rays alone: the same frame at maxdepth 1 (identical primary rays, nothing bounces)
    vp1 = R.canonical_viewport(args.width, args.height, 1, args.spp)
    st1 = counted(vp1)
    flops_p, bytes_p = work(st1)
    flops_b, bytes_b = flops - flops_p, alg_bytes - bytes_p
    frame_ms = kernel_ms / args.steps                          # device time span of one frame in the timed region
    chip = flops / (frame_ms * 1e-3) / 1e12
    # per-launch durations, measured live with HIP events on the launch stream, kernel alone on the GPU (one stream)
    solo = R.HipRayCaster(seed=caster.seed, device=caster.device, options=base_opts, tuning=solo_tune)
    solo.walk_tile_device(vp, scene, tile, local.data_ptr(), stream.cuda_stream)
    ss = solo.walk_tile_device(vp, scene, tile, local.data_ptr(), stream.cuda_stream).stats
    split = ss["pipeline"] != 1 and ss["primary_ms"] > 0       # the primary rays have their own kernel
    n_launch_frame = max(ss["trace_launches"], 1)
    if split:
        n_dom = max(n_launch_frame - 1, 1)
        dom_ms, dom_flops, dom_bytes = ss["bounce_ms"] / n_dom, flops_b / n_dom, bytes_b / n_dom
    else:
        n_dom = n_launch_frame
        dom_ms, dom_flops, dom_bytes = ss["trace_ms"] / n_dom, flops / n_dom, alg_bytes / n_dom
    achieved = dom_flops / (dom_ms * 1e-3) / 1e12
    kernel = {"linear": "k_trace_linear"}.get(args.scene, "k_trace_oct")
    # rocprofv3 --pmc passes of the single-stream frame (a profile is a separate run: rocprofv3 cannot run inside bench.py)
    prof = None
    pj = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if os.path.exists(pj):
        try:
            prof = json.load(open(pj))
        except Exception:
            prof = None
    same_cfg = bool(prof) and prof.get("config") == {"scene": args.scene, "width": args.width, "height": args.height, "spp": args.spp, "fast": bool(args.fast)}
    pk = (prof.get("kernels", {}).get(kernel) if same_cfg else None) or None
    traffic = pk.get("fabric_bytes_per_launch") if pk else None
    other = None
    if split:
        pp = (prof.get("kernels", {}).get("k_path_primary") if same_cfg else None) or {}
        a_p = flops_p / (ss["primary_ms"] * 1e-3) / 1e12
        other = {"kernel": "k_path_primary", "launches_per_frame": 1, "avg_launch_ms": round(ss["primary_ms"], 3),
                 "achieved_TFLOPs": round(a_p, 3), "frac": round(a_p / FP32_NOFMA_PEAK_TF, 4),
                 "Grays_per_s": round(st1["rays"] / ss["primary_ms"] / 1e6, 3), "traffic": pp.get("fabric_bytes_per_launch"),
                 "note": "its flops are those of the primary rays' closest hits only; pixel_ray (Philox + 2 unit()) and color_ray of every primary ray run in the same launch and are not counted"}

    def issue(pkk, ms):  # VALU issue fraction of a kernel over its own duration: 256 CUs x 4 SIMDs, one wave64 instruction per 2 clocks
        if not pkk or not pkk.get("valu_wave_insts_per_launch"):
            return None
        per_clock = pkk["valu_wave_insts_per_launch"] / (ms * 1e-3) / (256 * 4 / 2.0)
        clk = pkk.get("effective_clock_GHz")
        return {"valu_wave_insts_per_launch": pkk["valu_wave_insts_per_launch"], "valu_issue_frac_at_2.4GHz": round(per_clock / 2.4e9, 4),
                "effective_clock_GHz": clk, "valu_issue_frac_at_held_clock": round(per_clock / (clk * 1e9), 4) if clk else None,
                "valu_lane_utilisation": pkk.get("valu_lane_utilisation"), "wait_any_frac": pkk.get("wait_any_frac"), "l2_hit_rate": pkk.get("l2_hit_rate")}

    rays_b = st["rays"] - st1["rays"]
    return {
        "bound": "valu", "kernel": kernel, "unit": "TFLOP/s",
        "achieved": round(achieved, 3), "peak": FP32_NOFMA_PEAK_TF, "frac": round(achieved / FP32_NOFMA_PEAK_TF, 4),
        "traffic": traffic,
        "launch_set": f"single-stream frame: {n_dom} x {kernel}" + (" (bounce passes) + 1 x k_path_primary" if split else "") + "; achieved, traffic and issue figures are per launch of THIS set",
        "raw": {"rays": st["rays"], "box_tests": st["box_tests"], "tri_tests": st["tri_tests"], "full_tests": st["full_tests"], "nodes": st["nodes"],
                "leaves": st["leaves"], "primary_only": {k: st1[k] for k in ("rays", "box_tests", "tri_tests", "full_tests")},
                "launches_per_frame": n_dom, "avg_launch_ms": round(dom_ms, 3), "flops_per_launch": int(dom_flops),
                "algorithmic_bytes_per_launch": int(dom_bytes), "rays_in_these_launches": int(rays_b if split else st["rays"]),
                "single_stream_frame_device_ms": round(ss["kernel_ms"], 3), "pipeline": ss["pipeline"],
                "timed_region": {"frame_device_ms": round(frame_ms, 3), "launches_per_frame": int(launches / args.steps), "avg_launch_ms_sharing_the_gpu": round(trace_ms / max(launches, 1), 3)}},
        "per_ray": {"box_tests": round(st["box_tests"] / max(st["rays"], 1), 1), "tri_tests": round(st["tri_tests"] / max(st["rays"], 1), 1),
                    "full_tests": round(st["full_tests"] / max(st["rays"], 1), 2), "flops": round(flops / max(st["rays"], 1)),
                    "algorithmic_bytes": round(alg_bytes / max(st["rays"], 1)),
                    "bounce_rays": {"flops": round(flops_b / max(rays_b, 1)), "algorithmic_bytes": round(bytes_b / max(rays_b, 1))} if split else None},
        "other_kernel": other,
        "fractions": {
            "i_algorithmic_bytes_GBs_vs_hbm_peak": {"achieved": round(dom_bytes / (dom_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                                                    "ratio": round(dom_bytes / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                                    "frame": round(alg_bytes / (frame_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                                    "note": "cache-served, NOT a bound: child boxes are implicit and the scene sits in L2/Infinity Cache, so this ratio may exceed 1"},
            "ii_fp32_TFLOPs": {"per_launch": round(achieved, 3), "chip_frame": round(chip, 3), "peak_no_fma": FP32_NOFMA_PEAK_TF,
                               "frac_no_fma_chip": round(chip / FP32_NOFMA_PEAK_TF, 4), "peak_fma": FP32_FMA_PEAK_TF,
                               "frac_fma_chip": round(chip / FP32_FMA_PEAK_TF, 4)},
            "iii_measured": ({"fabric_GBs": round(traffic / (dom_ms * 1e-3) / 1e9, 1) if traffic else None, "hbm_peak_GBs": HBM_PEAK_GBS,
                              "fabric_ratio": round(traffic / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
                              "traffic_over_algorithmic_bytes": round(traffic / dom_bytes, 4) if traffic else None,
                              kernel: issue(pk, dom_ms),
                              "k_path_primary": issue(prof.get("kernels", {}).get("k_path_primary"), ss["primary_ms"]) if split else None,
                              "source": prof.get("source"), "profiled_commit": prof.get("commit"),
                              "note": "counters from the committed single-stream profile of this workload (profiles/pmc_latest.json), per launch; "
                                      "rates against THIS run's single-stream launch durations"} if pk else None),
        },
        "note": "achieved = algorithmic FP32 flops of ONE launch of the dominant kernel (device counters of a counting pass: the frame's minus "
                "the primary rays', / launches) / its average HIP-event duration with the kernel alone on the GPU (single-stream frame; what "
                "rocprofv3 --kernel-trace of RTMI_STREAMS=1 lists, profiles/); peak = 157.3/2 TFLOP/s because a*b+c may not be contracted "
                "(bit parity).  The timed region runs three sub-tiles on their own streams, whose launches overlap: chip_frame prices a frame's "
                "flops against the frame's device time there.",
    }


def cpu_quota_cores():
    """CPU bandwidth the cgroup of this process may use, in cores (None = unlimited / unknown).  sched_getaffinity() does
    not see it: on the GPU box the affinity mask shows every core of the host while the container's quota is a fraction."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]  # cgroup v2
        if q != "max":
            return round(float(q) / float(per), 2)
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())  # cgroup v1
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and per > 0:
            return round(q / per, 2)
    except (OSError, ValueError):
        pass
    return None


def cpu_topology():
    """sockets / cores per socket / threads per core from /proc/cpuinfo (what lscpu prints)."""
    phys, cores, sib = set(), None, None
    try:
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "physical id":
                phys.add(v)
            elif k == "cpu cores" and cores is None:
                cores = int(v)
            elif k == "siblings" and sib is None:
                sib = int(v)
    except (OSError, ValueError):
        pass
    return {"sockets": len(phys) or None, "cores_per_socket": cores,
            "threads_per_core": (sib // cores) if cores and sib else None, "logical_cpus": os.cpu_count()}


def cpu_baseline_object(args, obj):
    """The oracle (C++ restatement of the reference's CPU path) with rows/pixel groups pulled from a shared counter like
    DefaultRayCaster (raytrace.rs:1179-1194).  The thread count is SWEPT on a short sample first (the affinity mask of the
    GPU box lists all host cores, its cgroup quota allows far fewer: oversubscribed threads are throttled and every one
    of them crawls), then the best count renders the whole 2048 x 2048 frame of the same scene/camera/seed at a sample
    count sized for about --cpu-seconds.  `value`/`cores` are that best configuration; the sweep, the 1-thread rate and
    the quota are printed beside it."""
    from oracle import orc
    so = orc.canonical_scene(obj)
    affinity = len(os.sched_getaffinity(0))
    quota = cpu_quota_cores()

    def run(w, h, spp, threads):
        t1 = time.perf_counter()
        _, cn = so.render(w, h, orc.canonical_viewport(w, h), args.maxdepth, spp, seed=args.seed, threads=threads)
        return cn["rays"], time.perf_counter() - t1

    r1, d1 = run(256, 256, 1, 1)  # one thread: the per-core rate nothing can exceed
    one = r1 / d1
    cand = sorted({t for t in (2, 4, 8, 12, 16, 24, 32, 64, 128, 256, affinity) if 1 < t <= affinity})
    sweep = {"1": {"Mrays_s": round(one / 1e6, 4), "krays_s_per_thread": round(one / 1e3, 1)}}
    best_t, best_rate = 1, one
    for t in cand:
        side = 512 if t <= 8 else 1024
        r, d = run(side, side, 1, t)
        rate = r / d
        sweep[str(t)] = {"Mrays_s": round(rate / 1e6, 4), "krays_s_per_thread": round(rate / t / 1e3, 1)}
        if rate > best_rate * 1.03:  # more threads only when they pay
            best_t, best_rate = t, rate
    cw = ch = 2048
    cspp = int(max(1, min(64, round(args.cpu_seconds * best_rate / (cw * ch * 1.41)))))
    rays, cdt = run(cw, ch, cspp, best_t)
    value = rays / cdt
    return {"value": round(value / 1e6, 4), "unit": "Mrays/s", "cores": best_t, "kind": "port",
            "sample": f"canonical scene, same camera and seed, the whole {cw}x{ch} frame @ {cspp} spp, depth {args.maxdepth}: "
                      f"{rays} rays in {cdt:.2f} s wall on {best_t} threads (the best of the sweep)",
            "krays_s_per_thread": round(value / best_t / 1e3, 1), "one_thread_krays_s": round(one / 1e3, 1),
            "cpu_quota_cores": quota, "affinity_cores": affinity, "topology": cpu_topology(),
            "thread_sweep": sweep,
            "cpu": cpu_model(), "compiler_flags": oracle_flags(),
            "note": "C++ restatement of raytrace_lib's CPU path (the Rust crate cannot be built here: no rustc/cargo); it omits the "
                    "reference's per-ray HashMap bookkeeping (raytrace.rs:1275-1278), so it is at least as fast as the reference. "
                    "cores = threads of the fastest configuration of thread_sweep (short 512^2/1024^2 @ 1 spp samples); "
                    "cpu_quota_cores = the cgroup CPU quota of this container (null = none), which bounds what any thread count can deliver"}


if __name__ == "__main__":
    main()
