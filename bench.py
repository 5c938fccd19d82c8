#!/usr/bin/env python3
"""Benchmark of the hot path: Mrays/s and spp/s on the knob/Disney scene at 768x768.

One "step" renders ``--strata-per-step`` (256) full-frame stratified passes (every pixel, one path
each, depth 50) on each GPU; ranks own disjoint stratum ranges (no data-path collective)
and the float accumulators are summed onto rank 0 with one reduce at the end of the timed
region.  Rank 0 prints one JSON line.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic bytes per tested item (what one ray must read at minimum)
B_NODE, B_TRI, B_SPHERE, B_QUAD, B_RAY_IO = 32, 36, 16, 64, 32 + 16
HBM_PEAK_GBS = 8000.0
# wave64 vector instructions the chip can issue per second: 256 CUs x 4 SIMDs x 2.4 GHz, one every 2 cycles
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 2
# one scalar unit per CU, shared by its four SIMDs: a scalar instruction per 1.15 cycles per CU with eight waves per SIMD
# issuing nothing else (tools/micro/valu_rate.hip, profiles/r03_valu_issue_rates.txt)
SALU_ISSUE_PEAK = 256 * 2.4e9 / 1.15


def csrc_sha256() -> str:
    """Identity of the kernel sources: counter files recorded for other sources are not used."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "pathtracer_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".h", ".hip")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()


def config_tag(args) -> str:
    """Names the workload a counter file belongs to: profiles/r03_counters_<tag>.json (tools/pmc_counters.py)."""
    tag = f"{args.scene}_{args.width}"
    if args.extensions:
        tag += "_ext"
    if args.bsdf_probability != 1.0:
        tag += f"_p{args.bsdf_probability:g}"
    if args.integrator != "reference":
        tag += f"_{args.integrator}"
    return tag


def cpu_baseline(src, budget_s: float = 12.0, bsdf_probability: float = 1.0):
    """The CPU restatement of the reference renderer on this host's cores, bounded sample."""
    from oracle.pyoracle import OracleScene
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)
    o = OracleScene(src)
    if bsdf_probability != 1.0:
        o.set_bsdf_probability(bsdf_probability)
    t0 = time.perf_counter()
    rays = 0
    strata = 0
    while True:
        r = o.render(strata, strata + 1, seed=1, threads=threads, iterative=False, tonemap_each_pass=True, want_rgb8=True)
        rays += r["rays"]
        strata += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or strata >= 4096:
            break
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "spp_per_s": strata / dt,
            "sample": f"{strata} full-frame stratified passes of the same scene ({o.width}x{o.height}, depth {src.camera.max_depth}), "
                      f"recursive reference-order renderer, {dt:.1f} s wall"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=768)
    ap.add_argument("--strata-per-step", type=int, default=256)
    ap.add_argument("--scene", default="knob", choices=["knob", "cornell", "dragon_box", "interior", "next_week_like"])
    ap.add_argument("--pipes", type=int, default=0, help="batches in flight per GPU (0: library default)")
    ap.add_argument("--refill-below", type=int, default=0)
    ap.add_argument("--leaf-batch", type=int, default=0)
    ap.add_argument("--blocks-per-cu", type=int, default=0)
    ap.add_argument("--tail-below", type=int, default=-1)
    ap.add_argument("--rolling", type=int, default=-1,
                    help="paths started per launch of the rolling schedule (0: one batch at a time per pipe; -1: library default)")
    ap.add_argument("--live-factor", type=float, default=0.0, help="rays in flight a pipe has room for, as a multiple of the paths started per launch (0: library default)")
    ap.add_argument("--integrator", default="reference", choices=["reference", "nee"],
                    help="nee: next-event estimation with shadow rays (not in the reference); no counted pass, so no roofline figure")
    ap.add_argument("--bsdf-probability", type=float, default=1.0,
                    help="BSDF_SAMPLE_PROBABILITY of the reference's mixture sampler (src/camera.rs:317); below 1 some rays are aimed at lights")
    ap.add_argument("--lights", type=int, default=64, help="interior: number of emissive quads")
    ap.add_argument("--extensions", action="store_true", help="next_week_like with its noise texture, media and moving sphere")
    ap.add_argument("--max-depth", type=int, default=0, help="knob: override the scene's depth of 50 (experiments only)")
    ap.add_argument("--packet", type=int, default=-1, help="PT_OPT_PACKET (-1: library default)")
    ap.add_argument("--batch-ticks", type=int, default=-1, help="PT_OPT_BATCH_TICKS (-1: library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")

    import torch
    import torch.distributed as dist

    from pathtracer_amd import scenes
    from pathtracer_amd.core import (PT_OPT_BATCH_TICKS, PT_OPT_BLOCKS_PER_CU, PT_OPT_BSDF_SAMPLE_PROBABILITY, PT_OPT_COUNTERS, PT_OPT_TAIL_BELOW, PT_OPT_ROLLING, PT_OPT_INTEGRATOR, PT_OPT_LEAF_BATCH, PT_OPT_LIVE_FACTOR, PT_OPT_MAX_PATHS, PT_OPT_PACKET, PT_OPT_PIPES, PT_OPT_PROFILE, PT_OPT_REFILL_BELOW,
                                     HostScene, Renderer)
    from pathtracer_amd.sharding import reduce_accumulator, stratum_ranges

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    builders = {"knob": lambda: scenes.knob(width=args.width, **({"max_depth": args.max_depth} if args.max_depth else {})), "cornell": lambda: scenes.cornell(width=args.width),
                "dragon_box": lambda: scenes.dragon_box(width=args.width), "interior": lambda: scenes.interior(width=args.width, n_lights=args.lights),
                "next_week_like": lambda: scenes.next_week_like(width=args.width, extensions=args.extensions)}
    src = builders[args.scene]()
    host = HostScene(src)
    r = Renderer(host, device=local_rank)
    S = args.strata_per_step
    npix = r.npix
    if args.pipes:
        r.set_option(PT_OPT_PIPES, args.pipes)
    if args.refill_below:
        r.set_option(PT_OPT_REFILL_BELOW, args.refill_below)
    if args.leaf_batch:
        r.set_option(PT_OPT_LEAF_BATCH, args.leaf_batch)
    if args.blocks_per_cu:
        r.set_option(PT_OPT_BLOCKS_PER_CU, args.blocks_per_cu)
    if args.tail_below >= 0:
        r.set_option(PT_OPT_TAIL_BELOW, args.tail_below)
    if args.rolling >= 0:
        if args.rolling > 32_000_000:  # beyond the library's default cap on paths per batch
            r.set_option(PT_OPT_MAX_PATHS, args.rolling)
        r.set_option(PT_OPT_ROLLING, args.rolling)
    if args.live_factor:
        r.set_option(PT_OPT_LIVE_FACTOR, args.live_factor)
    if args.packet >= 0:
        r.set_option(PT_OPT_PACKET, args.packet)
    if args.batch_ticks >= 0:
        r.set_option(PT_OPT_BATCH_TICKS, args.batch_ticks)
    if args.integrator == "nee":
        r.set_option(PT_OPT_INTEGRATOR, 1)
    if args.bsdf_probability != 1.0:
        r.set_option(PT_OPT_BSDF_SAMPLE_PROBABILITY, args.bsdf_probability)
    total_strata = r.sqrt_spp * r.sqrt_spp
    lo, hi = stratum_ranges(total_strata, world)[rank]
    S = min(S, hi - lo)
    if S <= 0:
        raise SystemExit(f"rank {rank}: empty stratum shard")
    span = (hi - lo) - S + 1  # first strata a step may start at; long runs wrap around inside the shard

    def step(k):
        # a run longer than the shard starts over inside it with another seed: no (seed, pixel, stratum) sample is drawn twice
        b = lo + (k * S) % span
        r.render(b, b + S, seed=1 + (k * S) // span)

    for k in range(args.warmup):
        step(k)
    r.sync()
    r.reset_accum()
    r.stats_reset()
    r.set_option(PT_OPT_PROFILE, 1)

    accum_dev = torch.empty(npix * 3, dtype=torch.float32, device="cuda")

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
    r.copy_accum_to_device(accum_dev.data_ptr())
    reduce_accumulator(accum_dev, dst=0)
    barrier()
    dt = time.perf_counter() - t0
    st = r.stats()
    r.set_option(PT_OPT_PROFILE, 0)

    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    tot = torch.tensor([float(st["rays"]), float(st["paths"])], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    dt = float(tmax.item())
    rays_all, paths_all = float(tot[0].item()), float(tot[1].item())

    if rank == 0:
        if world > 1:
            r.set_accum_from_device(accum_dev.data_ptr(), args.steps * S * world)
        img = r.read_rgb8()
        # traversal work per ray on this rank's last step, counted by the device in the reference's order; once more with
        # the camera's depth set to 1, which leaves the camera rays alone: the difference is the rays of later bounces
        r.stats_reset()
        if args.integrator == "nee":  # the counters exist for the reference's estimator only: its rays stand in
            r.set_option(PT_OPT_INTEGRATOR, 0)
        r.set_option(PT_OPT_COUNTERS, 1)
        r.reset_accum()
        zero = {"rays": 0, "node_tests": 0, "tri_tests": 0, "sphere_tests": 0, "quad_tests": 0}

        def counted():
            try:
                r.stats_reset()
                step(args.warmup)
                return r.stats()
            except Exception as e:
                print(f"no counted pass: {e}", file=sys.stderr)
                return dict(zero)

        def bytes_per_ray(c):
            return (B_NODE * c["node_tests"] + B_TRI * c["tri_tests"] + B_SPHERE * c["sphere_tests"] + B_QUAD * c["quad_tests"]) / max(1, c["rays"]) + B_RAY_IO

        sc = counted()
        import ctypes as C
        cam_full = type(host.camera).from_buffer_copy(host.camera)
        cam_one = type(host.camera).from_buffer_copy(host.camera)
        cam_one.max_depth = 1
        r._check(r.L.pt_set_camera(r.ctx, C.byref(cam_one)), "pt_set_camera")
        sc_cam = counted() if sc["rays"] else dict(zero)
        r._check(r.L.pt_set_camera(r.ctx, C.byref(cam_full)), "pt_set_camera")
        r.set_option(PT_OPT_COUNTERS, 0)
        sc_bounce = {k: sc[k] - sc_cam[k] for k in zero}
        rays_c = max(1, sc["rays"])
        b_ray, b_ray_cam, b_ray_bounce = bytes_per_ray(sc), bytes_per_ray(sc_cam), bytes_per_ray(sc_bounce)
        # The closest-hit search of a step on this rank over the timed region, bracketed by HIP events on its stream:
        # averages over the launches that had rays.
        launches = max(1, st["extend_launches"])
        ext_ms_per_launch = st["extend_ms"] / launches
        rays_per_launch = st["rays"] / launches
        busy_s = st["extend_busy_ms"] * 1e-3
        union_busy = st["rays"] * b_ray / busy_s / 1e9 if busy_s > 0 else 0.0
        launch_s = ext_ms_per_launch * 1e-3
        achieved = rays_per_launch * b_ray / launch_s / 1e9 if launch_s > 0 else 0.0
        # each of the two kernels with its own rays, its own bytes per ray and its own events (they run side by side,
        # so each one's time is the time it took beside the other)
        per_kernel = {}
        if st["packet_launches"]:
            kernels = (("k_extend_packet", st["paths"], st["packet_ms"], st["packet_launches"], b_ray_cam, "the paths that start with a step (camera rays), a wave at a time"),
                       ("k_extend_walk", st["rays"] - st["paths"] - st["shadow_rays"], st["walk_ms"], st["walk_launches"], b_ray_bounce, "the rays queued by earlier steps (later bounces), drawn by persistent waves"))
        else:
            kernels = (("k_extend_walk", st["rays"] - st["shadow_rays"], st["walk_ms"], st["walk_launches"], b_ray, "all rays of a step, drawn by persistent waves (the camera rays have no kernel of their own in this configuration)"),)
        for kname, krays, kms, kn, kb, what in kernels:
            if kms > 0 and kn:
                a = krays * kb / (kms * 1e-3) / 1e9
                per_kernel[kname] = {"what": what, "rays": krays, "launches": kn, "ms_per_launch": kms / kn, "rays_per_launch": krays / kn, "Grays_per_s": krays / kms / 1e6,
                                     "bytes_per_ray": kb, "achieved": a, "frac": a / HBM_PEAK_GBS, "saturated": a > HBM_PEAK_GBS}
        # counter-based figures: per-ray numbers of rocprofv3 --pmc passes over this very command (tools/pmc_passes.sh,
        # tools/pmc_counters.py), recorded in a file per configuration; used only when that file was taken with the
        # kernel sources this library was built from
        traffic = hbm_counter_frac = None
        binding = None
        counters_from = None
        cfile = os.path.join(ROOT, "profiles", f"r03_counters_{config_tag(args)}.json")
        if os.path.exists(cfile) and launch_s > 0:
            with open(cfile) as f:
                pc = json.load(f)
            if pc.get("csrc_sha256") == csrc_sha256():
                traffic = pc["hbm_bytes_per_ray"] * rays_per_launch
                hbm_counter_frac = traffic / launch_s / (HBM_PEAK_GBS * 1e9)
                valu_issue_frac = pc["valu_per_ray"] * rays_per_launch / launch_s / VALU_ISSUE_PEAK
                binding = {"bound": "valu_issue", "frac": valu_issue_frac, "lane_util": pc["lane_util"], "useful_lane_frac": valu_issue_frac * pc["lane_util"],
                           "peak": VALU_ISSUE_PEAK, "unit": "wave64 vector instructions/s (256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles)",
                           "valu_per_ray": pc["valu_per_ray"], "salu_per_ray": pc["salu_per_ray"], "waves_parked": pc["wait_share_of_wave_cycles"]}
                # what a vector instruction of these kernels costs a SIMD on average (2 cycles for adds, multiplies, moves;
                # 4 for min / max / compares / shifts; 8 for reciprocals: measured, tools/micro/valu_rate.hip) from the static
                # mix of each kernel (tools/issue_mix.py), when that file was made from these sources
                mix = {}
                mfile = os.path.join(ROOT, "profiles", "r03_issue_mix.json")
                if os.path.exists(mfile):
                    with open(mfile) as f:
                        mj = json.load(f)
                    if mj.get("csrc_sha256") == csrc_sha256():
                        mix = mj["kernels"]
                inst = "<instances>" if host.desc.n_instances else ""
                cyc_sum = valu_sum = 0.0
                for kname in per_kernel:
                    k = pc.get("per_kernel", {}).get(kname)
                    if k:
                        pk = per_kernel[kname]
                        per_s = pk["rays_per_launch"] / (pk["ms_per_launch"] * 1e-3)
                        f_issue = k["valu_per_ray"] * per_s / VALU_ISSUE_PEAK
                        pk["binding"] = {"bound": "valu_issue", "frac": f_issue, "lane_util": k["lane_util"], "useful_lane_frac": f_issue * k["lane_util"],
                                         "valu_per_ray": k["valu_per_ray"], "salu_per_ray": k["salu_per_ray"],
                                         "scalar_unit_frac": k["salu_per_ray"] * per_s / SALU_ISSUE_PEAK}
                        m = mix.get(kname + inst) or mix.get(kname)
                        if m:
                            pk["binding"]["cycles_per_vector_instruction"] = m["cycles_per_vector_instruction"]
                            pk["binding"]["vector_pipe_busy"] = f_issue * m["cycles_per_vector_instruction"] / 2.0
                            cyc_sum += m["cycles_per_vector_instruction"] * k["valu_per_ray"] * pk["rays"]
                            valu_sum += k["valu_per_ray"] * pk["rays"]
                        pk["traffic"] = k["hbm_bytes_per_ray"] * pk["rays_per_launch"]
                        pk["hbm_counter_frac"] = pk["traffic"] / (pk["ms_per_launch"] * 1e-3) / (HBM_PEAK_GBS * 1e9)
                binding["scalar_unit_frac"] = pc["salu_per_ray"] * rays_per_launch / launch_s / SALU_ISSUE_PEAK
                if valu_sum > 0:
                    binding["cycles_per_vector_instruction"] = cyc_sum / valu_sum
                    binding["vector_pipe_busy"] = valu_issue_frac * binding["cycles_per_vector_instruction"] / 2.0
                    binding["note"] = ("frac prices every vector instruction at 2 cycles per SIMD; vector_pipe_busy with what the kernels' own mix costs "
                                       "(profiles/r03_issue_mix.json); scalar_unit_frac is the CU's shared scalar unit")
                counters_from = f"profiles/{os.path.basename(cfile)}: rocprofv3 --pmc passes of this command with these kernel sources (per-ray figures x this run's rays per launch / this run's launch time)"
            else:
                counters_from = f"profiles/{os.path.basename(cfile)} was recorded with other kernel sources: not used"
        out = {
            "metric": "Mrays/s & spp/s, mitsuba knob 768^2 Disney" if args.scene == "knob" else f"Mrays/s & spp/s, {args.scene}",
            "value": rays_all / dt / 1e6,
            "unit": "Mrays/s",
            "spp_per_s": args.steps * S * world / dt,
            "paths_per_s": paths_all / dt,
            "rays": rays_all,
            "paths": paths_all,
            "shadow_rays": st["shadow_rays"],
            "mean_path_length": rays_all / max(1.0, paths_all),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{src.name} scene {r.width}x{r.height}, depth {src.camera.max_depth}, {host.desc.n_tris} triangles, "
                                   f"{host.desc.n_spheres} spheres, {host.desc.n_quads} quads, {host.desc.n_instances} instances"
                                   + (" (procedural stand-in for mitsuba_smooth.obj, Disney + Lambertian + emitter)" if args.scene == "knob" else "")
                                   + f", {S} full-frame strata per step per GPU",
                       "strata_per_step": S, "parallelism": f"spp-sharded x{world}, one reduce",
                       "integrator": args.integrator, "bsdf_sample_probability": args.bsdf_probability,
                       "schedule": (f"rolling, up to {args.rolling} paths started per launch" if args.rolling > 0 else
                                    "library default (rolling, one pipe, up to 32 M paths started per launch)" if args.rolling < 0 else "one batch at a time per pipe")},
            "roofline": {"bound": "hbm", "kernel": "k_extend_walk + k_extend_packet (one step's closest-hit search: queued rays and new paths, side by side)", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "saturated": achieved > HBM_PEAK_GBS,
                         "note": "algorithmic bytes (what a ray would read with no cache) over measured launch time; above 1 the tree is served by the caches and this yardstick is saturated: `binding` says what bounds the kernel",
                         "traffic": traffic, "hbm_counter_frac": hbm_counter_frac, "binding": binding, "counters_from": counters_from,
                         "per_kernel": per_kernel,
                         "bytes_per_ray": b_ray, "bytes_per_camera_ray": b_ray_cam, "bytes_per_bounce_ray": b_ray_bounce,
                         "rays_per_launch": rays_per_launch, "ms_per_launch": ext_ms_per_launch,
                         "launches": st["extend_launches"], "busy_ms": st["extend_busy_ms"],
                         "achieved_over_union_of_launches": union_busy,
                         "launches_in_flight": st["extend_ms"] / st["extend_busy_ms"] if st["extend_busy_ms"] > 0 else 0.0,
                         "node_tests_per_ray": sc["node_tests"] / rays_c, "tri_tests_per_ray": sc["tri_tests"] / rays_c},
            "kernel_ms": {"extend": st["extend_ms"], "shade": st["shade_ms"], "other": st["other_ms"]},
            "device_bytes": st["device_bytes"],
            "csrc_sha256": csrc_sha256(),
            "host_wait_ms": st["host_wait_ms"],
            "image_mean": float(img.mean()),
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(src, args.cpu_budget, args.bsdf_probability)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
