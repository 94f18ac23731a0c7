#!/usr/bin/env python3
"""Headline benchmark: Msamples/s (pixels x spp / s) of the per-pixel render loop on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config C2] [--spp S]

A "step" is one pass of the hot path over one batch: `spp` samples per pixel of the configured frame
(default BASELINE.json configs[1]: Cornell box, 1280x720, 8 bounces, 256 spp), i.e. 256 RenderFrame()
calls as one trace_kernel + one accumulate_kernel launch, followed — for N > 1 — by the exchange step (RCCL gather of
every rank's owned tiles to rank 0).  The scene (replicated), the accumulator and the primary-ray
table are resident in HBM before the timed region.  N > 1: one process per GPU (torchrun), the frame
is tile-split across ranks (16x16 tiles, round-robin), total work fixed -> "strong" scaling.

Prints ONE JSON line on rank 0 (see the driver contract) including
  roofline      algorithmic bytes per launch / measured kernel time vs the 8 TB/s HBM peak
  cpu_baseline  the reference's own OpenMP CPU path (oracle/_ref, built from /root/reference by
                __graft_entry__.build()) — or the oracle port if that .so is absent — timed on this
                box's host cores on a bounded sample of the same workload (N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
VALU_PEAK_TFLOPS = 157.3     # FP32 vector: 256 CUs x 128 lanes x 2 flops (FMA) x 2.4 GHz

# device record sizes (pbrpathtracer_amd/csrc/ptk_device.h) -> algorithmic bytes, DESIGN.md §Roofline
BYTES_NODE = 64              # BVH2 node record: two child boxes + two child links
BYTES_TRI = 48               # triangle intersection record: v0, e1, e2, ids
BYTES_SHADE = 112            # shading record of the accepted hit
BYTES_MATERIAL = 96
BYTES_LIGHT = 64             # light record per shadow ray
BYTES_TEXEL = 4
BYTES_SAMPLE_OUT = 16        # one float4 radiance sample stored per path (trace -> accumulate kernel)
BYTES_PRIMARY = 16           # primary direction, fetched once per work item (chunk of samples)


def algorithmic_bytes_per_sample(stats: dict, chunk: int, flat: bool) -> float:
    """trace_kernel: bytes one sample (one pixel x one spp) needs, from the kernel's own counts.
    In FLAT mode (scenes of <= 16 triangles) a triangle record is fetched once per WAVE with a scalar
    load and broadcast, so a lane's test accounts for 48/64 bytes."""
    s = float(stats["samples"])
    tri_bytes = BYTES_TRI / 64.0 if flat else BYTES_TRI
    return (BYTES_SAMPLE_OUT + BYTES_PRIMARY / float(chunk)
            + stats["node_visits"] / s * BYTES_NODE
            + stats["tri_tests"] / s * tri_bytes
            + stats["hits_shaded"] / s * (BYTES_SHADE + BYTES_MATERIAL)
            + stats["shadow_rays"] / s * BYTES_LIGHT
            + stats["tex_fetches"] / s * BYTES_TEXEL)


def cpu_baseline(scene, name: str, budget_s: float = 12.0):
    """Time the CPU path on a bounded sample of the same workload (whole frames of 1 spp)."""
    from oracle import ref_binding
    w, h = scene.width, scene.height
    if ref_binding.available():
        ref = ref_binding.Ref()
        ref.load_scene(scene, exact_pinhole=True)
        ref.lib.ref_seed(12345)
        import ctypes
        omp = ctypes.CDLL("libgomp.so.1")
        maxt = omp.omp_get_max_threads()
        workers = maxt - 3 if maxt > 2 else (maxt - 2 if maxt > 1 else maxt - 1)   # pathtracer.cpp:768-774
        frames, sec = 0, 0.0
        while sec < budget_s and frames < 64:
            sec += ref.render(1, 0)
            frames += 1
        return {"value": round(w * h * frames / sec / 1e6, 4), "unit": "Msamples/s", "cores": int(workers), "kind": "reference",
                "sample": f"{frames} RenderFrame() calls (1 spp each) of {name} at {w}x{h}, depth {scene.trace_depth}; "
                          f"reference sources compiled -O2 -fopenmp, workers = omp_get_max_threads()-3, one shared mt19937"}
    # fall back to the oracle port (our own C restatement), all cores, per-path counter RNG
    from oracle import oracle_binding as OB
    from pbrpathtracer_amd.pathtracer import PathTracer, camera_from_scene
    pt = PathTracer()
    pts = os.path.join(tempfile.mkdtemp(prefix="bench_cpu_"), "s.pts")
    from pbrpathtracer_amd import scenes as S
    S.write_pts(pts, scene)
    pt.LoadSceneFile(pts)
    arrays = pt.StagedScene()
    cam = camera_from_scene(scene)
    if scene.pinhole:
        cam["aperture"] = 0.0
    o = OB.Oracle(arrays)
    ocam = OB.make_camera(cam["pos"], cam["dir"], cam["up"], cam["focal"], cam["fovy"], cam["focal_dist"], cam["aperture"])
    cores = os.cpu_count() or 1
    total = np.zeros((h, w, 3), np.float32)
    frames, sec = 0, 0.0
    while sec < budget_s and frames < 64:
        t0 = time.time()
        o.render(ocam, w, h, scene.trace_depth, frames, 1, 1, total=total, threads=cores, want_rgb8=False)
        sec += time.time() - t0
        frames += 1
    return {"value": round(w * h * frames / sec / 1e6, 4), "unit": "Msamples/s", "cores": int(cores), "kind": "port",
            "sample": f"{frames} frames (1 spp each) of {name} at {w}x{h}, depth {scene.trace_depth}; oracle/pt_oracle.c, OpenMP all cores"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="C2", choices=["C1", "C2", "C3", "C4", "C5"])
    ap.add_argument("--spp", type=int, default=0, help="samples per pixel per step (default: the config's spp)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--force-exchange", action="store_true", help="run the exchange step even at N=1 (rehearsal of the N>1 path)")
    ap.add_argument("--opts", default=os.environ.get("PTK_OPTS", ""), help="ptk_set_option pairs, k=v[,k=v...] (tuning experiments)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = REHEARSAL of the N>1 control flow where ranks must share one GPU (RCCL refuses that): the exchange "
                         "then goes through host copies; the numbers mean nothing")
    ap.add_argument("--share-of", type=int, default=0, help="rehearsal: render only rank 0's tiles of an N-rank split on this one GPU")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print("bench.py: --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)", file=sys.stderr)
            sys.exit(2)

    import torch
    import torch.distributed as dist
    from pbrpathtracer_amd import scenes as S
    from pbrpathtracer_amd.pathtracer import PathTracer

    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the render path has no CPU fallback", file=sys.stderr)
        sys.exit(2)
    rehearsal = args.backend == "gloo"
    if rehearsal:
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    if world > 1 and rehearsal:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    elif world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        try:
            # RCCL's own stream at high priority: the collective's workgroups get the wave slots that retiring
            # trace_kernel waves free (the trace kernel of the next batch is already running, see step())
            opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank), pg_options=opts)
        except (AttributeError, TypeError):
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    # ---- scene: synthesised through the reference's own formats (.obj + .pts) ---------------------
    tmp = tempfile.mkdtemp(prefix=f"bench_{args.config}_r{rank}_")
    t0 = time.time()
    pts, scene, cfg_spp = S.build_config(args.config, tmp)
    spp = args.spp if args.spp > 0 else cfg_spp
    t_gen = time.time() - t0
    pt = PathTracer(device=local_rank)
    t0 = time.time()
    pt.LoadSceneFile(pts)                       # LoadObject/SetMaterial/.../BuildBVH/SetCamera/SetResolution
    t_load = time.time() - t0
    if scene.pinhole:
        pt.SetCameraAperture(0.0)               # exact pinhole through the API; the .pts carries F = 1e9
    pt.SetSeed(args.seed)
    pt.SetTile(rank, world)
    if args.share_of > 1 and world == 1:
        pt.SetTile(0, args.share_of)
    W, H = pt.GetResolution()
    D = pt.GetTraceDepth()
    pt.RenderFrames(1)                          # creates the frame buffers, primary-ray table; 1 spp
    if pt.LastError():
        print("bench.py: " + pt.LastError(), file=sys.stderr)
        sys.exit(3)
    ctx = pt.context()
    for kv in args.opts.split(","):
        if "=" in kv:
            k, v = kv.split("=")
            ctx.set_option(k, float(v))
    # the kernel renders into a torch-owned accumulator on torch's current stream, so the exchange
    # step (torch.distributed -> RCCL) is ordered behind the render without host synchronisation
    accum = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.bind_accum(accum.data_ptr())
    ctx.reset()

    from pbrpathtracer_amd.distributed import AccumulatorExchange
    host_accum = torch.zeros(H * W * 3, dtype=torch.float32) if rehearsal else None
    exchange = AccumulatorExchange(host_accum if rehearsal else accum, dst=0, width=W, height=H) if (world > 1 or args.force_exchange) else None

    def step(first):
        ctx.render(first, spp, args.seed)
        if exchange is not None and rehearsal:
            host_accum.copy_(accum)                 # (synchronises; rehearsal only)
            exchange.start()
        elif exchange is not None:
            # snapshot + RCCL reduce on a side stream: the collective of step k overlaps the trace
            # kernel of step k+1 (tiles of other ranks are exact zeros, so the sum is a gather)
            exchange.start()

    def fence():
        if exchange is not None:
            exchange.wait()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    first = 0
    if exchange is not None:
        # one untimed exchange up front: if the packed gather is not available in this RCCL / torch build every rank
        # falls back to the sum-reduce form together (same result, 8x the payload)
        try:
            exchange.start(); exchange.wait(); torch.cuda.synchronize()
        except Exception as e:                              # pragma: no cover - depends on the installed collectives
            if rank == 0:
                print(f"bench.py: packed exchange unavailable ({type(e).__name__}: {e}); using reduce", file=sys.stderr)
            exchange = AccumulatorExchange(host_accum if rehearsal else accum, dst=0, mode="reduce")
    for _ in range(args.warmup):
        step(first); first += spp
    fence()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(first); first += spp
        if world == 1:
            pass
    fence()
    elapsed = time.perf_counter() - t0
    if rehearsal and world > 1 and exchange is not None:
        # property check of the exchange: the gathered image holds exactly what the ranks hold together
        part = torch.tensor([float(accum.double().sum().item())], dtype=torch.float64)
        dist.all_reduce(part, op=dist.ReduceOp.SUM)
        if rank == 0:
            got = float(exchange.result.double().sum().item())
            ok = abs(got - float(part.item())) <= 1e-9 * max(1.0, abs(got))
            print(f"bench.py rehearsal: gathered-image checksum {'OK' if ok else 'MISMATCH'} ({got:.6f} vs {float(part.item()):.6f}), "
                  f"exchange mode {exchange.mode}", file=sys.stderr)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # kernel time per launch, measured live with HIP events recorded on the kernels' own stream
    # around each launch (trace_kernel and accumulate_kernel separately), over a few more steps
    # (launch by launch, un-overlapped: in the timed loop above the trace kernel of a batch starts while the previous
    # one's last few waves - its longest paths - are still finishing, which is what `value` measures; a per-launch
    # duration only means something in isolation)
    ev_ms, acc_ms = [], []
    ctx.set_option("overlap", 0)
    for _ in range(min(args.steps, 5)):
        ctx.render(first, spp, args.seed); first += spp
        t_ms, a_ms = ctx.last_kernel_ms()
        ev_ms.append(t_ms); acc_ms.append(a_ms)
    ctx.set_option("overlap", 1)
    for kv in args.opts.split(","):
        if kv.startswith("overlap="):
            ctx.set_option("overlap", float(kv.split("=")[1]))
    fence()

    total_samples = float(W) * H * spp * args.steps
    value = total_samples / elapsed / 1e6

    out = None
    if rank == 0:
        # algorithmic bytes from the kernel's own traversal counts (untimed counters-enabled variant)
        ntri_for_chunk = ctx.bvh_info()[2]
        # ptk's automatic samples per work item (ptk_api.hip run_passes)
        from pbrpathtracer_amd.distributed import owned_tile_count
        chunk = 8 if spp * owned_tile_count(W, H, rank, world) * 4.0 / 8.0 >= 49152.0 else 4
        chunk = min(chunk, spp)
        # counters over 64 spp of the whole frame, with the work-item size the timed launches used
        ctx.set_tile(0, 1)
        ctx.set_option("chunk", chunk)
        stats = ctx.collect_stats(0, min(spp, 64), args.seed)         # long enough for the persistent waves' steady state
        ctx.set_option("chunk", 0)
        ctx.set_tile(rank, world)
        flat = ntri_for_chunk <= 16
        bps = algorithmic_bytes_per_sample(stats, chunk, flat)
        launch_samples = float(W) * H * spp / world
        avg_ms = float(np.mean(ev_ms))
        achieved = bps * launch_samples / (avg_ms * 1e-3) / 1e9
        s = float(stats["samples"])
        # SURVEY.md §8(d4)'s own per-unit figures, for cross-checking: a visited node here holds BOTH child
        # boxes = two of its 32-B node records / two of its 30-flop box tests
        d4_bytes = (27.0 + stats["node_visits"] / s * 2 * 32 + stats["tri_tests"] / s * 36
                    + stats["hits_shaded"] / s * (104 + 48) + stats["tex_fetches"] / s * 4)
        d4_flops = stats["node_visits"] / s * 2 * 30 + stats["tri_tests"] / s * 50 + stats["hits_shaded"] / s * 250
        valu_tflops = d4_flops * launch_samples / (avg_ms * 1e-3) / 1e12
        roofline = {
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
            "kernel": "trace_kernel<false>", "kernel_ms": round(avg_ms, 4),
            "accumulate_kernel_ms": round(float(np.mean(acc_ms)), 4),
            "algorithmic_bytes_per_sample": round(bps, 1),
            "per_sample": {"rays": round(stats["rays"] / s, 3), "shadow_rays": round(stats["shadow_rays"] / s, 3),
                           "node_visits": round(stats["node_visits"] / s, 2), "tri_tests": round(stats["tri_tests"] / s, 2),
                           "hits_shaded": round(stats["hits_shaded"] / s, 3), "tex_fetches": round(stats["tex_fetches"] / s, 3)},
            "kernel_variant": "FLAT (no BVH walk, scalar triangle loads)" if flat else "BVH2 walk",
            "simd_lane_utilisation": {"walk": round(stats["walk_lane_iters"] / max(1, stats["walk_wave_iters"]) / 64.0, 3),
                                      "shade": round(stats["shade_lanes"] / max(1, stats["shade_wave_execs"]) / 64.0, 3),
                                      "camera": round(stats["gen_lanes"] / max(1, stats["gen_wave_execs"]) / 64.0, 3)},
            "survey_d4": {"bytes_per_sample": round(d4_bytes, 1),
                          "achieved_GBps": round(d4_bytes * launch_samples / (avg_ms * 1e-3) / 1e9, 1),
                          "frac": round(d4_bytes * launch_samples / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
            "valu": {"achieved": round(valu_tflops, 2), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(valu_tflops / VALU_PEAK_TFLOPS, 4), "flops_per_sample": round(d4_flops, 1),
                     "note": "algorithmic flops of SURVEY §8(d4); the peak counts an FMA as 2 flops, which the "
                             "parity contract (-ffp-contract=off) forbids, so 0.5 is the ceiling of this fraction"},
            "note": ("the scene records are cache-resident or cache-friendly: most algorithmic bytes are served by the scalar "
                     "cache / L1 / L2 / MALL (so `frac` can exceed 1), HBM sees `traffic`; the kernel is VALU-issue- and "
                     "divergence-bound (DESIGN.md §5)"),
        }
        traffic_file = os.path.join(ROOT, "profiles", f"traffic_{args.config}.json")
        if os.path.exists(traffic_file):
            try:
                tr = json.load(open(traffic_file))
                if tr.get("spp") == spp and tr.get("n_gpus", 1) == world:
                    roofline["traffic"] = tr["hbm_bytes_per_launch"]
                    roofline["traffic_source"] = tr.get("source")
            except Exception:
                pass
        nodes, depth, ntri = ctx.bvh_info()
        out = {
            "metric": "Msamples/s (pixels*spp/s)", "value": round(value, 2), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": {"C1": "Cornell box (12 tris), 512x512, 4 bounces, 16 spp",
                                    "C2": "Cornell box (12 tris, no textures), 1280x720, 8 bounces, 256 spp",
                                    "C3": "Textured PBR spheres + DOF, 1280x720, 8 bounces, 512 spp",
                                    "C4": "bunny stand-in (~70k tris), 1920x1080, 8 bounces, 256 spp",
                                    "C5": "1M-triangle height field, 1920x1080, 12 bounces, 1024 spp"}[args.config],
                       "name": args.config, "width": W, "height": H, "max_depth": D, "spp_per_step": spp,
                       "triangles": ntri, "bvh_nodes": nodes, "bvh_depth": depth,
                       "parallelism": f"tile-split x{world}" if world > 1 else "single GPU",
                       "exchange": (("RCCL gather of each rank's owned tiles (packed, 1/N of the float accumulator)" if exchange.mode == "gather"
                                     else "RCCL sum-reduce of the float accumulator") + " to rank 0, once per step, overlapped with "
                                    "the next step's trace kernel") if world > 1 else "none"},
            "roofline": roofline,
            "host": {"scene_gen_s": round(t_gen, 3), "scene_load_bvh_upload_s": round(t_load, 3)},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                # the CPU path is timed on the reference's own CPU-runnable shape of the same scene
                base = cpu_baseline(scene, args.config)
                out["cpu_baseline"] = base
                out["gpu_over_cpu"] = round(value / base["value"], 1) if base["value"] > 0 else None
            except Exception as e:  # the baseline is reported, never required
                out["cpu_baseline"] = {"value": None, "unit": "Msamples/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
