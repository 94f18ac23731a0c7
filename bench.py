#!/usr/bin/env python3
"""Headline benchmark: Msamples/s (pixels x spp / s) of the per-pixel render loop on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config C2] [--spp S]

A "step" is one pass of the hot path over one batch: `spp` samples per pixel of the configured frame (default
BASELINE.json configs[1]: Cornell box, 1280x720, 8 bounces, 256 spp), i.e. 256 RenderFrame() calls as one trace_kernel +
one accumulate_kernel launch.  The scene (replicated), the accumulator and the primary-ray table are resident in HBM
before the timed region.

N > 1: one process per GPU.  `python bench.py --gpus N` starts them itself (a child `python -m torch.distributed.run
--nproc-per-node N bench.py ...`, spawned before this process touches torch or the GPU; rank 0's JSON line is relayed
and the child's exit code returned); under torchrun it is simply one of the ranks.  The frame is tile-split across
ranks (16x16 tiles, round-robin), total work fixed -> "strong" scaling; the exchange step is the library's native
packed RCCL gather (ptk_gather_accum: pack kernel -> grouped ncclSend/ncclRecv -> unpack kernel on rank 0), issued every
`--exchange-every` steps (default 8) and once more after the last step, inside the timed region, overlapping the next
step's trace kernel.

Prints ONE JSON line on rank 0 (see the driver contract) including
  roofline       the roof that bounds trace_kernel, chosen from the data: "valu" when the PMC-measured HBM traffic
                 (profiles/traffic_<config>.json) is under 10 % of the HBM peak, else "hbm".  VALU: SURVEY 8(d4)'s
                 algorithmic flops / ms_per_step vs the 157.3 TFLOP/s FP32 vector peak.  HBM: 8(d4)'s algorithmic
                 bytes / launch duration - or, where the caches serve most of those (the rate asked exceeds what the
                 counters saw leave the L2), the counter traffic itself.  Both sets of figures stay beside the head,
                 with the offline instruction-class bounds of the VALU pipe's occupancy (profiles/r02/).
  cpu_baseline   both CPU modes of SURVEY 8(d5), timed on this box's host cores on a bounded sample of the same
                 workload (N = 1 only): the reference's own OpenMP path as shipped (oracle/_ref, built from
                 /root/reference by __graft_entry__.build()) and the oracle port with a per-path RNG on all cores;
                 `value` is the FASTER of the two (the >= 10x target is judged against it).
  other_configs  (N = 1, default run) the BVH-walk configs C3, C4 (2 steps each) and C5 (1 step) after the timed headline, so the
                 driver's line carries the BVH-walk kernel's Msamples/s and roofline too.
"""
from __future__ import annotations

import argparse
import hashlib
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

WORKLOADS = {"C1": "Cornell box (12 tris), 512x512, 4 bounces, 16 spp",
             "C2": "Cornell box (12 tris, no textures), 1280x720, 8 bounces, 256 spp",
             "C3": "Textured PBR spheres + DOF, 1280x720, 8 bounces, 512 spp",
             "C4": "bunny stand-in (~70k tris), 1920x1080, 8 bounces, 256 spp",
             "C5": "1M-triangle height field, 1920x1080, 12 bounces, 1024 spp"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)          # ~2 s timed at the headline config: long enough for outside samplers to see it
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="C2", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=0, help="samples per pixel per step (default: the config's spp)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the C3 / C4 / C5 block after the headline")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--exchange-every", type=int, default=8,
                    help="N > 1: gather the accumulator to rank 0 every K steps (and always after the last timed step)")
    ap.add_argument("--force-exchange", action="store_true", help="run the exchange step even at N=1 (one-rank RCCL communicator)")
    ap.add_argument("--opts", default=os.environ.get("PTK_OPTS", ""), help="ptk_set_option pairs, k=v[,k=v...] (tuning experiments)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = REHEARSAL of the N>1 control flow where ranks must share one GPU (RCCL refuses that): the exchange "
                         "then goes through host copies in the library's packing order; the numbers mean nothing")
    ap.add_argument("--share-of", type=int, default=0, help="rehearsal: render only rank 0's tiles of an N-rank split on this one GPU")
    return ap.parse_args(argv)


def self_launch(args) -> int:
    """`python bench.py --gpus N` outside torchrun: start the N ranks as a CHILD process (never exec: this process may
    be watched by a profiler that has initialised the GPU), relay rank 0's JSON line, return the child's exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(1, args.gpus))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        elif ln.strip():
            print(ln, file=sys.stderr)
    if line is not None:
        print(line)
    elif p.returncode == 0:
        print("bench.py: the ranks exited 0 but rank 0 printed no result line", file=sys.stderr)
        return 4
    return p.returncode


def sha256_of(path: str):
    try:
        h = hashlib.sha256()
        with open(path, "rb") as f:
            for blk in iter(lambda: f.read(1 << 20), b""):
                h.update(blk)
        return h.hexdigest()
    except OSError:
        return None


def cpu_baseline(scene, name: str, budget_s: float = 10.0):
    """Both CPU modes of SURVEY.md 8(d5) on a bounded sample of the same workload (whole frames of 1 spp)."""
    import ctypes
    from oracle import ref_binding
    w, h = scene.width, scene.height
    modes = {}
    # (1) the reference as shipped: its own sources compiled here, workers = omp_get_max_threads() - 3, one std::mt19937
    #     shared (and raced) by all workers (pathtracer.cpp:367-371, :768-774)
    if ref_binding.available():
        ref = ref_binding.Ref()
        ref.load_scene(scene, exact_pinhole=True)
        ref.lib.ref_seed(12345)
        omp = ctypes.CDLL("libgomp.so.1")
        maxt = omp.omp_get_max_threads()
        workers = maxt - 3 if maxt > 2 else (maxt - 2 if maxt > 1 else maxt - 1)   # pathtracer.cpp:768-774
        frames, sec = 0, 0.0
        while sec < budget_s and frames < 64:
            sec += ref.render(1, 0)
            frames += 1
        modes["reference"] = {
            "value": round(w * h * frames / sec / 1e6, 4), "unit": "Msamples/s", "cores": int(workers), "kind": "reference",
            "sample": f"{frames} RenderFrame() calls (1 spp each) of {name} at {w}x{h}, depth {scene.trace_depth}; reference "
                      f"sources compiled -O2 -fopenmp, workers = omp_get_max_threads()-3, one shared mt19937",
            "binary_sha256": sha256_of(ref_binding.LIB_PATH),
            "recipe_sha256": sha256_of(os.path.join(ROOT, "oracle", "Makefile.ref"))}
    else:
        print("bench.py: oracle/_ref/libptref.so is ABSENT (it is built from /root/reference by __graft_entry__.build() in the "
              "build container): the reference-as-shipped CPU mode cannot be timed; reporting the oracle port only", file=sys.stderr)
    # (2) the "fixed" mode: our own C restatement (oracle/pt_oracle.c), per-path counter RNG, all cores
    from oracle import oracle_binding as OB
    from pbrpathtracer_amd.pathtracer import PathTracer, camera_from_scene
    from pbrpathtracer_amd import scenes as S
    pt = PathTracer()
    pts = os.path.join(tempfile.mkdtemp(prefix="bench_cpu_"), "s.pts")
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
    modes["port"] = {"value": round(w * h * frames / sec / 1e6, 4), "unit": "Msamples/s", "cores": int(cores), "kind": "port",
                     "sample": f"{frames} frames (1 spp each) of {name} at {w}x{h}, depth {scene.trace_depth}; oracle/pt_oracle.c, "
                               f"per-path counter RNG (no shared engine), OpenMP on all cores"}
    best = max(modes.values(), key=lambda m: m["value"])
    out = dict(best)
    out["modes"] = modes
    out["reference_missing"] = "reference" not in modes
    return out


def measure(name: str, args, rank: int, world: int, local_rank: int, steps: int, warmup: int, headline: bool):
    """Build config `name`, render `warmup` + `steps` steps on this rank's share, return (result dict | None, scene)."""
    import torch
    import torch.distributed as dist
    from pbrpathtracer_amd import scenes as S
    from pbrpathtracer_amd.pathtracer import PathTracer
    from pbrpathtracer_amd.distributed import NativeExchange, HostPackedExchange, owned_tile_count

    rehearsal = args.backend == "gloo"
    # ---- scene: synthesised through the reference's own formats (.obj + .pts) ---------------------
    tmp = tempfile.mkdtemp(prefix=f"bench_{name}_r{rank}_")
    t0 = time.time()
    pts, scene, cfg_spp = S.build_config(name, tmp)
    spp = args.spp if (args.spp > 0 and headline) else cfg_spp
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
    ctx.reset()

    # ---- exchange step ------------------------------------------------------------------------------
    exchange, host_accum = None, None
    if world > 1 and rehearsal:
        host_accum = torch.zeros(H * W * 3, dtype=torch.float32)
        exchange = HostPackedExchange(host_accum, W, H, dst=0)
    elif world > 1 or (args.force_exchange and headline):
        def bcast(b):
            if world == 1:
                return b
            box = [b]
            dist.broadcast_object_list(box, src=0)
            return box[0]
        NativeExchange.init_communicator(ctx, rank, world, bcast)
        if args.share_of > 1 and world == 1:
            ctx.set_tile(0, 1)
        exchange = NativeExchange(ctx, root=0)
    every = max(1, args.exchange_every)
    exchanges = [0]

    def step(first, i, last):
        ctx.render(first, spp, args.seed)
        if exchange is not None and ((i + 1) % every == 0 or last):
            if rehearsal:
                host_accum.copy_(torch.from_numpy(ctx.read_accum().reshape(-1)))    # (synchronises; rehearsal only)
            exchange.start()
            exchanges[0] += 1

    def fence():
        if exchange is not None:
            exchange.wait()
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    first = 0
    for i in range(warmup):
        step(first, i, i == warmup - 1); first += spp
    fence()
    exchanges[0] = 0
    t0 = time.perf_counter()
    for i in range(steps):
        step(first, i, i == steps - 1); first += spp
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1 and exchange is not None:
        # property check of the exchange: the gathered image holds exactly what the ranks hold together
        mine = ctx.read_accum()
        part = torch.tensor([float(mine.astype(np.float64).sum())], dtype=torch.float64)
        if not rehearsal:
            part = part.cuda()
        dist.all_reduce(part, op=dist.ReduceOp.SUM)
        if rank == 0:
            got = exchange.result()
            got = float(np.asarray(got, dtype=np.float64).sum()) if not hasattr(got, "double") else float(got.double().sum().item())
            ok = abs(got - float(part.item())) <= 1e-9 * max(1.0, abs(got))
            print(f"bench.py: gathered-image checksum {'OK' if ok else 'MISMATCH'} ({got:.6f} vs {float(part.item()):.6f}), "
                  f"exchange mode {exchange.mode}, {exchanges[0]} exchanges in the timed region", file=sys.stderr)
            if not ok:
                sys.exit(5)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # kernel time per launch, measured live with HIP events recorded on the kernels' own stream around each launch
    # (trace_kernel and accumulate_kernel separately), launch by launch, un-overlapped: in the timed loop above the
    # trace kernel of a batch starts while the previous one's last few waves - its longest paths - are still
    # finishing, which is what `value` measures; a per-launch duration only means something in isolation
    ev_ms, acc_ms = [], []
    ctx.set_option("overlap", 0)
    for _ in range(max(1, min(steps, 3))):
        ctx.render(first, spp, args.seed); first += spp
        t_ms, a_ms = ctx.last_kernel_ms()
        ev_ms.append(t_ms); acc_ms.append(a_ms)
        passes = max(1, ctx.last_render_ms()[1] // 2)          # trace_kernel launches per step (the sample buffer bounds a launch)
    ctx.set_option("overlap", 1)
    for kv in args.opts.split(","):
        if kv.startswith("overlap="):
            ctx.set_option("overlap", float(kv.split("=")[1]))
    fence()

    total_samples = float(W) * H * spp * steps
    value = total_samples / elapsed / 1e6
    ms_per_step = elapsed / steps * 1e3
    if rank != 0:
        pt.close()
        return None, scene

    # ---- traversal counts of the kernel's own counters-enabled (untimed) variant ---------------------
    ntri = ctx.bvh_info()[2]
    chunk = 8 if spp * owned_tile_count(W, H, rank, world) * 4.0 / 8.0 >= 49152.0 else 4   # ptk's automatic work-item size
    chunk = min(chunk, spp)
    ctx.set_tile(0, 1)
    ctx.set_option("chunk", chunk)
    stats = ctx.collect_stats(0, min(spp, 64), args.seed)             # long enough for the persistent waves' steady state
    ctx.set_option("chunk", 0)
    ctx.set_tile(rank, world)
    flat = ntri <= 16
    s = float(stats["samples"])
    per = {k: stats[k] / s for k in ("rays", "shadow_rays", "node_visits", "tri_tests", "hits_shaded", "tex_fetches")}
    per["max_nodes_one_ray"] = stats["max_walk_nodes"]
    live_fraction = stats["paths_started"] / s                         # paths actually traced (camera ray not a cached miss)
    nodes, depth, _ = ctx.bvh_info()
    node_boxes = ctx.node_width() if hasattr(ctx, "node_width") else 2
    # SURVEY.md §8(d4), its own record sizes and flop counts; a visited node here holds `node_boxes` child boxes =
    # that many of the survey's 32-B node records / 30-flop box tests
    d4_bytes = 27.0 + per["node_visits"] * node_boxes * 32 + per["tri_tests"] * 36 + per["hits_shaded"] * (104 + 48) + per["tex_fetches"] * 4
    d4_flops = per["node_visits"] * node_boxes * 30 + per["tri_tests"] * 50 + per["hits_shaded"] * 250
    step_samples = float(W) * H * spp                                  # whole job per step (all ranks)
    step_s = ms_per_step * 1e-3
    valu_tflops = d4_flops * step_samples / step_s / 1e12 / world      # per GPU
    d4_gbps = d4_bytes * step_samples / step_s / 1e9 / world
    kernel_ms = float(np.mean(ev_ms))
    traffic, traffic_src, cache = None, None, None
    traffic_file = os.path.join(ROOT, "profiles", f"traffic_{name}.json")
    if os.path.exists(traffic_file):
        try:
            tr = json.load(open(traffic_file))
            if tr.get("spp") == spp and tr.get("n_gpus", 1) == world:
                traffic = tr["hbm_bytes_per_launch"] * passes            # per step = per launch x launches per step
                traffic_src = tr.get("source")
                cache = tr.get("cache")
        except Exception:
            pass
    counter_gbps = traffic / (kernel_ms * 1e-3) / 1e9 if traffic is not None else None
    # which roof bounds the kernel: HBM only if the counters see at least a tenth of the HBM peak
    if counter_gbps is not None:
        bound = "hbm" if counter_gbps >= 0.1 * HBM_PEAK_GBS else "valu"
    else:
        bound = "valu"                          # no counter file for this shape: cache-resident scenes are the rule here
    if bound == "hbm":
        # The algorithmic bytes of 8(d4) are what the rays ask of the memory system.  Where the caches serve most of them
        # (asked > what the counters saw leave the L2) the distance to the HBM roof is what the counters saw, not what
        # was asked: a fraction above 1 would say nothing.
        if d4_gbps > counter_gbps:
            head = {"bound": "hbm", "achieved": round(counter_gbps, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(counter_gbps / HBM_PEAK_GBS, 4),
                    "achieved_basis": "counters: bytes that left the L2 (FETCH_SIZE x2 + WRITE_SIZE; Infinity-Cache hits included) - the "
                                      "algorithmic bytes (roofline.hbm.achieved_GBps) are mostly served by the caches"}
        else:
            head = {"bound": "hbm", "achieved": round(d4_gbps, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(d4_gbps / HBM_PEAK_GBS, 4),
                    "achieved_basis": "algorithmic bytes of SURVEY 8(d4)"}
    else:
        head = {"bound": "valu", "achieved": round(valu_tflops, 2), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(valu_tflops / VALU_PEAK_TFLOPS, 4)}
    roofline = dict(head)
    roofline.update({
        "traffic": traffic, "traffic_source": traffic_src,
        "kernel": "trace_kernel<FLAT>" if flat else "trace_kernel<BVH>", "kernel_ms_isolated": round(kernel_ms, 4),
        "trace_launches_per_step": passes,
        "accumulate_kernel_ms": round(float(np.mean(acc_ms)), 4),
        "computed_from": "ms_per_step (timed region; consecutive launches overlap their tails, so a step is shorter than an isolated launch)",
        "valu": {"flops_per_sample": round(d4_flops, 1), "achieved_TFLOPs": round(valu_tflops, 2), "frac": round(valu_tflops / VALU_PEAK_TFLOPS, 4),
                 "note": "algorithmic flops of SURVEY 8(d4): 30 per box test, 50 per triangle test, 250 per shaded hit; the peak counts an FMA "
                         "as 2 flops, which the parity contract (-ffp-contract=off) forbids, so 0.5 is this fraction's ceiling; calibrated "
                         "issue costs per instruction class: profiles/r02/valu_calibration.json"},
        "hbm": {"algorithmic_bytes_per_sample": round(d4_bytes, 1), "achieved_GBps": round(d4_gbps, 1), "frac": round(d4_gbps / HBM_PEAK_GBS, 4),
                "counter_GBps": round(counter_gbps, 1) if counter_gbps is not None else None,
                "counter_frac": round(counter_gbps / HBM_PEAK_GBS, 4) if counter_gbps is not None else None,
                "cache": cache,
                "note": "algorithmic bytes of SURVEY 8(d4) (32 B per box, 36 B per triangle, 152 B per hit, 27 B per sample); a fraction "
                        "above 1 means the bytes are served by caches, not HBM - the counter figures say how much HBM saw"},
        "per_sample": {k: round(v, 3) for k, v in per.items()},
        "live_fraction": round(live_fraction, 4),
        "kernel_variant": "FLAT (no BVH walk, scalar triangle loads)" if flat else f"BVH{node_boxes} walk, {64}-byte nodes",
        "simd_lane_utilisation": {"walk": round(stats["walk_lane_iters"] / max(1, stats["walk_wave_iters"]) / 64.0, 3),
                                  "triangles": round(stats["tri_lanes"] / max(1, stats["tri_wave_execs"]) / 64.0, 3),
                                  "shade": round(stats["shade_lanes"] / max(1, stats["shade_wave_execs"]) / 64.0, 3),
                                  "camera": round(stats["gen_lanes"] / max(1, stats["gen_wave_execs"]) / 64.0, 3)},
    })
    sq_file = os.path.join(ROOT, "profiles", "r02", f"pmc_sq_trace_kernel_{name}.json")
    if os.path.exists(sq_file) and world == 1:
        try:
            sq = json.load(open(sq_file))
            roofline["valu"]["pipe_busy_bounds_offline"] = [sq.get("valu_pipe_busy_low"), min(1.0, sq.get("valu_pipe_busy_high", 1.0))]
            roofline["valu"]["lane_utilisation_offline"] = sq.get("lane_utilisation")
            roofline["valu"]["offline_source"] = ("profiles/r02/pmc_sq_trace_kernel_%s.json: instruction-class counters of this command x the "
                                                  "calibrated issue cost of each class (lower bound: every unclassified instruction full rate; "
                                                  "upper: half rate)" % name)
        except Exception:
            pass
    if not flat:
        # the walk's own ceiling: node + triangle records gathered per CU per second vs the dependent-gather rate the
        # chip sustains for 64-byte records at this occupancy (tools/microbench/gather_bench.hip)
        rec = (per["node_visits"] + per["tri_tests"] * 0.75) * step_samples / step_s / world / 256.0 / 1e9
        roofline["gather"] = {"records64_per_s_per_cu_G": round(rec, 4),
                              "note": "64-byte-record equivalents (a 48-B triangle record = 0.75) fetched per CU; ceiling: "
                                      "profiles/r02/gather_ceiling.json"}
    out = {
        "metric": "Msamples/s (pixels*spp/s)", "value": round(value, 2), "unit": "Msamples/s",
        "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "traced_samples_per_s": round(value * live_fraction, 2),
        "config": {"workload": WORKLOADS[name], "name": name, "width": W, "height": H, "max_depth": D, "spp_per_step": spp,
                   "triangles": ntri, "bvh_nodes": nodes, "bvh_depth": depth,
                   "parallelism": f"tile-split x{world}" if world > 1 else "single GPU",
                   "exchange": ((("native packed RCCL gather (ptk_gather_accum: each rank's owned tiles, 1/N of the float accumulator, "
                                  "ncclSend/ncclRecv to rank 0)" if not rehearsal else "REHEARSAL: packed gather over gloo through host copies")
                                 + f", every {every} steps and after the last, overlapped with the next step's trace kernel")
                                if exchange is not None else "none"),
                   "exchange_every": every if exchange is not None else None,
                   "exchanges_in_timed_region": exchanges[0] if exchange is not None else 0},
        "roofline": roofline,
        "host": {"scene_gen_s": round(t_gen, 3), "scene_load_bvh_upload_s": round(t_load, 3)},
    }
    pt.close()
    return out, scene


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))             # before torch / the GPU are touched in this process

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the render path has no CPU fallback", file=sys.stderr)
        sys.exit(2)
    rehearsal = args.backend == "gloo"
    if rehearsal:
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            # torch.distributed is the rendezvous (RCCL id broadcast, barrier, max-over-ranks time); the data-path
            # collective runs on the library's own communicator and stream
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    out, scene = measure(args.config, args, rank, world, local_rank, args.steps, args.warmup, headline=True)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            base = cpu_baseline(scene, args.config)
            out["cpu_baseline"] = base
            out["gpu_over_cpu"] = round(out["value"] / base["value"], 1) if base["value"] > 0 else None
        except Exception as e:  # the baseline is reported, never required
            out["cpu_baseline"] = {"value": None, "unit": "Msamples/s", "cores": 0, "kind": "port", "sample": f"failed: {type(e).__name__}: {e}"}
    if rank == 0 and world == 1 and not args.no_other_configs and args.config == "C2" and not args.opts:
        others = {}
        for name, k in (("C3", 2), ("C4", 2), ("C5", 1)):
            try:
                o, _ = measure(name, args, 0, 1, local_rank, k, 1, headline=False)
                others[name] = {"value": o["value"], "unit": o["unit"], "ms_per_step": o["ms_per_step"], "steps": k, "warmup": 1,
                                "traced_samples_per_s": o["traced_samples_per_s"], "config": o["config"], "roofline": o["roofline"],
                                "host": o["host"]}
            except Exception as e:
                others[name] = {"value": None, "error": f"{type(e).__name__}: {e}"}
        out["other_configs"] = others
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
