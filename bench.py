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
  roofline       the roof that bounds trace_kernel, chosen from the data: "valu" when the PMC-measured traffic that left
                 the L2 (profiles/traffic_<config>.json - used only while the kernel sources still hash to what was
                 profiled, else `traffic` is null and says so) is under 10 % of the HBM peak, else "l2_miss_fabric"
                 (FETCH_SIZE counts Infinity-Cache hits too: not all of it reached HBM).  VALU: SURVEY 8(d4)'s
                 algorithmic flops / ms_per_step vs the 157.3 TFLOP/s FP32 vector peak.
  parity         image parity beside the number: after the timed region one step is rendered again from a
                 reset accumulator on the tiles of one rank of a K-way split and the CPU oracle renders the same tiles -
                 per-channel RMSE of the mean image, fraction of accumulator words that agree exactly, pixel count.
  value_contracted  the same step with the contracted build of the trace kernels (ptk_set_option "contract" 2), its own
                 roofline fraction and its own parity against the oracle (tolerance 1e-3 RMSE); `value` is the exact build.
  interactive    (N = 1; C2, C4) the reference's own use: one RenderFrame() per loop iteration + hand-off into the caller's
                 buffer (main.cpp:3563-3618, :3026-3029) - ms per RenderFrame() with and without the hand-off, launches per frame.
  cpu_baseline   both CPU modes of SURVEY 8(d5), timed on this box's host cores on a bounded sample of the same
                 workload (N = 1 only): the reference's own OpenMP path as shipped (oracle/_ref, built from
                 /root/reference by __graft_entry__.build()) and the oracle port with a per-path RNG on all cores, timed
                 at 1 and at 16 samples per call; `value` is the FASTEST (the >= 10x target is judged against it).
  other_configs  N = 1, default run: the BVH-walk configs C3, C4 (3 steps each) and C5 (2 steps) after the timed headline,
                 numbers and short keys only.  N > 1 with --with-c5 (implied by --gpus 8): BASELINE config 5 as stated -
                 C5 tile-split over the ranks, the exchange inside the timed region, gathered-image checksum.
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
    ap.add_argument("--with-c5", action="store_true", help="N > 1: append BASELINE config 5 (C5 tile-split over the ranks, 1 step) as other_configs.C5; implied by --gpus 8")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle spot check beside the number")
    ap.add_argument("--no-contracted", action="store_true", help="skip the contracted-build measurement")
    ap.add_argument("--no-interactive", action="store_true", help="skip the RenderFrame()-per-iteration measurement")
    ap.add_argument("--rank-timeout", type=float, default=150.0,
                    help="N > 1: bound in seconds of every wait on another rank (rendezvous, RCCL communicator, exchange, barriers)")
    ap.add_argument("--wall-limit", type=float, default=540.0, help="N > 1, self-launched: the parent ends the ranks after this many seconds")
    ap.add_argument("--rehearse-launch", action="store_true",
                    help="N > 1: rendezvous + barrier only, no GPU work - exercises the launcher's failure handling (tests)")
    return ap.parse_args(argv)


T_START = time.time()


def progress(msg: str) -> None:
    """One stderr line per rank per phase: what a hung or failed N > 1 run was doing last."""
    print(f"bench.py[rank {os.environ.get('RANK', '0')} +{time.time() - T_START:6.1f}s]: {msg}", file=sys.stderr, flush=True)


def fault(phase: str) -> None:
    """Test hook (PTK_BENCH_FAULT=absent:RANK:PHASE | crash:RANK:PHASE): this rank stops taking part at `phase`."""
    spec = os.environ.get("PTK_BENCH_FAULT", "")
    if not spec:
        return
    kind, rank, at = (spec.split(":") + ["", ""])[:3]
    if at == phase and rank == os.environ.get("RANK", "0"):
        progress(f"FAULT INJECTED: {kind} at {phase}")
        if kind == "crash":
            os._exit(7)
        time.sleep(1e6)


def self_launch(args) -> int:
    """`python bench.py --gpus N` outside torchrun: start the N ranks as a CHILD process group (never exec: this process may
    be watched by a profiler that has initialised the GPU), relay rank 0's JSON line, return the child's exit code.  The
    ranks' stderr - one progress line per rank per phase - passes straight through; a child that is not done after
    --wall-limit seconds is ended (its own process group, nothing else) and the run exits 124; any rank's failure ends the
    others (torch.distributed.run) and is named here."""
    import signal
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(1, args.gpus))))
    # what the run should take: start-up (the first `import torch` of a fresh box pages the image in: up to 2 min), the headline
    # steps, the contracted build's steps, per-launch timing, counters, the oracle's parity tiles, and - at 8 ranks - C5
    est = 45.0 + 2.0 * (args.steps + args.warmup) * 0.008 / max(1, args.gpus) + 25.0 + (40.0 if args.gpus == 8 or args.with_c5 else 0.0)
    print(f"bench.py: starting {args.gpus} ranks on 127.0.0.1:{port}; expected wall time about {est:.0f} s (up to {est + 120:.0f} s on a "
          f"freshly booted box), every wait on another rank bounded by {args.rank_timeout:.0f} s, hard limit {args.wall_limit:.0f} s", file=sys.stderr, flush=True)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, _ = p.communicate(timeout=args.wall_limit)
    except subprocess.TimeoutExpired:
        print(f"bench.py: the ranks were not done after {args.wall_limit:.0f} s - ending them (the last progress line of each rank above "
              f"says where it stood)", file=sys.stderr, flush=True)
        try:
            os.killpg(p.pid, signal.SIGTERM)            # the child's own session: torch.distributed.run and its ranks, nothing else
            out, _ = p.communicate(timeout=15)
        except (subprocess.TimeoutExpired, ProcessLookupError):
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
            out, _ = p.communicate()
        return 124
    line = None
    for ln in (out or "").splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        elif ln.strip():
            print(ln, file=sys.stderr)
    if p.returncode != 0:
        print(f"bench.py: torch.distributed.run exited with code {p.returncode}: a rank failed or timed out (its own message and "
              f"torchrun's failure table - rank, local_rank, exitcode - are above)", file=sys.stderr, flush=True)
        return p.returncode if p.returncode > 0 else 1
    if line is not None:
        print(line)
    else:
        print("bench.py: the ranks exited 0 but rank 0 printed no result line", file=sys.stderr)
        return 4
    return 0


def sha256_of(path: str):
    try:
        h = hashlib.sha256()
        with open(path, "rb") as f:
            for blk in iter(lambda: f.read(1 << 20), b""):
                h.update(blk)
        return h.hexdigest()
    except OSError:
        return None


def kernel_source_sha256() -> str:
    """What the committed counter files are tied to: the sources of the kernels that ran."""
    h = hashlib.sha256()
    for f in ("ptk_kernels.hip", "ptk_device.h"):
        with open(os.path.join(ROOT, "pbrpathtracer_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def cpu_baseline(scene, name: str, budget_s: float = 8.0):
    """Both CPU modes of SURVEY.md 8(d5) on a bounded sample of the same workload (whole frames)."""
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
    # (2) the "fixed" mode: our own C restatement (oracle/pt_oracle.c), per-path counter RNG, all cores, 16 x 16 tiles dealt
    #     dynamically - timed with 1 sample per call (what a RenderFrame() of the reference is) and with 16 per call
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
    for per_call, key in ((1, "port"), (16, "port_16spp")):
        total = np.zeros((h, w, 3), np.float32)
        calls, sec = 0, 0.0
        while sec < budget_s * 0.6 and calls < 64:
            t0 = time.time()
            o.render(ocam, w, h, scene.trace_depth, calls * per_call, per_call, 1, total=total, threads=cores, want_rgb8=False)
            sec += time.time() - t0
            calls += 1
        modes[key] = {"value": round(w * h * per_call * calls / sec / 1e6, 4), "unit": "Msamples/s", "cores": int(cores), "kind": "port",
                      "sample": f"{calls} calls of {per_call} spp of {name} at {w}x{h}, depth {scene.trace_depth}; oracle/pt_oracle.c, per-path "
                                f"counter RNG (no shared engine), OpenMP on all cores, 16x16 tiles dealt dynamically"}
    best = max(modes.values(), key=lambda m: m["value"])
    out = dict(best)
    out["modes"] = {k: {kk: vv for kk, vv in v.items() if kk in ("value", "cores", "kind")} for k, v in modes.items()}
    if "reference" in modes:
        out["reference_binary_sha256"] = modes["reference"]["binary_sha256"]
        out["reference_recipe_sha256"] = modes["reference"]["recipe_sha256"]
    out["reference_missing"] = "reference" not in modes
    return out


def parity_check(pt, ctx, scene, W, H, D, spp, seed, levels):
    """Image parity beside the number: one step from a reset accumulator on the tiles of one rank of a K-way
    split (K sized for ~3 M samples), against the CPU oracle on the same tiles.  Per contract level: per-channel RMSE of the
    mean image, fraction of accumulator words equal bit for bit, pixels compared."""
    from oracle import oracle_binding as OB
    from pbrpathtracer_amd.pathtracer import camera_from_scene
    from pbrpathtracer_amd.distributed import tile_owner_mask, owned_tile_count
    tiles = owned_tile_count(W, H, 0, 1)
    budget = 1.0e6 if len(pt.StagedScene()["material"]) > 300000 else 3.0e6      # (the oracle walks a 1 M-triangle tree ~4x slower)
    K = max(1, int(np.ceil(tiles * 256.0 * spp / budget)))
    r = K // 3
    cam = camera_from_scene(scene)
    if scene.pinhole:
        cam["aperture"] = 0.0
    t0 = time.time()
    o = OB.Oracle(pt.StagedScene())
    ocam = OB.make_camera(cam["pos"], cam["dir"], cam["up"], cam["focal"], cam["fovy"], cam["focal_dist"], cam["aperture"])
    ref, _ = o.render(ocam, W, H, D, 0, spp, seed, rank=r, world=K, want_rgb8=False)
    o.close()
    t_oracle = time.time() - t0
    mask = tile_owner_mask(W, H, r, K)[::-1]                      # accumulator rows are bottom-up
    out = {}
    for level in levels:
        ctx.set_option("contract", level)
        ctx.set_tile(r, K)
        ctx.reset()
        ctx.render(0, spp, seed)
        got = ctx.read_accum()
        d = (got[mask].astype(np.float64) - ref[mask].astype(np.float64)) / spp
        rmse = np.sqrt((d ** 2).mean(axis=0))
        out[level] = {"rmse": [float(f"{x:.3e}") for x in rmse], "exact_fraction": round(float(np.mean(got[mask] == ref[mask])), 6),
                      "pixels": int(mask.sum()), "spp": int(spp), "split": f"rank {r} of {K}", "oracle_s": round(t_oracle, 2),
                      "tolerance_rmse": 1e-3, "ok": bool((rmse <= 1e-3).all() and np.isfinite(got).all())}
    ctx.set_option("contract", 0)
    return out


def interactive_probe(pt, ctx, W, H, frames=200):
    """The reference's own caller: one RenderFrame() per loop iteration followed by the hand-off of texData
    (PathTracerLoop main.cpp:3563-3618, glTexSubImage2D :3026-3029)."""
    out = {}
    pinned = pt.AllocOutImage()
    pt.SetOutImage(pinned); pt.ResetImage()
    for _ in range(10):
        pt.RenderFrame()
    t0 = time.perf_counter()
    for _ in range(frames):
        pt.RenderFrame()
    out["with_handoff_ms"] = round((time.perf_counter() - t0) / frames * 1e3, 4)
    out["launches_per_frame"] = int(ctx.last_render_ms()[1])
    host = np.array(pinned)
    dev = ctx.resolve_rgb8()                    # into another buffer: a real copy of the device's resolved frame
    pt.SetOutImage(None)
    for _ in range(10):
        pt.RenderFrame()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        pt.RenderFrame()
    ctx.synchronize()
    out["ms_per_RenderFrame"] = round((time.perf_counter() - t0) / frames * 1e3, 4)
    out["handoff"] = "8-bit resolve written by accumulate_kernel straight into the caller's page-locked buffer (ptk_host_alloc, ptk_bind_out_image); no copy command"
    # ... and into ORDINARY memory, what the viewer's `new GLubyte[w*h*3]` is (main.cpp:3435): never registered with the runtime (round 4),
    # the resolved frame is copied into it after every RenderFrame()
    plain = np.zeros((H, W, 3), np.uint8)
    pt.SetOutImage(plain)
    for _ in range(10):
        pt.RenderFrame()
    t0 = time.perf_counter()
    for _ in range(frames):
        pt.RenderFrame()
    out["with_pageable_handoff_ms"] = round((time.perf_counter() - t0) / frames * 1e3, 4)
    out["pageable_frame_matches_device"] = bool(np.array_equal(plain, ctx.resolve_rgb8()))
    pt.SetOutImage(None)
    # ... and with the frame staying on the GPU (ptk_bind_out_device: what ptk_bind_gl_buffer maps the viewer's pixel-unpack buffer
    # to - the OpenGL leg itself cannot run on a headless box)
    import torch
    dbuf = torch.zeros((H, W, 3), dtype=torch.uint8, device=f"cuda:{torch.cuda.current_device()}"); torch.cuda.synchronize()
    pt.SetOutDeviceImage(dbuf.data_ptr())
    for _ in range(10):
        pt.RenderFrame()
    t0 = time.perf_counter()
    for _ in range(frames):
        pt.RenderFrame()                        # (returns with the frame in the buffer)
    out["with_device_handoff_ms"] = round((time.perf_counter() - t0) / frames * 1e3, 4)
    out["device_frame_matches"] = bool(np.array_equal(dbuf.cpu().numpy(), ctx.resolve_rgb8()))
    pt.SetOutDeviceImage(None)
    out["frame_matches_device"] = bool(np.array_equal(host, dev)) and bool(host.any())
    out["Msamples_per_s_with_handoff"] = round(W * H / out["with_handoff_ms"] / 1e3, 1)
    del pinned
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
    ctx.set_option("comm_timeout_s", args.rank_timeout)           # every wait of the exchange step is bounded
    for kv in args.opts.split(","):
        if "=" in kv:
            k, v = kv.split("=")
            ctx.set_option(k, float(v))
    ctx.reset()
    if world > 1:
        progress(f"{name}: scene loaded ({t_load:.2f} s), BVH built, first frame rendered")

    # ---- exchange step ------------------------------------------------------------------------------
    exchange, host_accum = None, None
    comm = {"rccl_ranks": None, "devices": [local_rank]}
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
        info = ctx.comm_info()                  # what the library's own communicator reports - not torch's WORLD_SIZE
        comm["rccl_ranks"] = info["world"]
        comm["devices"] = [info["comm_device"]]
        if world > 1:
            progress(f"{name}: RCCL communicator up ({info['world']} ranks, this one on device {info['comm_device']})")
    if world > 1:
        # the distinct HIP ordinals the ranks render on (one node: N ranks must mean N devices)
        box = [None] * world
        dist.all_gather_object(box, (comm["devices"][0], comm["rccl_ranks"]))
        comm["devices"] = sorted({int(d) for d, _ in box})
        ranks_seen = {r for _, r in box}
        if len(ranks_seen) == 1 and comm["rccl_ranks"] is None:
            comm["rccl_ranks"] = None
    every = max(1, args.exchange_every)
    exchanges = [0]

    def step(first, i, last):
        ctx.render(first, spp, args.seed)
        if exchange is not None and ((i + 1) % every == 0 or last):
            fault("exchange")
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

    first = [0]

    launch_log = {}

    def timed(n_warm, n_steps, log_key=None):
        """n_warm untimed steps, then exactly n_steps between two fences; max over ranks; returns seconds.  With log_key the
        duration of EVERY trace launch inside the timed region is kept (HIP events on the launch's own stream: ptk_kernel_log)."""
        for i in range(n_warm):
            step(first[0], i, i == n_warm - 1); first[0] += spp
        fence()
        if log_key:
            ctx.kernel_log(min(n_steps * 16 + 16, 1 << 16))
        exchanges[0] = 0
        t0 = time.perf_counter()
        for i in range(n_steps):
            step(first[0], i, i == n_steps - 1); first[0] += spp
        fence()
        el = time.perf_counter() - t0
        if log_key:
            launch_log[log_key] = ctx.kernel_log_read()
            ctx.kernel_log(0)
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    elapsed = timed(warmup, steps, log_key="exact")
    timed_exchanges = exchanges[0]
    if world > 1:
        progress(f"{name}: {steps} timed steps done, {elapsed / steps * 1e3:.3f} ms per step, {timed_exchanges} exchanges")
    checksum = None
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
            checksum = {"gathered": got, "sum_of_ranks": float(part.item()), "ok": bool(ok)}
            print(f"bench.py: {name} gathered-image checksum {'OK' if ok else 'MISMATCH'} ({got:.6f} vs {float(part.item()):.6f}), "
                  f"exchange mode {exchange.mode}, {timed_exchanges} exchanges in the timed region", file=sys.stderr)
            if not ok:
                sys.exit(5)

    # ---- the contracted build of the trace kernels on the same steps (every rank; value stays the exact build's) ----
    contracted = None
    if not args.no_contracted and "contract=" not in args.opts:
        ctx.set_option("contract", 2)
        n_c = max(1, min(steps, 100))
        el_c = timed(min(warmup, 2), n_c)
        ctx.set_option("contract", 0)
        contracted = {"level": 2, "steps": n_c, "elapsed": el_c}

    # kernel time per launch, measured live with HIP events recorded on the kernels' own stream around each launch
    # (trace_kernel and accumulate_kernel separately), launch by launch, un-overlapped: in the timed loop above the
    # trace kernel of a batch starts while the previous one's last few waves - its longest paths - are still
    # finishing, which is what `value` measures; a per-launch duration only means something in isolation
    ev_ms, acc_ms = [], []
    ctx.set_option("overlap", 0)
    for _ in range(max(1, min(steps, 5))):
        ctx.render(first[0], spp, args.seed); first[0] += spp
        t_ms, a_ms = ctx.last_kernel_ms()                       # (waits for the render: launches are isolated from one another)
        ev_ms.append(t_ms); acc_ms.append(a_ms)
        passes = max(1, ctx.last_render_ms()[1] // 3)          # trace_kernel launches per step (the sample buffer bounds a launch)
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
    # what the kernels really ask of the memory system: a 64-B BVH4 node, a 48-B triangle record, 112-B shading + 96-B material
    req_bytes = 27.0 + 16.0 * live_fraction + per["node_visits"] * 64 + per["tri_tests"] * 48 + per["hits_shaded"] * (112 + 96) + per["tex_fetches"] * 4
    step_samples = float(W) * H * spp                                  # whole job per step (all ranks)
    step_s = ms_per_step * 1e-3
    valu_tflops = d4_flops * step_samples / step_s / 1e12 / world      # per GPU
    d4_gbps = d4_bytes * step_samples / step_s / 1e9 / world
    # per step, isolated: the MEDIAN of the launches above (every one of them is in the line: the first after a pause runs at
    # another clock than the rest, which a mean hides - VERDICT r03 item 3)
    kernel_ms = float(np.median(ev_ms))
    # per launch, INSIDE the timed region (overlapped tails included): what roofline.frac_from_kernel_ms is computed from
    in_loop = np.array(launch_log.get("exact") or [0.0], np.float64)
    loop_launch_ms = float(np.median(in_loop))
    traffic, traffic_src, cache, traffic_note = None, None, None, None
    traffic_file = os.path.join(ROOT, "profiles", f"traffic_{name}.json")
    if os.path.exists(traffic_file):
        try:
            tr = json.load(open(traffic_file))
            if tr.get("spp") == spp and tr.get("n_gpus", 1) == world:
                if tr.get("kernel_source_sha256") == kernel_source_sha256():
                    traffic = tr["hbm_bytes_per_launch"] * passes        # per step = per launch x launches per step
                    traffic_src = tr.get("source")
                    cache = tr.get("cache")
                else:
                    traffic_note = ("profiles/traffic_%s.json was collected for other kernel sources (its kernel_source_sha256 differs "
                                    "from pbrpathtracer_amd/csrc/ptk_kernels.hip + ptk_device.h as built): not used" % name)
        except Exception:
            pass
    counter_gbps = traffic / (kernel_ms * 1e-3) / 1e9 if traffic is not None else None
    # which roof bounds the kernel: the memory side only if the counters see at least a tenth of the HBM peak leave the L2
    if counter_gbps is not None and counter_gbps >= 0.1 * HBM_PEAK_GBS:
        # FETCH_SIZE counts what left the L2, Infinity-Cache hits included: an upper bound of what HBM saw
        head = {"bound": "l2_miss_fabric", "achieved": round(counter_gbps, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(counter_gbps / HBM_PEAK_GBS, 4),
                "achieved_basis": "counters: bytes that left the L2 (FETCH_SIZE x2 + WRITE_SIZE), Infinity-Cache hits included - an upper bound "
                                  "of the HBM traffic, priced against the HBM peak"}
    else:
        head = {"bound": "valu", "achieved": round(valu_tflops, 2), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(valu_tflops / VALU_PEAK_TFLOPS, 4)}
    roofline = dict(head)
    roofline.update({
        "traffic": traffic, "traffic_source": traffic_src, "traffic_note": traffic_note,
        "kernel": "trace_kernel<FLAT>" if flat else "trace_kernel<BVH>", "kernel_ms_isolated": round(kernel_ms, 4),
        "kernel_ms_isolated_each": [round(float(x), 4) for x in ev_ms],
        "kernel_ms_isolated_note": "trace launches of one step summed, per isolated step (overlap off, each waited for); the headline figure is their median",
        "trace_launches_per_step": passes,
        # every trace launch of the TIMED region, event-timed on its own stream while the launches overlap their tails
        "kernel_ms_in_loop": {"launches": int(in_loop.size), "min": round(float(in_loop.min()), 4), "median": round(loop_launch_ms, 4),
                              "mean": round(float(in_loop.mean()), 4), "max": round(float(in_loop.max()), 4)},
        "frac_from_kernel_ms": round(d4_flops * step_samples / max(1e-9, loop_launch_ms * passes * 1e-3) / 1e12 / world / VALU_PEAK_TFLOPS, 4),
        "frac_from_kernel_ms_note": "SURVEY 8(d4) flops of one step / (median in-loop launch duration x launches per step) / 157.3 TFLOP/s: recompute it from "
                                    "valu.flops_per_sample x width x height x spp_per_step and profiles/r04/rocprofv3_kernel_stats_<config>.csv",
        "accumulate_kernel_ms": round(float(np.mean(acc_ms)), 4),
        "computed_from": "ms_per_step (timed region; consecutive launches overlap their tails - profiles/r04/overlap_trace_C2.json - so a step is shorter than an isolated launch)",
        "valu": {"flops_per_sample": round(d4_flops, 1), "achieved_TFLOPs": round(valu_tflops, 2), "frac": round(valu_tflops / VALU_PEAK_TFLOPS, 4),
                 "note": "SURVEY 8(d4) flops: 30 / box test, 50 / triangle test, 250 / shaded hit; the peak counts an FMA as 2 flops, which the "
                         "exact build (-ffp-contract=off) forgoes: 0.5 is its ceiling (value_contracted lifts that)"},
        "hbm": {"algorithmic_bytes_per_sample": round(d4_bytes, 1), "requested_bytes_per_sample": round(req_bytes, 1),
                "achieved_GBps": round(d4_gbps, 1), "frac": round(d4_gbps / HBM_PEAK_GBS, 4),
                "counter_GBps": round(counter_gbps, 1) if counter_gbps is not None else None,
                "counter_frac": round(counter_gbps / HBM_PEAK_GBS, 4) if counter_gbps is not None else None,
                "cache": cache,
                "note": "algorithmic = SURVEY 8(d4) record sizes (32 B / box, 36 B / triangle, 152 B / hit, 27 B / sample); requested = the "
                        "kernels' own records (64-B BVH4 node, 48-B triangle, 208 B / hit, 16-B sample write); > 1 means served by caches"},
        "per_sample": {k: round(v, 3) for k, v in per.items()},
        "live_fraction": round(live_fraction, 4),
        "kernel_variant": "FLAT (no BVH walk, scalar triangle loads)" if flat else f"BVH{node_boxes} walk, {64}-byte nodes",
        "simd_lane_utilisation": {"walk": round(stats["walk_lane_iters"] / max(1, stats["walk_wave_iters"]) / 64.0, 3),
                                  "triangles": round(stats["tri_lanes"] / max(1, stats["tri_wave_execs"]) / 64.0, 3),
                                  "shade": round(stats["shade_lanes"] / max(1, stats["shade_wave_execs"]) / 64.0, 3),
                                  "camera": round(stats["gen_lanes"] / max(1, stats["gen_wave_execs"]) / 64.0, 3)},
    })
    sq = None
    for rdir in ("r04", "r03", "r02"):
        sq_file = os.path.join(ROOT, "profiles", rdir, f"pmc_sq_trace_kernel_{name}.json")
        if os.path.exists(sq_file) and world == 1:
            try:
                sq = json.load(open(sq_file))
                if rdir != "r02" and sq.get("kernel_source_sha256") != kernel_source_sha256():
                    sq = None
                    continue
                roofline["valu"]["pipe_busy_bounds_offline"] = [sq.get("valu_pipe_busy_low"), min(1.0, sq.get("valu_pipe_busy_high", 1.0))]
                roofline["valu"]["lane_utilisation_offline"] = sq.get("lane_utilisation")
                roofline["valu"]["offline_source"] = f"profiles/{rdir}/pmc_sq_trace_kernel_{name}.json" + ("" if rdir != "r02" else " (an earlier round's kernels)")
                break
            except Exception:
                pass
    if not flat:
        # the walk's own ceiling: node + triangle records gathered per CU per second vs the dependent-gather rate the
        # chip sustains for 64-byte records at this occupancy (tools/microbench/gather_bench.hip)
        rec = (per["node_visits"] + per["tri_tests"] * 0.75) * step_samples / step_s / world / 256.0 / 1e9
        roofline["gather"] = {"records64_per_s_per_cu_G": round(rec, 4), "ceiling_source": "profiles/r02/gather_ceiling.json"}
    out = {
        "metric": "Msamples/s (pixels*spp/s)", "value": round(value, 2), "unit": "Msamples/s",
        "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        # what compares across rounds and configs: the samples that are actually traced (value counts every pixel of the frame;
        # live_fraction = pixels whose camera rays can hit anything - the exact culls skip the rest)
        "traced_samples_per_s": round(value * live_fraction, 2), "live_fraction": round(live_fraction, 4),
        "config": {"workload": WORKLOADS[name], "name": name, "width": W, "height": H, "max_depth": D, "spp_per_step": spp,
                   "triangles": ntri, "bvh_nodes": nodes, "bvh_depth": depth,
                   "parallelism": f"tile-split x{world}" if world > 1 else "single GPU",
                   "rccl_ranks": comm["rccl_ranks"], "devices": comm["devices"],
                   "exchange": ((("native packed RCCL gather (ptk_gather_accum: each rank's owned tiles, 1/N of the float accumulator, "
                                  "ncclSend/ncclRecv to rank 0)" if not rehearsal else "REHEARSAL: packed gather over gloo through host copies")
                                 + f", every {every} steps and after the last, overlapped with the next step's trace kernel")
                                if exchange is not None else "none"),
                   "exchange_every": every if exchange is not None else None,
                   "exchanges_in_timed_region": timed_exchanges if exchange is not None else 0,
                   "gathered_checksum": checksum},
        "roofline": roofline,
        "host": {"scene_gen_s": round(t_gen, 3), "scene_load_bvh_upload_s": round(t_load, 3)},
    }
    if contracted is not None:
        v_c = float(W) * H * spp * contracted["steps"] / contracted["elapsed"] / 1e6
        out["value_contracted"] = {"value": round(v_c, 2), "unit": "Msamples/s", "ms_per_step": round(contracted["elapsed"] / contracted["steps"] * 1e3, 4),
                                   "steps": contracted["steps"], "over_exact": round(v_c / value, 4),
                                   "build": "trace kernels with -ffp-contract=fast + 1-ulp v_rcp / v_sqrt / v_rsq (ptk_set_option contract=2)",
                                   "roofline_frac": round(valu_tflops * (v_c / value) / VALU_PEAK_TFLOPS, 4) if head["bound"] == "valu" else None}
    # ---- image parity beside the number (rank 0; the other ranks wait at the next rendezvous) ----------------------------
    if not args.no_parity:
        try:
            levels = [0] + ([2] if contracted is not None else [])
            ctx.set_tile(0, 1)
            par = parity_check(pt, ctx, scene, W, H, D, spp, args.seed, levels)
            ctx.set_tile(rank, world)
            out["parity"] = par[0]
            if 2 in par and "value_contracted" in out:
                out["value_contracted"]["parity"] = {k: par[2][k] for k in ("rmse", "exact_fraction", "pixels", "ok")}
        except Exception as e:
            out["parity"] = {"error": f"{type(e).__name__}: {e}"}
    # ---- the interactive loop (N = 1): RenderFrame() per iteration + hand-off ------------------------------------------
    if world == 1 and not args.no_interactive and name in ("C2", "C4") and not args.opts and args.share_of <= 1:
        try:
            out["interactive"] = interactive_probe(pt, ctx, W, H)
        except Exception as e:
            out["interactive"] = {"error": f"{type(e).__name__}: {e}"}
    pt.close()
    return out, scene


def short(o):
    """numbers and short keys of a measure() result, for other_configs (the driver keeps only the tail of the line)"""
    r = o["roofline"]
    d = {"value": o["value"], "ms_per_step": o["ms_per_step"], "steps": o["steps"], "warmup": o["warmup"], "n_gpus": o["n_gpus"],
         "traced_samples_per_s": o["traced_samples_per_s"], "live_fraction": o["live_fraction"],
         "workload": o["config"]["workload"], "triangles": o["config"]["triangles"], "bvh_nodes": o["config"]["bvh_nodes"],
         "roofline": {"bound": r["bound"], "achieved": r["achieved"], "peak": r["peak"], "unit": r["unit"], "frac": r["frac"],
                      "traffic": r["traffic"], "kernel_ms_isolated": r["kernel_ms_isolated"], "kernel_ms_isolated_each": r["kernel_ms_isolated_each"],
                      "kernel_ms_in_loop": r["kernel_ms_in_loop"], "frac_from_kernel_ms": r["frac_from_kernel_ms"], "launches_per_step": r["trace_launches_per_step"],
                      "valu_frac": r["valu"]["frac"], "lane_utilisation_offline": r["valu"].get("lane_utilisation_offline"),
                      "l2_hit_rate": (r["hbm"]["cache"] or {}).get("l2_hit_rate") if r["hbm"].get("cache") else None,
                      "counter_GBps": r["hbm"]["counter_GBps"], "per_sample": r["per_sample"], "simd_lane_utilisation": r["simd_lane_utilisation"]},
         "parity": o.get("parity"), "host": o["host"]}
    if "value_contracted" in o:
        vc = o["value_contracted"]
        d["value_contracted"] = {k: vc.get(k) for k in ("value", "over_exact", "roofline_frac", "parity")}
    if "interactive" in o:
        d["interactive"] = o["interactive"]
    if o["n_gpus"] > 1:
        d["config"] = {k: o["config"][k] for k in ("parallelism", "rccl_ranks", "devices", "exchanges_in_timed_region", "gathered_checksum")}
    return d


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

    import datetime
    import torch
    import torch.distributed as dist
    pg_timeout = datetime.timedelta(seconds=args.rank_timeout)

    if args.rehearse_launch:
        # launcher rehearsal (CPU): rendezvous, one barrier, a stub line - no GPU, no rendering
        progress("up")
        fault("rendezvous")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=pg_timeout)
        progress("process group up")
        fault("barrier")
        dist.barrier()
        progress("barrier passed")
        dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"metric": "launch rehearsal", "value": 0.0, "n_gpus": world}))
        return

    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the render path has no CPU fallback", file=sys.stderr)
        sys.exit(2)
    rehearsal = args.backend == "gloo"
    if rehearsal:
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        progress(f"up on HIP device {local_rank} of {torch.cuda.device_count()}")
        fault("rendezvous")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=pg_timeout)
        else:
            # torch.distributed is the rendezvous (RCCL id broadcast, barrier, max-over-ranks time); the data-path
            # collective runs on the library's own communicator and stream
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank), timeout=pg_timeout)
        progress("process group up")

    out, scene = measure(args.config, args, rank, world, local_rank, args.steps, args.warmup, headline=True)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            base = cpu_baseline(scene, args.config)
            out["cpu_baseline"] = base
            out["gpu_over_cpu"] = round(out["value"] / base["value"], 1) if base["value"] > 0 else None
        except Exception as e:  # the baseline is reported, never required
            out["cpu_baseline"] = {"value": None, "unit": "Msamples/s", "cores": 0, "kind": "port", "sample": f"failed: {type(e).__name__}: {e}"}
    others = {}
    if world == 1 and not args.no_other_configs and args.config == "C2" and not args.opts:
        # the BVH-walk configs at their full BASELINE size, through the identical code path (numbers + short keys only)
        for name, k in (("C3", 4), ("C4", 4), ("C5", 3)):
            try:
                o, _ = measure(name, args, 0, 1, local_rank, k, 2, headline=False)
                others[name] = short(o)
            except Exception as e:
                others[name] = {"value": None, "error": f"{type(e).__name__}: {e}"}
    elif world > 1 and (args.with_c5 or world == 8) and args.config != "C5":
        # BASELINE config 5 as stated: the 1 M-triangle frame tile-split over the ranks, 1024 spp per step, the exchange
        # inside the timed region (after the last step), gathered-image checksum - every rank runs it
        try:
            o, _ = measure("C5", args, rank, world, local_rank, 1, 1, headline=False)
            if rank == 0:
                others["C5"] = short(o)
        except Exception as e:
            if rank == 0:
                others["C5"] = {"value": None, "error": f"{type(e).__name__}: {e}"}
    if rank == 0 and others:
        out["other_configs"] = others
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
