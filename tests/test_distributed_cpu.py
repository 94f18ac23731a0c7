"""N > 1 path on CPU: world_size-2 gloo processes, each renders only its tiles (with the oracle standing
in for the GPU), then the exchange step of pbrpathtracer_amd.distributed combines them.  Checks that the
tile partition is disjoint and complete and that the reduced image equals the single-rank image bit for
bit — the property the RCCL path relies on."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, GOLDEN


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from oracle import oracle_binding as OB
    from pbrpathtracer_amd import distributed as D
    dist.init_process_group("gloo", rank=rank, world_size=world)
    z = np.load(os.path.join(GOLDEN, "tier_s_glass.npz"))
    arrays = {k[6:]: z[k] for k in z.files if k.startswith("scene_")}
    o = OB.Oracle(arrays)
    cam = OB.make_camera(z["cam"][0:3], z["cam"][3:6], z["cam"][6:9], float(z["proj"][0]), float(z["proj"][1]),
                         float(z["focal_dist"]), float(z["aperture"]))
    W, H, Dp, spp = 70, 50, 5, 3
    total, _ = o.render(cam, W, H, Dp, 0, spp, 9, rank=rank, world=world, threads=2, want_rgb8=False)
    # ownership as the Python mirror predicts it (rows of `total` are bottom-up)
    mask = D.tile_owner_mask(W, H, rank, world)[::-1]
    assert not total[~mask].any()
    local = torch.from_numpy(total.reshape(-1).copy())
    out = D.gather_accumulator(local, dst=0)
    # a second batch keeps accumulating locally and is gathered again
    total2, _ = o.render(cam, W, H, Dp, spp, 2, 9, total=total.copy(), rank=rank, world=world, threads=2, want_rgb8=False)
    out2 = D.gather_accumulator(torch.from_numpy(total2.reshape(-1).copy()), dst=0)
    if rank == 0:
        full, _ = o.render(cam, W, H, Dp, 0, spp, 9, threads=2, want_rgb8=False)
        full2, _ = o.render(cam, W, H, Dp, spp, 2, 9, total=full.copy(), threads=2, want_rgb8=False)
        np.save(os.path.join(out_dir, "ok.npy"), np.array([
            np.array_equal(out.numpy().reshape(H, W, 3), full),
            np.array_equal(out2.numpy().reshape(H, W, 3), full2)]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_tile_split_and_gather(tmp_path):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    ok = np.load(str(tmp_path / "ok.npy"))
    assert ok.all()


def test_tile_masks_partition_the_frame():
    from pbrpathtracer_amd import distributed as D
    for (w, h, world) in [(1280, 720, 8), (1920, 1080, 8), (70, 50, 3), (16, 16, 2), (5, 5, 4)]:
        acc = np.zeros((h, w), np.int32)
        for r in range(world):
            m = D.tile_owner_mask(w, h, r, world)
            acc += m
            tiles = {(y // 16, x // 16) for y, x in zip(*np.nonzero(m))}
            assert len(tiles) == D.owned_tile_count(w, h, r, world)
        assert (acc == 1).all()
    # load balance of the benchmark frame: every rank owns the same number of tiles +- 1
    counts = [D.owned_tile_count(1280, 720, r, 8) for r in range(8)]
    assert max(counts) - min(counts) <= 1


def _exchange_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from pbrpathtracer_amd import distributed as D
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W, H = 70, 50                                   # ragged: 5 x 4 tiles, the last column / row partial
    rng = np.random.default_rng(123)
    full = rng.random((H, W, 3), dtype=np.float32) + 1.0            # what a single rank would hold (rows bottom-up)
    mask = D.tile_owner_mask(W, H, rank, world)[::-1]
    mine = np.where(mask[..., None], full, 0.0).astype(np.float32)
    ok = []
    local = torch.from_numpy(mine.reshape(-1).copy())
    ex = D.HostPackedExchange(local, W, H, dst=0)                   # packing order = the library's ptk_packed_layout
    ex.start()
    res = ex.wait()
    if rank == 0:
        ok += [np.array_equal(res.numpy().reshape(H, W, 3), full)]
    # a second batch: the local accumulator has grown, the exchange object is re-used
    local += torch.from_numpy(mine.reshape(-1))
    ex.start()
    res2 = ex.wait()
    # the comparator: sum-reduce of the zero-padded buffers gives the same image bit for bit
    red = D.gather_accumulator(local, dst=0)
    if rank == 0:
        ok += [np.array_equal(res2.numpy().reshape(H, W, 3), 2 * full), np.array_equal(red.numpy(), res2.numpy())]
    if rank == 0:
        from pbrpathtracer_amd import ptk
        lens = [ptk.packed_floats(W, H, r, world) for r in range(world)]
        valid = [int((ptk.packed_layout(W, H, r, world) >= 0).sum()) for r in range(world)]
        ok += [sum(valid) == W * H * 3, all(n % 768 == 0 for n in lens), sum(lens) == 5 * 4 * 768]
        np.save(os.path.join(out_dir, "ok.npy"), np.array(ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_packed_gather_exchange(tmp_path, world):
    """The packed gather in the NATIVE packing order (ptk_packed_layout, the order the device pack / unpack kernels are
    tested against on the GPU): every rank sends the 768-float blocks of its owned tiles (uneven counts), rank 0
    scatters them into place; equals the single-rank image and the sum-reduce form, also on re-use."""
    import torch.multiprocessing as mp
    port = 31500 + (os.getpid() % 2000) + world
    mp.spawn(_exchange_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert np.load(str(tmp_path / "ok.npy")).all()


def test_packed_layout_matches_the_tile_map():
    """ptk_packed_layout (host-only entry of libptk) against the Python mirror of the tile -> rank map."""
    from pbrpathtracer_amd import distributed as D, ptk
    for (w, h, world) in [(70, 50, 3), (1280, 720, 8), (16, 16, 2), (5, 5, 4), (33, 17, 1)]:
        seen = np.zeros(w * h * 3, np.int32)
        for r in range(world):
            lay = ptk.packed_layout(w, h, r, world)
            assert len(lay) == D.owned_tile_count(w, h, r, world) * 768
            idx = lay[lay >= 0]
            seen[idx] += 1
            mask = D.tile_owner_mask(w, h, r, world)[::-1].reshape(-1)          # rows bottom-up like the accumulator
            assert mask[idx // 3].all()
            # inside a tile: row-major from the tile's top-left, RGB innermost
            if len(lay) == 0:
                continue                                        # a rank without tiles (16 x 16 frame, 2 ranks)
            first = lay[:768].reshape(16, 16, 3)
            on = first[..., 0] >= 0
            assert (first[..., 1][on] == first[..., 0][on] + 1).all() and (first[..., 2][on] == first[..., 0][on] + 2).all()
        assert (seen == 1).all()
    assert ptk.load().ptk_packed_floats(0, 10, 0, 1) < 0 and ptk.load().ptk_packed_floats(10, 10, 2, 2) < 0


@pytest.mark.parametrize("fault,culprit", [("absent:1:rendezvous", None), ("absent:1:barrier", None), ("crash:1:barrier", "exitcode  : 7")])
def test_bench_launcher_ends_a_run_with_a_missing_or_dead_rank(fault, culprit):
    """the first real `bench.py --gpus 8` is the driver's, so it must not be able to hang.  The launcher
    rehearsal (`--rehearse-launch`: rendezvous + barrier over gloo, no GPU work) with one rank that never arrives, stops before a
    barrier, or dies: every wait on another rank is bounded by --rank-timeout, the parent exits NON-ZERO well inside the bound,
    prints no result line, and the ranks' progress lines / torchrun's failure table say which rank it was."""
    import subprocess
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["PTK_BENCH_FAULT"] = fault
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-launch", "--rank-timeout", "6", "--wall-limit", "90"],
                       capture_output=True, text=True, timeout=240, cwd=ROOT, env=env)
    took = time.time() - t0
    assert r.returncode != 0 and took < 80, (r.returncode, took, r.stderr[-1500:])
    assert '"metric"' not in r.stdout
    assert "FAULT INJECTED" in r.stderr and "bench.py[rank 1" in r.stderr             # the culprit's last progress line
    assert "a rank failed or timed out" in r.stderr
    if culprit:
        assert culprit in r.stderr and "rank      : 1" in r.stderr


def test_bench_launcher_wall_limit_ends_ranks_that_never_finish():
    """... and when no rank's own bound fires in time (here: a bound of ten minutes), the parent's wall limit ends the child's
    process group - torch.distributed.run and its ranks, nothing else - and the run exits 124."""
    import subprocess
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["PTK_BENCH_FAULT"] = "absent:0:rendezvous"
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-launch", "--rank-timeout", "600", "--wall-limit", "12"],
                       capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
    assert r.returncode == 124 and time.time() - t0 < 60, (r.returncode, r.stderr[-1500:])
    assert "were not done after 12 s" in r.stderr and '"metric"' not in r.stdout
