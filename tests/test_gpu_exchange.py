"""The multi-GPU exchange step on the one GPU a test box has: a single-rank RCCL communicator exercises
ptk_gather_accum (native RCCL) and the torch.distributed path bench.py uses for N > 1
(bind_accum + set_stream + render + reduce), end to end.  The N = 2 arithmetic is covered on CPU
(tests/test_distributed_cpu.py)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import load_golden, scene_from_golden

pytestmark = pytest.mark.gpu


def _ctx():
    from pbrpathtracer_amd import ptk
    z = load_golden("tier_s_cornell.npz")
    c = ptk.Context(0)
    c.upload_scene(scene_from_golden(z))
    cam, proj = z["cam"], z["proj"]
    c.set_camera(cam[0:3], cam[3:6], cam[6:9], float(proj[0]), float(proj[1]), float(z["focal_dist"]), float(z["aperture"]))
    c.set_frame(80, 48, 4)
    return c


_NATIVE = r"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import test_gpu_exchange as T
rccl = C.CDLL("librccl.so")
class UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]
uid = UniqueId()
assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
comm = C.c_void_p()
rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
c = T._ctx()
c.reset(); c.render(0, 4, 3)
before = c.read_accum()
rc = c.L.ptk_gather_accum(c.h, comm, 0)
assert rc == 0, c.L.ptk_last_error(c.h)
after = c.read_accum()
assert np.array_equal(before, after) and before.any()
assert c.L.ptk_gather_accum(c.h, None, 0) != 0          # null communicator is an error, not a crash
rccl.ncclCommDestroy.argtypes = [C.c_void_p]
rccl.ncclCommDestroy(comm)
c.close()
print("native-rccl-ok")
"""


def test_native_rccl_gather_single_rank():
    """ptk_gather_accum over a single-rank ncclComm_t.  Runs in its own process: it talks to the system
    librccl directly, which must not share a process with the copy torch bundles."""
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, "-c", _NATIVE, ROOT], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "native-rccl-ok" in r.stdout, r.stdout + r.stderr


def test_torch_distributed_path_world_1():
    import torch
    import torch.distributed as dist
    from pbrpathtracer_amd.distributed import gather_accumulator
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29700 + os.getpid() % 200))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        c = _ctx()
        c.reset(); c.render(0, 5, 11)
        own = c.read_accum()                                  # internal accumulator
        accum = torch.zeros(48 * 80 * 3, dtype=torch.float32, device="cuda")
        c.set_stream(torch.cuda.current_stream().cuda_stream)
        c.bind_accum(accum.data_ptr())
        c.reset(); c.render(0, 5, 11)
        out = gather_accumulator(accum, dst=0)
        dist.barrier(); torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().reshape(48, 80, 3), own)
        # a second batch keeps accumulating in the bound buffer; overlapped exchange object as in bench.py
        from pbrpathtracer_amd.distributed import AccumulatorExchange
        ex = AccumulatorExchange(accum, dst=0)
        c.render(5, 3, 11)
        ex.start()
        c.render(8, 2, 11)                                    # next batch is enqueued while the exchange runs
        res = ex.wait(); torch.cuda.synchronize()
        c3 = _ctx(); c3.reset(); c3.render(0, 8, 11)
        assert np.array_equal(res.cpu().numpy().reshape(48, 80, 3), c3.read_accum())     # snapshot after 8 samples
        c3.close()
        c2 = _ctx(); c2.reset(); c2.render(0, 10, 11)
        assert np.array_equal(accum.cpu().numpy().reshape(48, 80, 3), c2.read_accum())
        # the packed form bench.py uses for N > 1 (index_select -> RCCL gather -> index_copy), forced in this one-rank
        # group: the root's result is the accumulator, through the same calls
        for mode in ("gather", "reduce"):
            exp = AccumulatorExchange(accum, dst=0, width=80, height=48, mode=mode, force=True)
            assert exp.mode == mode
            exp.start()
            got = exp.wait(); torch.cuda.synchronize()
            assert np.array_equal(got.cpu().numpy().reshape(48, 80, 3), c2.read_accum()), mode
        c.close(); c2.close()
    finally:
        dist.destroy_process_group()


def test_bench_two_ranks_rehearsal():
    """bench.py's N > 1 control flow (one process per rank, tile split, packed exchange, barrier + max-over-ranks timing,
    one JSON line from rank 0) with two ranks sharing this box's one GPU: RCCL refuses two ranks on one device, so the
    rehearsal backend (gloo, exchange through host copies) stands in; the gathered image must hold exactly what the two
    ranks hold together."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    port = 29600 + os.getpid() % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--config", "C1", "--backend", "gloo", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "gathered-image checksum OK" in r.stderr, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["parallelism"] == "tile-split x2" and "gather" in d["config"]["exchange"]
