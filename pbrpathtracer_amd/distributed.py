"""Multi-GPU exchange step: one process per GPU, the frame is tile-split across ranks (16x16 pixel tiles; tile
(tx, ty) belongs to rank (ty*tiles_x + (tx + 3*ty) % tiles_x) % world, i.e. round-robin with every tile row rotated by 3
so a rank's tiles form diagonals — include/ptk.h `ptk_set_tile`) and the float accumulators are combined by a PACKED
GATHER of every rank's owned tiles to the root.

The reference has no distributed path at all; this is the MI355X-native addition the north star asks
for: an RCCL gather of the accumulation buffer over xGMI.  The product path is native — `ptk_gather_accum`
(pbrpathtracer_amd/csrc/ptk_api.hip): pack kernel -> grouped ncclSend / ncclRecv on the library's own communicator ->
unpack kernel on the root, all on the context's high-priority exchange stream; torch is not involved.  `NativeExchange`
is its thin Python handle.

`HostPackedExchange` runs the same exchange on CPU tensors over a `torch.distributed` group (gloo): it exists for the
world-size-2/3 CPU tests and for bench.py's one-GPU rehearsal of the N > 1 control flow (RCCL refuses two ranks on one
device).  Its packing order is NOT restated here: it comes from the library's host-only `ptk_packed_layout`, the
function the device kernels are tested against, so the CPU tests pin the native order.
"""
from __future__ import annotations

import numpy as np

TILE = 16


def tile_owner_mask(width: int, height: int, rank: int, world: int) -> np.ndarray:
    """Boolean [H, W] mask (rows top-down) of the pixels rank `rank` renders."""
    tiles_x = (width + TILE - 1) // TILE
    ty, tx = np.meshgrid(np.arange(height) // TILE, np.arange(width) // TILE, indexing="ij")
    return ((ty * tiles_x + (tx + 3 * ty) % tiles_x) % world) == rank


def owned_tile_count(width: int, height: int, rank: int, world: int) -> int:
    n = ((width + TILE - 1) // TILE) * ((height + TILE - 1) // TILE)
    return 0 if n <= rank else (n - rank + world - 1) // world


def gather_accumulator(local, out=None, dst: int = 0):
    """The simplest correct exchange, kept as the comparator of the packed form: SUM-reduce the per-rank accumulators
    (flat float32 tensors, non-owned pixels exactly 0, so the sum is a gather) to rank `dst`; returns `out`."""
    import torch
    import torch.distributed as dist
    if out is None:
        out = torch.empty_like(local)
    out.copy_(local)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(out, dst=dst, op=dist.ReduceOp.SUM)
    return out


class NativeExchange:
    """Handle of the library's RCCL exchange (`ptk_gather_accum`).  `start()` queues the exchange of the accumulator as
    it stands after the renders issued so far and returns at once — the next render's trace kernel overlaps the
    transfer; `wait()` blocks the host until the last exchange has landed; `result()` reads the gathered image on the
    root ([H, W, 3] float32, rows bottom-up)."""

    mode = "gather"

    def __init__(self, ctx, root: int = 0):
        self.ctx, self.root = ctx, root

    @staticmethod
    def init_communicator(ctx, rank: int, world: int, broadcast_bytes):
        """Rank 0 makes the RCCL unique id, `broadcast_bytes(b | None) -> bytes` carries it to the other ranks (any
        rendezvous: bench.py uses torch.distributed's object broadcast), every rank joins."""
        from . import ptk
        uid = broadcast_bytes(ptk.comm_unique_id() if rank == 0 else None)
        ctx.comm_init(uid, rank, world)

    def start(self):
        self.ctx.gather_accum(self.root)

    def wait(self):
        self.ctx.gather_wait()

    def result(self):
        return self.ctx.read_gathered()


class HostPackedExchange:
    """The packed gather on CPU tensors over gloo, in the library's packing order (`ptk_packed_layout`): every rank
    sends the 768-float blocks of its owned tiles, the root scatters them into a full image.  Synchronous."""

    mode = "gather"

    def __init__(self, local, width: int, height: int, dst: int = 0):
        import torch
        import torch.distributed as dist
        from . import ptk
        self.local, self.dst = local, dst
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.layouts = [ptk.packed_layout(width, height, r, self.world) for r in range(self.world)]
        self.maxlen = max(len(x) for x in self.layouts)
        mine = self.layouts[self.rank]
        self.valid = torch.from_numpy(np.nonzero(mine >= 0)[0])
        self.src = torch.from_numpy(mine[mine >= 0])
        self.send = torch.zeros(self.maxlen, dtype=local.dtype)
        self.result_tensor = torch.zeros_like(local) if self.rank == dst else None
        self.recv = [torch.zeros(self.maxlen, dtype=local.dtype) for _ in range(self.world)] if self.rank == dst else None

    def start(self):
        import torch
        import torch.distributed as dist
        self.send.zero_()
        self.send[self.valid] = self.local[self.src]
        dist.gather(self.send, gather_list=self.recv, dst=self.dst)
        if self.recv is not None:
            for r, buf in enumerate(self.recv):
                lay = self.layouts[r]
                keep = np.nonzero(lay >= 0)[0]
                self.result_tensor[torch.from_numpy(lay[keep])] = buf[torch.from_numpy(keep)]

    def wait(self):
        return self.result_tensor

    def result(self):
        return self.result_tensor
