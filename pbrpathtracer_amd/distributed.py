"""Multi-GPU exchange step: one process per GPU (torchrun), the frame is tile-split across ranks
(16x16 pixel tiles; tile (tx, ty) belongs to rank (ty*tiles_x + (tx + 3*ty) % tiles_x) % world, i.e.
round-robin with every tile row rotated by 3 so a rank's tiles form diagonals — include/ptk.h `ptk_set_tile`) and the float
accumulators are combined with ONE collective per batch of samples.

The reference has no distributed path at all; this is the MI355X-native addition the
north star asks for: an RCCL gather of the accumulation buffer over xGMI.  Because every pixel is
owned by exactly one rank and the others hold exact zeros there, a SUM reduce to the root reproduces
each owned value bit for bit, so the gather is expressed as a single `reduce` (one ring/tree pass
over 7 xGMI links instead of 7 point-to-point receives serialised at the root).

Works with backend "nccl" (= RCCL, device tensors) and, for CPU tests, "gloo".
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
    """Sum-reduce the per-rank accumulators (flat float32 tensors, non-owned pixels exactly 0) to
    rank `dst`.  `out` (same shape) receives the result on dst so the local accumulator can keep
    accumulating further samples; returns `out`."""
    import torch
    import torch.distributed as dist
    if out is None:
        out = torch.empty_like(local)
    out.copy_(local)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(out, dst=dst, op=dist.ReduceOp.SUM)
    return out


class AccumulatorExchange:
    """The exchange step overlapped with rendering: `start()` snapshots the local accumulator and
    launches the reduce on a side stream, so the collective of batch k runs while the trace kernel of
    batch k+1 (which does not touch the accumulator) is already on the GPU; the render stream only
    waits for the 11-25 MB device-to-device snapshot.  `wait()` orders the render stream behind the last
    collective; `result` holds the gathered image on `dst`.  On CPU tensors (gloo) it is synchronous."""

    def __init__(self, local, dst: int = 0):
        import torch
        self.local = local
        self.dst = dst
        self.result = torch.empty_like(local)
        self.cuda = local.is_cuda
        if self.cuda:
            # high priority: its copy and the collective take wave slots ahead of the render stream's queued workgroups
            self.side = torch.cuda.Stream(device=local.device, priority=-1)
            self.rendered = torch.cuda.Event()
            self.copied = torch.cuda.Event()

    def start(self):
        import torch
        import torch.distributed as dist
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if not self.cuda:
            self.result.copy_(self.local)
            if multi:
                dist.reduce(self.result, dst=self.dst, op=dist.ReduceOp.SUM)
            return
        main = torch.cuda.current_stream(self.local.device)
        self.rendered.record(main)
        with torch.cuda.stream(self.side):
            self.side.wait_event(self.rendered)
            self.result.copy_(self.local, non_blocking=True)
            self.copied.record(self.side)
            if multi:
                dist.reduce(self.result, dst=self.dst, op=dist.ReduceOp.SUM)
        main.wait_event(self.copied)          # the next accumulate_kernel may overwrite `local` from here on

    def wait(self):
        import torch
        if self.cuda:
            torch.cuda.current_stream(self.local.device).wait_stream(self.side)
        return self.result
