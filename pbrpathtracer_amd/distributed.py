"""Multi-GPU exchange step: one process per GPU (torchrun), the frame is tile-split across ranks
(16x16 pixel tiles; tile (tx, ty) belongs to rank (ty*tiles_x + (tx + 3*ty) % tiles_x) % world, i.e.
round-robin with every tile row rotated by 3 so a rank's tiles form diagonals — include/ptk.h `ptk_set_tile`) and the float
accumulators are combined with ONE collective per batch of samples.

The reference has no distributed path at all; this is the MI355X-native addition the
north star asks for: an RCCL gather of the accumulation buffer over xGMI.  Every pixel is owned by exactly one
rank, so the root needs each rank's owned values and nothing else: `AccumulatorExchange` packs them (1/world of the
buffer) and gathers them point-to-point - the root's 7 xGMI links receive in parallel.  Because the other ranks hold
exact zeros at a pixel they do not own, a SUM `reduce` of the whole buffers gives the same image bit for bit; that
simpler form is `gather_accumulator()` (and `ptk_gather_accum` natively).

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


def owned_float_index(width: int, height: int, rank: int, world: int) -> np.ndarray:
    """Flat indices (int64) into the float RGB accumulator (rows BOTTOM-up, as `mTotalImg`) of the values rank
    `rank` owns - what the packed exchange sends instead of the whole buffer."""
    mask = tile_owner_mask(width, height, rank, world)[::-1]
    px = np.nonzero(mask.reshape(-1))[0].astype(np.int64)
    return (px[:, None] * 3 + np.arange(3, dtype=np.int64)[None, :]).reshape(-1)


class AccumulatorExchange:
    """The exchange step overlapped with rendering: `start()` snapshots the local accumulator and launches the
    collective on a side stream, so the exchange of batch k runs while the trace kernel of batch k+1 (which does not
    touch the accumulator) is already on the GPU; the render stream only waits for the device-to-device snapshot.
    `wait()` orders the render stream behind the last collective; `result` holds the gathered image on `dst`.  On
    CPU tensors (gloo) it is synchronous.

    With the frame geometry (`width`, `height`) the exchange is PACKED: every rank sends only the values of the
    tiles it owns (1/world of the buffer: 1.4 MB instead of 11 MB per rank at 720p on 8 GPUs) with one `gather`
    - point-to-point transfers that arrive at the root over its 7 xGMI links in parallel - and the root scatters
    them into place.  Pure copies, so the image is the single-GPU one bit for bit, like the sum-reduce of the
    zero-padded buffers that is used when no geometry is given (mode "reduce")."""

    def __init__(self, local, dst: int = 0, width: int = 0, height: int = 0, mode: str = "gather", force: bool = False):
        import torch
        import torch.distributed as dist
        self.local = local
        self.dst = dst
        self.result = torch.zeros_like(local)
        self.cuda = local.is_cuda
        # `force`: run the collective even in a one-rank group (rehearsal of the N > 1 code path on one GPU)
        self.multi = dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force)
        self.packed = bool(self.multi and mode == "gather" and width > 0 and height > 0)
        self.mode = "gather" if self.packed else "reduce"
        if self.packed:
            rank, world = dist.get_rank(), dist.get_world_size()
            idx = [owned_float_index(width, height, r, world) for r in range(world)]
            self.lens = [len(i) for i in idx]
            self.maxlen = max(self.lens)
            dev = local.device
            self.own_idx = torch.from_numpy(idx[rank]).to(dev)
            self.send = torch.zeros(self.maxlen, dtype=local.dtype, device=dev)
            if rank == dst:
                self.all_idx = [torch.from_numpy(i).to(dev) for i in idx]
                self.recv = [torch.zeros(self.maxlen, dtype=local.dtype, device=dev) for _ in range(world)]
            else:
                self.all_idx, self.recv = None, None
        if self.cuda:
            # high priority: its copy and the collective take wave slots ahead of the render stream's queued workgroups
            self.side = torch.cuda.Stream(device=local.device, priority=-1)
            self.rendered = torch.cuda.Event()
            self.copied = torch.cuda.Event()

    def _snapshot(self):
        if self.packed:
            import torch
            torch.index_select(self.local, 0, self.own_idx, out=self.send[: self.own_idx.numel()])
        else:
            self.result.copy_(self.local, non_blocking=True)

    def _collective(self):
        import torch.distributed as dist
        if not self.multi:
            return
        if self.packed:
            dist.gather(self.send, gather_list=self.recv, dst=self.dst)
            if self.recv is not None:
                for r, buf in enumerate(self.recv):
                    self.result.index_copy_(0, self.all_idx[r], buf[: self.lens[r]])
        else:
            dist.reduce(self.result, dst=self.dst, op=dist.ReduceOp.SUM)

    def start(self):
        import torch
        if not self.cuda:
            self._snapshot()
            self._collective()
            return
        main = torch.cuda.current_stream(self.local.device)
        self.rendered.record(main)
        with torch.cuda.stream(self.side):
            self.side.wait_event(self.rendered)
            self._snapshot()
            self.copied.record(self.side)
            self._collective()
        main.wait_event(self.copied)          # the next accumulate_kernel may overwrite `local` from here on

    def wait(self):
        import torch
        if self.cuda:
            torch.cuda.current_stream(self.local.device).wait_stream(self.side)
        return self.result
