"""Row-tile sharding of one frame over the GPUs of a node: one process per GPU, no data-path collective while
rendering, one gather of the packed tiles to rank 0 (RCCL over xGMI with the "nccl" backend; "gloo" in the CPU
tests).  The reference renders whole frames on one GPU; this layer has no counterpart there.

Partition: the frame is cut into tiles of `tile_rows` rows, rank r owns tiles t ≡ r (mod world) — the bulb sits in
the middle of the image, contiguous bands would give the middle GPUs several times the work of the outer ones.
"""
from dataclasses import dataclass

from ._lib import lib


@dataclass(frozen=True)
class ShardPlan:
    H: int
    tile_rows: int
    world: int

    def rows(self, rank):
        """Rows rank `rank` renders (rm_shard_rows)."""
        return lib().rm_shard_rows(self.H, self.tile_rows, rank, self.world)

    @property
    def slot_rows(self):
        """Rows of one gather slot: the largest shard's (shard 0, or shard 1 under rm_set_root_relief), so every slot holds any shard."""
        return max(self.rows(k) for k in range(min(self.world, 2)))

    def frame_rows(self, rank):
        """Frame row of each packed row of `rank` (rm_shard_row_to_frame)."""
        L = lib()
        return [L.rm_shard_row_to_frame(self.H, self.tile_rows, rank, self.world, r) for r in range(self.rows(rank))]


def gather_to_root(local_slot, plan, rank, gathered=None):
    """dist.gather of equal-sized slots to rank 0.  `local_slot`: (slot_rows, W, 4) tensor whose first
    plan.rows(rank) rows are valid.  Returns the (world·slot_rows, W, 4) tensor on rank 0, None elsewhere."""
    import torch
    import torch.distributed as dist
    if rank == 0:
        if gathered is None:
            gathered = torch.empty((plan.world * plan.slot_rows,) + tuple(local_slot.shape[1:]), dtype=local_slot.dtype,
                                   device=local_slot.device)
        dist.gather(local_slot, list(gathered.view(plan.world, plan.slot_rows, *local_slot.shape[1:]).unbind(0)), dst=0)
        return gathered
    dist.gather(local_slot, None, dst=0)
    return None


class FramePipeline:
    """Frames are independent, so several are in flight: `depth` slots (local tiles + gather buffer each); the gather of
    frame i (xGMI) is issued asynchronously and runs under the render of frame i+1 (CUs), and it is only joined `depth − 1`
    submits later.

        for i in range(K): pipe.submit(render_into)        # render_into(slot_tensor) renders this rank's tiles
        pipe.drain()

    `finish(gathered_slots)` runs on rank 0 once per frame (de-interleave); everything is stream-ordered on the
    device (Work.wait() on the "nccl" backend blocks the current stream, not the host).

    multi_stream=True gives every slot its own CUDA stream, so consecutive frames' RENDER kernels overlap too.  That is what
    lets a sharded frame scale: a frame holds a few rays that march all 256 steps without converging — a serial chain of
    ≈0.7–0.9 ms — and on one stream a 1/8-frame shard (0.3 ms of work) cannot end before its longest chain does
    (measured: 0.95 ms per shard-frame on one stream, 0.27–0.31 ms with three frames in flight on three streams;
    scripts/shard_probe.py).  The library keeps scratch and the tile-order feedback per stream, so the slots do not interfere."""

    def __init__(self, plan, rank, slot_shape, dtype, device, finish=None, depth=2, multi_stream=False, rotate_root=False):
        """rotate_root=False: every frame is gathered to rank 0 (what a host that shows or saves frames from one
        process wants).  rotate_root=True: frame i is gathered to rank i mod world and `finish` runs there — every frame still
        ends as ONE complete frame on ONE GPU, but the root's extra work (receiving world − 1 slots over its xGMI links, the
        de-interleave pass over the whole frame) is spread over the ranks instead of making rank 0 the longest pole of every
        frame (profiles/r04_i_submit_rate.md: rank 0 bounds the headline's N = 8 scaling at ≈5.9 ×)."""
        import torch
        assert depth >= 2
        self.plan, self.rank, self.finish, self.depth, self.rotate = plan, rank, finish, depth, bool(rotate_root)
        self.local = [torch.zeros((plan.slot_rows,) + tuple(slot_shape), dtype=dtype, device=device) for _ in range(depth)]
        self.gathered = [torch.empty((plan.world * plan.slot_rows,) + tuple(slot_shape), dtype=dtype, device=device)
                         for _ in range(depth)] if (rank == 0 or rotate_root) else [None] * depth
        self.root_of = [0] * depth  # the root of the frame each slot holds
        # the per-rank views of every gather buffer, made once (submit() runs several thousand times a second at N = 8)
        self.outs = [list(g.view(plan.world, plan.slot_rows, *tuple(slot_shape)).unbind(0)) if g is not None else None for g in self.gathered]
        self.work = [None] * depth
        self.streams = None
        if multi_stream and torch.device(device).type == "cuda":
            self.streams = [torch.cuda.Stream(device=device) for _ in range(depth)]
            for st in self.streams:
                st.wait_stream(torch.cuda.current_stream(device))  # the buffers above were filled on the current stream
        self.i = 0
        self.frames_finished = 0

    def _on(self, b):
        import contextlib
        import torch
        return torch.cuda.stream(self.streams[b]) if self.streams is not None else contextlib.nullcontext()

    def _join(self, b):
        if self.work[b] is None:
            return
        with self._on(b):
            self.work[b].wait()
            self.work[b] = None
            self.frames_finished += 1
            if self.rank == self.root_of[b] and self.finish is not None:
                self.finish(self.gathered[b])

    def submit(self, render_into):
        import torch.distributed as dist
        b = self.i % self.depth
        self._join(b)                       # frame i-depth used this slot (normally already joined below)
        root = (self.i % self.plan.world) if self.rotate else 0
        self.root_of[b] = root
        with self._on(b):
            render_into(self.local[b])
            self.work[b] = dist.gather(self.local[b], self.outs[b] if self.rank == root else None, dst=root, async_op=True)
        if self.i >= self.depth - 1:        # finish the oldest frame in flight while the newer ones render / travel
            self._join((self.i - (self.depth - 1)) % self.depth)
        self.i += 1

    def drain(self):
        for k in range(self.depth):         # oldest first
            self._join((self.i + k) % self.depth)
        if self.streams is not None:
            import torch
            for st in self.streams:
                torch.cuda.current_stream(st.device).wait_stream(st)


def deinterleave_host(gathered, plan):
    """Index-map de-interleave for host tensors (the GPU path is rm_deinterleave)."""
    import torch
    frame = torch.empty((plan.H,) + tuple(gathered.shape[1:]), dtype=gathered.dtype)
    for k in range(plan.world):
        rows = plan.frame_rows(k)
        frame[torch.tensor(rows, dtype=torch.long)] = gathered[k * plan.slot_rows:k * plan.slot_rows + len(rows)]
    return frame
