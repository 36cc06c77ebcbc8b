"""Multi-process path on CPU: world_size-2 (and 3) gloo groups run the same shard → gather → de-interleave logic
bench.py uses with RCCL, with the oracle standing in for the GPU render (tests may use the oracle)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
import helpers as h
from raymarcher_amd import abi
from raymarcher_amd.dist import ShardPlan, gather_to_root, deinterleave_host
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
W, H, T = 40, 50, 8            # last tile is partial
from raymarcher_amd import lib
lib().rm_set_root_relief(int(os.environ.get("RM_TEST_RELIEF", "0")))   # every rank the same partition
plan = ShardPlan(H, T, world)
scene = h.scene_mandelbulb(W, H)
s = abi.default_settings(fractalIters=12)
rows = plan.frame_rows(rank)
slot = torch.zeros((plan.slot_rows, W, 4), dtype=torch.float32)
for i, y in enumerate(rows):     # render this rank's rows with the oracle (one call per contiguous tile would also do)
    slot[i] = torch.from_numpy(h.oracle_render(scene, s, W, H, y, y + 1, threads=1)[0])
g = gather_to_root(slot, plan, rank)
ok = True
if rank == 0:
    frame = deinterleave_host(g, plan).numpy()
    ref = h.oracle_render(scene, s, W, H, threads=2)
    ok = bool((frame.view(np.uint32) == ref.view(np.uint32)).all())
    assert sum(plan.rows(k) for k in range(world)) == H
flag = torch.tensor([1 if ok else 0])
dist.broadcast(flag, src=0)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if int(flag.item()) == 1 else 3)
'''


@pytest.mark.parametrize("world,relief", [(2, 0), (3, 0), (3, 2), (2, 3)])
def test_shard_gather_deinterleave_gloo(world, relief, tmp_path):
    """… also under rm_set_root_relief (rank 0 owns fewer tiles: the largest slot is rank 1's)."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    port = 29400 + world + 7 * relief + (os.getpid() % 500)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)]
    env = dict(os.environ, OMP_NUM_THREADS="2", RM_TEST_RELIEF=str(relief))
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]


PIPE_WORKER = r'''
import os, sys
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
from raymarcher_amd.dist import FramePipeline, ShardPlan, deinterleave_host
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
W, H, T, K = 6, 53, 8, 5
from raymarcher_amd import lib
lib().rm_set_root_relief(int(os.environ.get("RM_TEST_RELIEF", "0")))
plan = ShardPlan(H, T, world)
rows = torch.tensor(plan.frame_rows(rank), dtype=torch.float32)
frames = []
DEPTH = int(os.environ.get("RM_TEST_DEPTH", "2"))
DT = torch.uint8 if os.environ.get("RM_TEST_DTYPE") == "uint8" else torch.float32   # bench.py --gather rgba8 moves bytes
ROTATE = os.environ.get("RM_TEST_ROTATE") == "1"   # frame k is gathered to rank k mod world instead of rank 0
pipe = FramePipeline(plan, rank, (W, 4), DT, torch.device("cpu"), depth=DEPTH, rotate_root=ROTATE,
                     finish=lambda g: frames.append(deinterleave_host(g, plan).clone()))
for k in range(K):
    def render_into(slot, k=k):   # pixel value = 1000·frame + frame row: any mix-up of slots, frames or rows shows
        slot[:len(rows)] = ((1000.0 if DT is torch.float32 else 50.0) * k + rows)[:, None, None].expand(len(rows), W, 4).to(DT)
    pipe.submit(render_into)
    assert pipe.frames_finished == max(k - (DEPTH - 2), 0), (k, pipe.frames_finished)   # frame k-(DEPTH-1) is joined inside submit(k)
pipe.drain()
ok = pipe.frames_finished == K
mine = [k for k in range(K) if (k % world if ROTATE else 0) == rank]   # the frames this rank is the root of, in order
ok = ok and len(frames) == len(mine)
for k, f in zip(mine, frames):
    want = ((1000.0 if DT is torch.float32 else 50.0) * k + torch.arange(H, dtype=torch.float32))[:, None, None].expand(H, W, 4).to(DT)
    ok = ok and f.dtype == DT and bool((f == want).all())
flag = torch.tensor([1 if ok else 0])
dist.all_reduce(flag, op=dist.ReduceOp.MIN)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if int(flag.item()) == 1 else 3)
'''


@pytest.mark.parametrize("world,depth,dtype,rotate", [(2, 2, "float32", 0), (4, 2, "float32", 0), (2, 3, "float32", 0), (3, 4, "float32", 0),
                                                      (2, 3, "uint8", 0), (3, 3, "float32", 1), (4, 2, "float32", 1), (2, 3, "uint8", 1),
                                                      (3, 3, "float32", 2)])
def test_frame_pipeline_gloo(world, depth, dtype, rotate, tmp_path):
    """The pipelined gather bench.py uses for N > 1 (`depth` frames in flight: frame i's gather under the renders of the
    frames after it): every frame arrives complete, in order, through the right slot — on rank 0, or with rotate_root on rank
    i mod world for frame i."""
    script = tmp_path / "pipe_worker.py"
    script.write_text(PIPE_WORKER.format(root=ROOT))
    port = 29900 + world + 10 * depth + 50 * rotate + (os.getpid() % 400)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)]
    p = subprocess.run(cmd, env=dict(os.environ, OMP_NUM_THREADS="1", RM_TEST_DEPTH=str(depth), RM_TEST_DTYPE=dtype, RM_TEST_ROTATE=str(rotate & 1), RM_TEST_RELIEF="2" if rotate == 2 else "0"), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
