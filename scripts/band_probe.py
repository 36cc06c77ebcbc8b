#!/usr/bin/env python3
"""Would ONE big wavefront-pipeline frame finish sooner as B row bands in flight on B streams (their tails and HBM-bound / VALU-bound
kernels overlapping) than as one pipeline?  C5 (8K Menger, 2 bounces) whole vs 2 / 3 / 4 / 6 bands forked from and joined to one stream.
GPU box only."""
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from raymarcher_amd import Renderer, lib  # noqa: E402

r = Renderer(0)
L = lib()
cfg = sys.argv[1] if len(sys.argv) > 1 else "c5"
t, s, W, H, _ = bench.build_config(cfg)
out = torch.empty((H, W, 4), dtype=torch.float32, device=r.device)
for _ in range(2):
    r.render(t, s, W, H, out=out)
torch.cuda.synchronize()
ref = out.clone()
n = 6
t0 = time.perf_counter()
for _ in range(n):
    r.render(t, s, W, H, out=out)
torch.cuda.synchronize()
print(f"{cfg} {W}x{H} whole frame: {(time.perf_counter() - t0) / n * 1e3:.2f} ms (path {L.rm_debug_last_path()})", flush=True)
main = torch.cuda.current_stream(r.device)
for B in (2, 3, 4, 6):
    subs = [torch.cuda.Stream(device=r.device) for _ in range(B)]
    cut = [((H * b // B) + 7) // 8 * 8 for b in range(B)] + [H]

    def frame():
        fork = torch.cuda.Event()
        fork.record(main)
        for b in range(B):
            subs[b].wait_event(fork)
            with torch.cuda.stream(subs[b]):
                r.render(t, s, W, H, row_begin=cut[b], row_end=cut[b + 1], out=out[cut[b]:cut[b + 1]])
            e = torch.cuda.Event()
            e.record(subs[b])
            main.wait_event(e)

    out.zero_()
    for _ in range(2):
        frame()
    torch.cuda.synchronize()
    same = bool((out.view(torch.int32) == ref.view(torch.int32)).all())
    t0 = time.perf_counter()
    for _ in range(n):
        frame()
    torch.cuda.synchronize()
    print(f"  {B} bands in flight: {(time.perf_counter() - t0) / n * 1e3:.2f} ms, identical {same} (path {L.rm_debug_last_path()})", flush=True)
