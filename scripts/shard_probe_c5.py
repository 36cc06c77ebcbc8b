#!/usr/bin/env python3
"""One shard of an N-way sharded C5 frame (8K Menger sponge, 5 levels, 2 bounces) on one GPU, by either schedule — one stream and
three frames in flight on three streams (what bench.py's FramePipeline does): which schedule a 1/8 shard (4.15 M pixels, just below
the wavefront pipeline's 2^22-pixel threshold) should take.  GPU box only."""
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from raymarcher_amd import Renderer, Scene, abi, lib  # noqa: E402

r = Renderer(0)
L = lib()
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (7680, 4320)
t = Scene(path=os.path.join(ROOT, "tests", "golden", "scenes", "simple", "unit_mengersponge.json")).tables(W, H)
s = abi.default_settings(mengerLevels=5, numReflection=2, enableReflection=1)
for N in (8, 4, 2):
    for path in (1, 5):
        L.rm_set_kernel_path(path)
        rows = L.rm_shard_rows(H, 8, 0, N)
        outs = [torch.empty((rows, W, 4), dtype=torch.float32, device=r.device) for _ in range(3)]
        streams = [torch.cuda.Stream(device=r.device) for _ in range(3)]
        for o in outs[:2]:
            r.render_tiles(t, s, W, H, 8, 0, N, out=o)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(6):
            r.render_tiles(t, s, W, H, 8, 0, N, out=outs[0])
        torch.cuda.synchronize()
        one = (time.perf_counter() - t0) / 6 * 1e3
        ran = L.rm_debug_last_path()
        for i in range(6):
            with torch.cuda.stream(streams[i % 3]):
                r.render_tiles(t, s, W, H, 8, 0, N, out=outs[i % 3])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(12):
            with torch.cuda.stream(streams[i % 3]):
                r.render_tiles(t, s, W, H, 8, 0, N, out=outs[i % 3])
        torch.cuda.synchronize()
        three = (time.perf_counter() - t0) / 12 * 1e3
        print(f"C5 {W}x{H} shard 1/{N} ({rows * W / 1e6:.2f} Mpx) path {path} (ran {ran}): one stream {one:.2f} ms, three in flight {three:.2f} ms per frame", flush=True)
L.rm_set_kernel_path(0)
