#!/usr/bin/env python3
"""Stress of the light split's hand-over through memory: C2 (1080p, soft shadows + AO) with 1/4 of the tiles split, N frames on one
stream and on three streams at once, every frame compared with the plain render bit for bit.  GPU box only."""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from raymarcher_amd import Renderer, lib  # noqa: E402

r = Renderer(0)
L = lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
t, s, W, H, _ = bench.build_config("c2")
L.rm_debug_set_light_split(0)
ref = r.render(t, s, W, H).clone()
L.rm_debug_set_tile_shape(3)
for div in (4, 1, 32):
    L.rm_debug_set_light_split(div)
    out = torch.empty_like(ref)
    bad = split = 0
    for k in range(N):
        r.render(t, s, W, H, out=out)
        if k % 8 == 7 or k < 8:
            bad += int((out.view(torch.int32) != ref.view(torch.int32)).sum())
        split = max(split, L.rm_debug_last_split())
    streams = [torch.cuda.Stream(device=r.device) for _ in range(3)]
    outs = [torch.empty_like(ref) for _ in range(3)]
    for k in range(N):
        with torch.cuda.stream(streams[k % 3]):
            r.render(t, s, W, H, out=outs[k % 3])
        if k % 30 == 29:
            torch.cuda.synchronize()
            for o in outs:
                bad += int((o.view(torch.int32) != ref.view(torch.int32)).sum())
    torch.cuda.synchronize()
    print(f"c2 {W}x{H}, 1/{div} of the tiles split ({split} tiles): {N} frames on one stream + {N} on three: {bad} mismatched words", flush=True)
L.rm_debug_set_light_split(-1)
L.rm_debug_set_tile_shape(-1)
