#!/usr/bin/env python3
"""Which frames are bound by the life of their heaviest waves rather than by work?  One frame after the other on one stream against
three frames in flight on three streams, ms per frame: BASELINE configurations and a few scenefiles with reflections on.  GPU box only."""
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from raymarcher_amd import Renderer, Scene, abi  # noqa: E402

r = Renderer(0)
cases = [(c,) + tuple(bench.build_config(c)[:4]) for c in ("c1", "c2", "c3", "c4")]
for name in ("reflections_complex", "reflections_basic", "test_reflectiveness", "point_light_2"):
    for W, H in ((1920, 1080),):
        t = Scene(path=os.path.join(ROOT, "tests", "golden", "scenes", "lighting", name + ".json")).tables(W, H, load_textures=False)
        for k in range(t.num_objects):
            t.objects[k].texLoc = -1
        cases.append((f"{name} {W}x{H}, reflection on, soft shadows + AO", t, abi.default_settings(enableReflection=1, enableSoftShadow=1, enableAmbientOcclusion=1), W, H))
for tag, t, s, W, H in cases:
    outs = [torch.empty((H, W, 4), dtype=torch.float32, device=r.device) for _ in range(3)]
    for _ in range(30):
        r.render(t, s, W, H, out=outs[0])
    torch.cuda.synchronize()
    n = 60
    t0 = time.perf_counter()
    for _ in range(n):
        r.render(t, s, W, H, out=outs[0])
    torch.cuda.synchronize()
    one = (time.perf_counter() - t0) / n * 1e3
    streams = [torch.cuda.Stream(device=r.device) for _ in range(3)]
    for k in range(90):
        with torch.cuda.stream(streams[k % 3]):
            r.render(t, s, W, H, out=outs[k % 3])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        with torch.cuda.stream(streams[k % 3]):
            r.render(t, s, W, H, out=outs[k % 3])
    torch.cuda.synchronize()
    three = (time.perf_counter() - t0) / n * 1e3
    print(f"{tag}: one stream {one:.3f} ms per frame, three in flight {three:.3f} ms ({one / three:.2f} x)", flush=True)
