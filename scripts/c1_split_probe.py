#!/usr/bin/env python3
"""The light split on the sampler class: unit_sphere.json (C1's scene: sphere over a textured floor, three spot lights) at several
sizes, kernel ms of the settled frame with the split off / forced at a few fractions / measured by the launcher.  GPU box only."""
import ctypes as C
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from raymarcher_amd import Renderer, Scene, abi, lib  # noqa: E402

r = Renderer(0)
L = lib()
for W, H, steps in ((256, 256, 64), (1920, 1080, 256), (3840, 2160, 256)):
    t = Scene(path=os.path.join(ROOT, "tests", "golden", "scenes", "simple", "unit_sphere.json")).tables(W, H)
    s = abi.default_settings(maxSteps=steps)
    out = torch.empty((H, W, 4), dtype=torch.float32, device=r.device)
    row = []
    for div in (0, 4, 64, 256, -1):
        L.rm_debug_set_light_split(div)
        for _ in range(30):
            r.render(t, s, W, H, out=out)
        torch.cuda.synchronize()
        L.rm_set_timing(1)
        for _ in range(100):
            r.render(t, s, W, H, out=out)
        torch.cuda.synchronize()
        k, n = C.c_double(), C.c_int()
        L.rm_get_timing(C.byref(k), C.byref(n))
        L.rm_set_timing(0)
        row.append(f"{'measured' if div < 0 else ('off' if div == 0 else '1/' + str(div))}: {k.value:.4f} ms ({L.rm_debug_last_split()} tiles)")
    print(f"unit_sphere.json {W}x{H}, {steps} steps: " + "; ".join(row), flush=True)
L.rm_debug_set_light_split(-1)
