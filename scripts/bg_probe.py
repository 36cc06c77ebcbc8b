#!/usr/bin/env python3
"""How long does a 4K frame of pure background take (every ray misses the bounding ball)?  Separates the launch /
latency floor of cheap waves from the marching work.  GPU box only."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from raymarcher_amd import Renderer, abi, lib, scenes
    from raymarcher_amd.render import build_camera
    import math
    W, H = 3840, 2160
    r = Renderer(0)
    L = lib()
    t = scenes.mandelbulb(W, H)
    t.camera, _, _ = build_camera((0, 0, 4.5), (0, 0, 4.5), (0, 1, 0), math.radians(30.0), W, H)  # looking away
    s = abi.default_settings(fractalIters=12)
    out = torch.empty((H, W, 4), dtype=torch.float32, device=r.device)
    for wpb in os.environ.get("WPBS", "0").split(","):
        for _ in range(3):
            r.render(t, s, W, H, out=out)
        torch.cuda.synchronize()
        L.rm_set_timing(1)
        for _ in range(10):
            r.render(t, s, W, H, out=out)
        torch.cuda.synchronize()
        ms, k = C.c_double(), C.c_int()
        L.rm_get_timing(C.byref(ms), C.byref(k))
        L.rm_set_timing(0)
        print(f"all-background 4K frame: {ms.value:.3f} ms per launch ({W * H / 64 / ms.value / 1e3:.1f} waves/us), hit fraction {(out[..., 0] != 1).float().mean().item():.4f}")


if __name__ == "__main__":
    main()
