#!/usr/bin/env python3
"""the self-balancing bulb pipelines (kernel paths 2 / 3 / 4, rm_bulb_pipeline.hip.h; last timed in
round 1) on the CURRENT build against the one-lane-per-pixel kernel (path 1) where a schedule that needs no history could win:
the cold frame (raster tile order: rm_set_tile_order(0)) and the orbiting sequence (1 degree per frame).  Same frame, same
bits (checked).  Usage: python scripts/measure_bulb_paths.py [out.md]"""
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from raymarcher_amd import Renderer, abi, lib, scenes
    from raymarcher_amd.render import build_camera
    r, L = Renderer(0), lib()
    W, H = 3840, 2160
    s = abi.default_settings(fractalIters=12)
    t = scenes.mandelbulb(W, H)
    out = torch.empty((H, W, 4), dtype=torch.float32, device=r.device)
    ref = None

    def wall(render, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            render(i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    frames = 24
    tabs = []
    for i in range(frames):
        a = math.radians(1.0 * i)
        pos = (4.5 * math.sin(a), 0.0, 4.5 * math.cos(a))
        ti = scenes.mandelbulb(W, H)
        ti.camera = build_camera(pos, tuple(-c for c in pos), (0.0, 1.0, 0.0), math.radians(30.0), W, H)[0]
        tabs.append(ti)
    rows = ["| schedule | static frame ms | cold frame ms (no history) | orbit 1 deg/frame ms | identical bits |", "|---|---|---|---|---|"]
    for path, order, label in ((1, 1, "path 1 render_kernel, tile-order feedback"), (1, 0, "path 1 render_kernel, raster order"),
                               (2, 0, "path 2 pipeline A (state machines + lane refill)"), (3, 0, "path 3 pipeline B (compacted lists)"),
                               (4, 0, "path 4 pipeline C (step-budgeted passes)")):
        L.rm_set_kernel_path(path)
        L.rm_set_tile_order(order)
        for _ in range(3):
            r.render(t, s, W, H, out=out)
        assert L.rm_debug_last_path() == path
        static = wall(lambda i: r.render(t, s, W, H, out=out), 20)
        same = True
        if ref is None:
            ref = out.clone()
        else:
            same = bool(torch.equal(out.view(torch.int32), ref.view(torch.int32)))
        # cold: every frame is a first frame — alternate two frame sizes so the feedback (if on) never has history
        if path == 1 and order == 1:
            small = torch.empty((H - 8, W, 4), dtype=torch.float32, device=r.device)
            cold = wall(lambda i: (r.render(t, s, W, H, row_begin=0, row_end=H - 8, out=small), r.render(t, s, W, H, out=out)), 10)
            cold_note = f"{cold / 2:.3f} (alternating sizes)"
        else:
            cold_note = f"{static:.3f}"
        r.render(tabs[0], s, W, H, out=out)
        orbit = wall(lambda i: r.render(tabs[i % frames], s, W, H, out=out), frames)
        rows.append(f"| {label} | {static:.3f} | {cold_note} | {orbit:.3f} | {same} |")
        print(rows[-1], flush=True)
    L.rm_set_kernel_path(0)
    L.rm_set_tile_order(-1)
    md = "\n".join(rows) + "\n"
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            f.write(md)


if __name__ == "__main__":
    main()
