#!/usr/bin/env python3
"""Per-wave life spans of a table-walk frame (stamped diagnostic build, rm_render_clocked): is the kernel as long as its longest
wave?  GPU box only.  Usage: python scripts/wave_lives.py [scenefile relative to tests/golden/scenes] [W H]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import numpy as np
    from raymarcher_amd import Renderer, Scene, abi
    rel = sys.argv[1] if len(sys.argv) > 1 else "lighting/directional_light_2.json"
    W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
    r = Renderer(0)
    t = Scene(path=os.path.join(ROOT, "tests", "golden", "scenes", rel)).tables(W, H, load_textures=False)
    for i in range(t.num_objects):
        t.objects[i].texLoc = -1
    s = abi.default_settings(enableSoftShadow=1, enableAmbientOcclusion=1)
    for _ in range(3):
        r.render(t, s, W, H)
    _, mhz, spans = r.render_clocked(t, s, W, H, wave_spans=True)
    sp = spans.cpu().numpy()
    idx = np.nonzero(sp[:, 1] > 0)[0]
    sp = sp[idx]
    t0, t1 = sp[:, 0].min(), sp[:, 1].max()
    life = (sp[:, 1] - sp[:, 0]) / 100.0
    print(f"{rel} {W}x{H}: {len(sp)} waves, kernel span {(t1 - t0) / 100:.0f} us (raster order, stamped build), shader clock {mhz:.0f} MHz")
    print(f"wave life: mean {life.mean():.1f} us, median {np.median(life):.1f}, p99 {np.percentile(life, 99):.0f}, max {life.max():.0f} us; "
          f"resident on average {life.sum() / ((t1 - t0) / 100):.0f} waves of 6144 slots")
    tilesX = (W + 7) // 8
    order = np.argsort(-life)[:12]
    print("longest waves: life us, start (% of span), tile x, tile y")
    for k in order:
        w = idx[k]
        print(f"  {life[k]:7.0f}  {100 * (sp[k, 0] - t0) / (t1 - t0):5.1f}  {w % tilesX:4d} {w // tilesX:4d}")
    edges = np.linspace(t0, t1, 11)
    print("resident waves at 5 %, 15 %, … of the span:", [int(((sp[:, 0] <= m) & (sp[:, 1] > m)).sum()) for m in (edges[:-1] + edges[1:]) / 2])


if __name__ == "__main__":
    main()
