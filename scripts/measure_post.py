#!/usr/bin/env python3
"""Time rm_post_process at 3840x2160 (torch events on the launch stream, 10 repetitions) for each pass combination.
Informative numbers for DESIGN.md / profiles/.  Usage: python scripts/measure_post.py [out.md]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from raymarcher_amd import Renderer, abi, scenes
    r = Renderer(0)
    W, H = 3840, 2160
    t = scenes.mandelbulb(W, H)
    frame, bright = r.render(t, abi.default_settings(fractalIters=12), W, H, bright=True)
    frame = frame * 1.4
    bright = torch.where((frame[..., :3] * torch.tensor([0.2126, 0.7152, 0.0722], device=frame.device)).sum(-1, keepdim=True) > 1.0,
                         frame, torch.zeros_like(frame))
    cases = [("gamma", dict(enableGammaCorrection=1), 16 + 16), ("HDR", dict(enableHDR=1, exposure=1.2), 16 + 16),
             ("FXAA", dict(enableFXAA=1), 16 + 4 + 4 + 16),
             ("bloom", dict(enableBloom=1), 16 + 8 + 9 * 16 + 16 + 8 + 16), ("bloom + HDR + FXAA", dict(enableBloom=1, enableHDR=1, enableFXAA=1, exposure=0.9),
                                                                            16 + 8 + 9 * 16 + 16 + 8 + 4 + 4 + 16)]
    rows = ["| passes | ms / 4K frame | algorithmic B/pixel | GB/s | of 8 TB/s |", "|---|---|---|---|---|"]
    for name, kw, bpp in cases:
        ps = abi.RmPostSettings(**{**dict(enableFXAA=0, enableGammaCorrection=0, enableHDR=0, enableBloom=0, exposure=1.0), **kw})
        r.post_process(frame, bright, ps)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            r.post_process(frame, bright, ps)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        gbs = W * H * bpp / (ms * 1e-3) / 1e9
        rows.append(f"| {name} | {ms:.3f} | {bpp} | {gbs:.0f} | {gbs / 80:.1f} % |")
        print(rows[-1], flush=True)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write("\n".join(rows) + "\n")


if __name__ == "__main__":
    main()
