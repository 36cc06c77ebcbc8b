#!/usr/bin/env python3
"""Occupancy timeline of the 4K bulb kernel from per-wave s_memrealtime stamps (diagnostic build): how many waves are
resident over the kernel's life, and how long the tail is.  GPU box only.  Usage: python scripts/wave_timeline.py [out.md]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    from raymarcher_amd import Renderer, abi, scenes
    W, H = 3840, 2160
    r = Renderer(0)
    t = scenes.mandelbulb(W, H)
    s = abi.default_settings(fractalIters=12)
    for _ in range(5):
        r.render(t, s, W, H)
    _, mhz, spans = r.render_clocked(t, s, W, H, wave_spans=True)
    sp = spans.cpu().numpy()
    sp = sp[sp[:, 1] > 0]
    t0, t1 = sp[:, 0].min(), sp[:, 1].max()
    dur = (t1 - t0) / 100.0  # µs
    lines = [f"4K bulb frame, stamped build: {len(sp)} waves, kernel span {dur:.0f} us, shader clock {mhz:.0f} MHz",
             f"wave life: mean {(sp[:, 1] - sp[:, 0]).mean() / 100:.1f} us, median {np.median(sp[:, 1] - sp[:, 0]) / 100:.1f} us, "
             f"max {(sp[:, 1] - sp[:, 0]).max() / 100:.0f} us; sum of lives / span = {(sp[:, 1] - sp[:, 0]).sum() / (t1 - t0):.0f} waves resident on average (5120 slots at 5 waves/SIMD)",
             "", "| time (% of span) | resident waves |", "|---|---|"]
    edges = np.linspace(t0, t1, 21)
    for a, b in zip(edges[:-1], edges[1:]):
        mid = (a + b) / 2
        lines.append(f"| {100 * (mid - t0) / (t1 - t0):.0f} | {int(((sp[:, 0] <= mid) & (sp[:, 1] > mid)).sum())} |")
    last_start = sp[:, 0].max()
    lines.append("")
    lines.append(f"last wave starts at {100 * (last_start - t0) / (t1 - t0):.1f} % of the span; "
                 f"time with < 2560 resident waves: {100 * sum(1 for m in np.linspace(t0, t1, 400) if ((sp[:, 0] <= m) & (sp[:, 1] > m)).sum() < 2560) / 400:.1f} % of the span")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(out + "\n")


if __name__ == "__main__":
    main()
