#!/usr/bin/env python3
"""Cost-sorted lane assignment on the north-star frame, priced offline (scripts/sim/wave_sim.c sim_sorted): the pixels of a
32x32 super-tile dealt to its 16 waves in cost order against the shipped 8x8 tiles, in wave-level instructions.
  python scripts/sim/run_sorted_sim.py [--stride 8]"""
import argparse
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--W", type=int, default=3840)
    ap.add_argument("--H", type=int, default=2160)
    ap.add_argument("--stride", type=int, default=8)
    ap.add_argument("--iters", type=int, default=12)
    a = ap.parse_args()
    here = os.path.dirname(os.path.abspath(__file__))
    so = os.path.join(here, "_build", "libwave_sim.so")
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call(["gcc", "-O2", "-std=gnu99", "-fopenmp", "-mfma", "-mavx2", "-mf16c", "-ffp-contract=off", "-fno-fast-math",
                           "-shared", "-fPIC", "-Wno-unused-function", "-o", so, os.path.join(here, "wave_sim.c"), "-lm"])
    from raymarcher_amd import abi, scenes
    sim = C.CDLL(so)
    t = scenes.mandelbulb(a.W, a.H)
    s = abi.default_settings(fractalIters=a.iters)
    out = (C.c_double * 5)()
    cost = (C.c_double * 6)(148, 75, 156, 90, 60, 1700)
    st = sim.sim_sorted(C.byref(t.camera), t.objects, t.num_objects, t.lights, t.num_lights, C.byref(t.globals_), C.byref(s),
                        a.W, a.H, a.stride, C.c_float(1.15), cost, out, len(os.sched_getaffinity(0)))
    assert st == 0
    base = out[0]
    print(f"{int(out[4])} pixels in {int(out[4]) // 1024} super-tiles of 32x32 (every {a.stride}th); wave-level instructions relative to the shipped 8x8 tiles:")
    print(f"  sorted by the pixel's own cost (a perfect predictor): {out[1] / base:.3f}")
    print(f"  sorted by the cost of the pixel (2, 1) away:          {out[2] / base:.3f}")
    print(f"  sorted by the cost of the pixel (6, 3) away:          {out[3] / base:.3f}")


if __name__ == "__main__":
    main()
