#!/usr/bin/env python3
"""Lane utilisation of the generic kernel's march schedule on a scenefile (constant-cost evaluations): the shipped schedule
(lanes meet at every march of the code) against a per-lane queue (every lane runs its own marches back to back).
  python scripts/sim/run_generic_sim.py tests/golden/scenes/simple/unit_mengersponge.json 1920 1080 --levels 5 --bounces 2 --reflection"""
import argparse
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scene")
    ap.add_argument("W", type=int)
    ap.add_argument("H", type=int)
    ap.add_argument("--stride", type=int, default=4)
    ap.add_argument("--levels", type=int, default=4)
    ap.add_argument("--bounces", type=int, default=1)
    ap.add_argument("--reflection", action="store_true")
    ap.add_argument("--soft", action="store_true")
    ap.add_argument("--ao", action="store_true")
    a = ap.parse_args()
    here = os.path.dirname(os.path.abspath(__file__))
    so = os.path.join(here, "_build", "libwave_sim.so")
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call(["gcc", "-O2", "-std=gnu99", "-fopenmp", "-mfma", "-mavx2", "-mf16c", "-ffp-contract=off", "-fno-fast-math",
                           "-shared", "-fPIC", "-Wno-unused-function", "-o", so, os.path.join(here, "wave_sim.c"), "-lm"])
    from raymarcher_amd import Scene, abi
    sim = C.CDLL(so)
    t = Scene(path=a.scene).tables(a.W, a.H)
    s = abi.default_settings(mengerLevels=a.levels, numReflection=a.bounces, enableReflection=int(a.reflection),
                             enableSoftShadow=int(a.soft), enableAmbientOcclusion=int(a.ao))
    out = (C.c_double * 10)()
    st = sim.sim_generic(C.byref(t.camera), t.objects, t.num_objects, t.lights, t.num_lights, C.byref(t.globals_), C.byref(s),
                         a.W, a.H, a.stride, out, len(os.sched_getaffinity(0)))
    assert st == 0
    lane, ta, tb, nw, npx, tc, ta2, d1, d8, d16 = list(out)
    print(f"{os.path.basename(a.scene)} {a.W}x{a.H}: {lane / npx:.1f} evaluations per pixel; "
          f"shipped schedule {ta / nw:.0f} evaluation trips per wave (lane utilisation {lane / (64 * ta):.3f}); "
          f"per-lane queue {tb / nw:.0f} trips (utilisation {lane / (64 * tb):.3f}) — {tb / ta:.3f} of the shipped trips; "
          f"shadow rays of one shading point as a per-lane queue: {tc / ta:.3f}")
    print(f"  with the blocks between marches priced (surface block 3.3, light block 0.7 evaluations): shipped {(ta + ta2) / nw:.0f} per wave; "
          f"per-lane queue, surface block parked until T lanes wait: T=1 {d1 / (ta + ta2):.3f}, T=8 {d8 / (ta + ta2):.3f}, T=16 {d16 / (ta + ta2):.3f} of it")


if __name__ == "__main__":
    main()
