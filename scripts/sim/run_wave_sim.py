#!/usr/bin/env python3
"""Build and run scripts/sim/wave_sim.c on the north-star frame (3840x2160 Mandelbulb, 12 iterations): prints, per wave
schedule, the predicted wave-level VALU instruction count relative to the shipped one and the lane utilisation.

  python scripts/sim/run_wave_sim.py [--stride 16] [--cull 2.1] [--W 3840 --H 2160]

CPU only (uses the oracle); the numbers steer which schedule is worth building and measuring on the GPU."""
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
    ap.add_argument("--stride", type=int, default=16, help="simulate every stride-th workgroup")
    ap.add_argument("--cull", type=float, default=2.1, help="object-space cull radius (0 = none)")
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--cost", type=float, nargs=6, default=[148, 75, 156, 90, 60, 1700],
                    metavar=("cIt", "cEv", "cItF", "cEvF", "cRay", "cHit"))
    args = ap.parse_args()
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(here, "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "libwave_sim.so")
    subprocess.check_call(["gcc", "-O2", "-std=gnu99", "-fopenmp", "-mfma", "-mavx2", "-mf16c", "-ffp-contract=off", "-fno-fast-math",
                           "-shared", "-fPIC", "-Wno-unused-function", "-o", so, os.path.join(here, "wave_sim.c"), "-lm"])
    from raymarcher_amd import abi, scenes
    sim = C.CDLL(so)
    sim.sim_schedule_name.restype = C.c_char_p
    n = sim.sim_num_schedules()
    t = scenes.mandelbulb(args.W, args.H)
    s = abi.default_settings(fractalIters=args.iters)
    res = (C.c_double * (5 * n + 5))()
    cost = (C.c_double * 6)(*args.cost)
    st = sim.sim_run(C.byref(t.camera), t.objects, t.num_objects, t.lights, t.num_lights, C.byref(t.globals_), C.byref(s),
                     args.W, args.H, args.stride, C.c_float(args.cull), cost, res, len(os.sched_getaffinity(0)))
    assert st == 0, st
    r = list(res)
    wave, lane, prim, norm, shad = (r[i * n:(i + 1) * n] for i in range(5))
    pixels, hits, rays, evals, iters = r[5 * n:]
    print(f"frame {args.W}x{args.H}, every {args.stride}th workgroup: {int(pixels)} px, hit {hits / pixels:.3f}, "
          f"rays/hit {rays / max(hits, 1):.2f}, evals/px {evals / pixels:.1f}, iters/px {iters / pixels:.1f}, cull R {args.cull}")
    print(f"cost model (wave instr): iteration {args.cost[0]}, eval overhead {args.cost[1]}, flattened {args.cost[2]}/{args.cost[3]}, "
          f"ray setup {args.cost[4]}, per-hit-wave shading {args.cost[5]}")
    print(f"shipped schedule: {wave[0] / pixels * 64:.0f} wave-instructions·64 per pixel ({lane[0] / pixels:.0f} useful lane-instructions per pixel)")
    print(f"{'schedule':58s} {'rel.instr':>9s} {'util':>6s}   primary  normals+shade  shadow (rel. to shipped total)")
    for i in range(n):
        print(f"{sim.sim_schedule_name(i).decode():58s} {wave[i] / wave[0]:9.3f} {lane[i] / (wave[i] * 64):6.3f}   "
              f"{prim[i] / wave[0]:7.3f}  {norm[i] / wave[0]:13.3f}  {shad[i] / wave[0]:6.3f}")


if __name__ == "__main__":
    main()
