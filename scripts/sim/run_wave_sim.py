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


    # tail estimate: list-schedule the waves, in dispatch order, onto the chip's wave slots (every slot runs at the same
    # rate: 1/occ of a SIMD); compare with the perfectly balanced time
    import numpy as np
    import heapq
    cost = (C.c_float * (1 << 18))()
    idx = (C.c_int * (1 << 18))()
    est = (C.c_float * (1 << 18))()
    m = sim.sim_wave_costs(cost, idx, est, 1 << 18)
    cst, ix, es = np.array(cost[:m], dtype=np.float64), np.array(idx[:m]), np.array(est[:m], dtype=np.float64)
    ba, bs = (C.c_double * 65)(), (C.c_double * 257)()
    sim.sim_histograms(ba, bs)
    ba, bs = np.array(ba), np.array(bs)
    tot = ba.sum()
    print("march cost of the shipped schedule by number of lanes still marching (share of all march cost):")
    for lo, hi in ((1, 1), (2, 4), (5, 8), (9, 16), (17, 32), (33, 48), (49, 64)):
        print(f"  {lo:2d}-{hi:2d} lanes: {ba[lo:hi + 1].sum() / tot:.3f}")
    print("march cost by step index: " + ", ".join(f"steps {a}-{b}: {bs[a:b + 1].sum() / tot:.3f}" for a, b in ((0, 15), (16, 31), (32, 63), (64, 127), (128, 256))))
    dout = (C.c_double * 30)()
    sim.sim_defer_results(dout)
    dd = np.array(dout).reshape(5, 6)
    print("tail deferral (a march with <= T live lanes hands its rest to a dense second kernel), relative to the shipped total:")
    for t, row in zip((0, 4, 8, 12, 16), dd):
        print(f"  T={t:2d}: main kernel {row[0] / wave[0]:.3f} (heaviest wave {row[1]:.0f}), tail kernels {row[2] / wave[0]:.3f}, sum {(row[0] + row[2]) / wave[0]:.3f}; "
              f"deferred primary rays/px {row[3] / pixels:.4f}, shadow rays/px {row[4] / pixels:.4f}")
    un = (C.c_double * 4)()
    sim.sim_unified(un)
    print("unified per-lane march state machine (primary → normals → surface work → shadow rays, one evaluation per trip), "
          "surface block parked until T lanes wait: " + ", ".join(f"T={t}: {v / wave[0]:.3f}" for t, v in zip((1, 8, 16, 32), un)))
    if args.stride == 1:
        s0, s1 = (C.c_float * (1 << 18))(), (C.c_float * (1 << 18))()
        sim.sim_wave_costs_spread(s0, s1, 1 << 18)
        for name, arr in (("shipped", cst), ("shadow rays of all lights together when <= 64", np.array(s0[:m], dtype=np.float64)),
                          ("shadow rays of all lights together always", np.array(s1[:m], dtype=np.float64))):
            order = np.argsort(ix, kind="stable")
            heap = [0.0] * 4096
            end = 0.0
            for c in arr[order]:
                t = heapq.heappop(heap) + c
                end = max(end, t)
                heapq.heappush(heap, t)
            print(f"  {name:48s}: total {arr.sum() / cst.sum():.3f}, heaviest wave {arr.max():.0f}, raster-order makespan {end / (cst.sum() / 4096):.3f} (x shipped balanced)")
        pr, hh, mp, ms_ = (C.c_float * (1 << 18))(), (C.c_short * (1 << 18))(), (C.c_short * (1 << 18))(), (C.c_short * (1 << 18))()
        sim.sim_wave_detail(pr, hh, mp, ms_, 1 << 18)
        pr, hh, mp, ms_ = (np.array(a[:m]) for a in (pr, hh, mp, ms_))
        gx = (args.W + 31) // 32
        top = np.argsort(-cst)[:25]
        print("heaviest waves: cost, primary share, hit lanes, max primary steps, max shadow steps, tile (x, y) of 480 x 270")
        for i in top:
            wv = ix[i]; bx = (wv // 4) % gx; by = (wv // 4) // gx
            print(f"   {cst[i]:8.0f} {pr[i] / cst[i]:5.2f} {hh[i]:3d} {mp[i]:4d} {ms_[i]:4d}   ({bx * 4 + wv % 4}, {by})")
        for lo, hi in ((0, 1), (1, 16), (16, 48), (48, 65)):
            sel = (hh >= lo) & (hh < hi)
            print(f"   waves with {lo}..{hi - 1} hit lanes: {sel.mean():.3f} of waves, {cst[sel].sum() / cst.sum():.3f} of cost, mean {cst[sel].mean():.0f}, max {cst[sel].max():.0f}")
        print("corr(est, cost) =", np.corrcoef(es, cst)[0, 1], " share of waves with cost <= 2x min:", (cst <= 2 * cst.min()).mean(),
              " cost share of the top 1% waves:", np.sort(cst)[-m // 100:].sum() / cst.sum())
        print(f"waves {m}: mean cost {cst.mean():.0f}, max {cst.max():.0f} wave-instructions; total/4096 slots = {cst.sum() / 4096:.0f}")
        for name, order in (("dispatch order (row-major workgroups)", np.argsort(ix, kind="stable")),
                            ("heaviest first (oracle knowledge)", np.argsort(-cst, kind="stable")),
                            ("by the cost of the tile's 4 centre pixels", np.argsort(-es, kind="stable")),
                            ("cheap (<= 2x min cost) waves last, rest row-major", np.lexsort((ix, cst <= 2 * cst.min()))),
                            ("centre rows first", None)):
            if order is None:
                gx = (args.W + 31) // 32
                row = (ix // 4) // gx
                gy = (args.H + 7) // 8
                order = np.lexsort((ix, np.abs(row - gy / 2)))
            for slots in (4096, 3072):
                heap = [0.0] * slots
                heapq.heapify(heap)
                end = 0.0
                for c in cst[order]:
                    t = heapq.heappop(heap) + c
                    end = max(end, t)
                    heapq.heappush(heap, t)
                print(f"  {name:40s} {slots} slots: makespan / balanced = {end / (cst.sum() / slots):.3f}")


if __name__ == "__main__":
    main()
