#!/usr/bin/env python3
"""Would the heaviest tiles of a latency-bound frame end sooner as four 16-lane waves?  Per 8×8 wave, from the oracle's per-pixel march
traces: the wave's length in evaluation trips under the shipped schedule (lanes meet at every raymarch, shadow marches as a per-lane
queue), whole and as the longest of its four quarters.
  python scripts/sim/run_split_sim.py tests/golden/scenes/lighting/directional_light_2.json 1920 1080 --soft --ao"""
import argparse
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scene")
    ap.add_argument("W", type=int)
    ap.add_argument("H", type=int)
    ap.add_argument("--soft", action="store_true")
    ap.add_argument("--ao", action="store_true")
    ap.add_argument("--reflection", action="store_true")
    ap.add_argument("--dump-above", type=float, default=1e300, help="print the longest lane's marches of every wave at least this long")
    a = ap.parse_args()
    here = os.path.dirname(os.path.abspath(__file__))
    so = os.path.join(here, "_build", "libwave_sim.so")
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call(["gcc", "-O2", "-std=gnu99", "-fopenmp", "-mfma", "-mavx2", "-mf16c", "-ffp-contract=off", "-fno-fast-math",
                           "-shared", "-fPIC", "-Wno-unused-function", "-o", so, os.path.join(here, "wave_sim.c"), "-lm"])
    from raymarcher_amd import Scene, abi
    sim = C.CDLL(so)
    t = Scene(path=a.scene).tables(a.W, a.H)
    s = abi.default_settings(enableReflection=int(a.reflection), enableSoftShadow=int(a.soft), enableAmbientOcclusion=int(a.ao))
    nw = ((a.W + 7) // 8) * ((a.H + 7) // 8)
    buf = np.zeros(2 * nw, dtype=np.float64)
    sim.sim_set_wave_len_buffer(buf.ctypes.data_as(C.c_void_p))
    pack2 = np.zeros(2 * nw, dtype=np.float64)
    sim.sim_set_pack_len_buffer(pack2.ctypes.data_as(C.c_void_p), nw)
    sim.sim_set_dump_above.argtypes = [C.c_double]
    sim.sim_set_dump_above(a.dump_above)
    out = (C.c_double * 10)()
    assert sim.sim_generic(C.byref(t.camera), t.objects, t.num_objects, t.lights, t.num_lights, C.byref(t.globals_), C.byref(s),
                           a.W, a.H, 1, out, len(os.sched_getaffinity(0))) == 0
    whole, quarter = buf[0::2], buf[1::2]
    order = np.argsort(-whole)
    print(f"{os.path.basename(a.scene)} {a.W}x{a.H}: {nw} waves, {whole.sum():.3g} trips in all (mean {whole.mean():.0f} per wave)")
    print(f"  longest wave {whole.max():.0f} trips; its longest quarter {quarter[order[0]]:.0f}")
    for top in (8, 64, 512):
        w, q = whole[order[:top]], quarter[order[:top]]
        print(f"  heaviest {top}: whole {w.mean():.0f} trips on average, longest quarter {q.mean():.0f} ({(q / w).mean():.2f}); max quarter {q.max():.0f}")
    pack, queue = pack2[:nw], pack2[nw:]
    print(f"  shadow rays of a shading round as a WAVE-level queue (a free lane takes the next ray): {queue.sum():.3g} trips in all "
          f"({queue.sum() / whole.sum():.3f} of the shipped), longest wave {queue.max():.0f}; the shipped schedule's heaviest 64 waves then "
          f"average {queue[order[:64]].mean():.0f}; work bound {queue.sum() / (256 * 4 * 6):.0f} per slot")
    print(f"  shadow rays of a shading round packed 64 to a pass (pixel-major): {pack.sum():.3g} trips in all ({pack.sum() / whole.sum():.3f} of the shipped), "
          f"longest wave {pack.max():.0f}; the shipped schedule's heaviest 64 waves then average {pack[order[:64]].mean():.0f}")
    slots = 256 * 4 * 6
    print(f"  work bound: {whole.sum() / slots:.0f} trips per resident-wave slot ({slots} slots) against the longest wave's {whole.max():.0f}")


if __name__ == "__main__":
    main()
