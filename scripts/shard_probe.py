#!/usr/bin/env python3
"""One shard of an 8-way sharded 4K bulb frame on one GPU: wall time per frame against the kernel time, i.e. the fixed
per-frame overhead (scene staging, sort launches, host) that limits multi-GPU scaling.  GPU box only."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from raymarcher_amd import Renderer, abi, lib, scenes
    W, H = 3840, 2160
    r = Renderer(0)
    L = lib()
    t = scenes.mandelbulb(W, H)
    s = abi.default_settings(fractalIters=12)
    n1 = float("nan")
    for N in (1, 2, 4, 8):
        rows = L.rm_shard_rows(H, 8, 0, N)
        out = torch.empty((rows, W, 4), dtype=torch.float32, device=r.device)
        for k in (0, N - 1):
            for _ in range(5):
                r.render_tiles(t, s, W, H, 8, k, N, out=out)
            torch.cuda.synchronize()
            L.rm_set_timing(1)
            n = 50
            t0 = time.perf_counter()
            for _ in range(n):
                r.render_tiles(t, s, W, H, 8, k, N, out=out)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n * 1e3
            ms, kk = C.c_double(), C.c_int()
            st = (C.c_double * 4)()
            L.rm_get_stage_timing(C.byref(ms), st, C.byref(kk))
            L.rm_set_timing(0)
            if N == 1:
                n1 = st[1]  # the whole frame's render time: the yardstick of the shards
            print(f"N={N} shard {k}: wall {dt:.3f} ms per frame, launch (events) {ms.value:.3f} ms = sort {st[0]:.3f} + render {st[1]:.3f}; "
                  f"ideal 1/N of the N=1 render would be {n1 / N:.3f}")
            # frames are independent: several in flight on their own streams hide the serial chain of a straggler ray
            for S in (2, 3, 4, 6):
                streams = [torch.cuda.Stream(device=r.device) for _ in range(S)]
                outs = [torch.empty_like(out) for _ in range(S)]
                for i in range(3 * S):
                    with torch.cuda.stream(streams[i % S]):
                        r.render_tiles(t, s, W, H, 8, k, N, out=outs[i % S])
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(n):
                    with torch.cuda.stream(streams[i % S]):
                        r.render_tiles(t, s, W, H, 8, k, N, out=outs[i % S])
                torch.cuda.synchronize()
                print(f"      {S} streams: wall {(time.perf_counter() - t0) / n * 1e3:.3f} ms per frame")


if __name__ == "__main__":
    main()
