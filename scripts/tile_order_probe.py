#!/usr/bin/env python3
"""Potential of launching heavy tiles first: measure every tile's cost (shader cycles, from the kernel itself), sort,
and time the 4K bulb frame with workgroups started heaviest-first.  GPU box only."""
import ctypes as C
import os
import sys

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
    out = torch.empty((H, W, 4), dtype=torch.float32, device=r.device)
    ntiles = ((W + 15) // 16) * ((H + 7) // 8)

    def timed(n=20):
        for _ in range(3):
            r.render(t, s, W, H, out=out)
        torch.cuda.synchronize()
        L.rm_set_timing(1)
        for _ in range(n):
            r.render(t, s, W, H, out=out)
        torch.cuda.synchronize()
        ms, k = C.c_double(), C.c_int()
        st = (C.c_double * 4)()
        L.rm_get_stage_timing(C.byref(ms), st, C.byref(k))
        L.rm_set_timing(0)
        timed.stages = list(st)
        return ms.value

    L.rm_set_tile_order(0)
    base = timed()
    ref = out.clone()
    for mode, name in ((1, "feedback from the previous frame"),):
        L.rm_set_tile_order(mode)
        ms = timed()
        same = bool((out.view(torch.int32) == ref.view(torch.int32)).all())
        print(f"mode {mode} {name:34s} {ms:.3f} ms  ({W * H / ms / 1e3:.0f} Mpix/s)  identical frame: {same}  stages {[round(x, 3) for x in timed.stages[:2]]}")
    L.rm_set_tile_order(0)
    cost = torch.zeros(ntiles, dtype=torch.int32, device=r.device)
    L.rm_debug_set_tile_order(None, C.c_void_p(cost.data_ptr()), ntiles)
    r.render(t, s, W, H, out=out)
    torch.cuda.synchronize()
    c = cost.clone()
    print(f"default order {base:.3f} ms; tile cost: mean {c.float().mean().item():.0f}, max {c.max().item()} (x64 cycles)")
    for name, order in (("heaviest first", torch.argsort(c, descending=True)), ("lightest first", torch.argsort(c)),
                        ("random", torch.randperm(ntiles, device=r.device)),
                        ("heaviest first, 8 buckets by log2", torch.argsort((c.float() + 1).log2().floor().clamp(min=c.float().max().log2().item() - 7), descending=True, stable=True))):
        o = order.to(torch.int32).contiguous()
        L.rm_debug_set_tile_order(C.c_void_p(o.data_ptr()), None, ntiles)
        ms = timed()
        same = bool((out.view(torch.int32) == ref.view(torch.int32)).all())
        print(f"{name:36s} {ms:.3f} ms  ({W * H / ms / 1e3:.0f} Mpix/s)  identical frame: {same}")
    L.rm_debug_set_tile_order(None, None, 0)


if __name__ == "__main__":
    main()
