#!/usr/bin/env python3
"""bench.py — Mpixels/s of the per-pixel raymarch on the north-star workload.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A step = one 3840×2160 Mandelbulb frame (scenefiles/simple/unit_mandelbulb.json as constants, power 8,
256 march steps, 12 fractal iterations — BASELINE.json configs[2]) rendered through the C-ABI into a
float4 HBM framebuffer.  With N > 1 the frame is strong-scaled: rank r renders row tiles t ≡ r (mod N)
(rm_render_tiles) and rank 0 gathers them with RCCL (dist.gather) and de-interleaves; the gather is
inside the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H = 3840, 2160
TILE_ROWS = 8
FRACTAL_ITERS = 12
# algorithmic work of the reference's formulation (fma = 2 flop, SFU = 1 flop-equivalent)
FLOP_PER_ITER = 67 + 12     # one Mandelbulb inner iteration (frag:786-798)
FLOP_PER_EVAL = 38 + 3      # one sdScene evaluation besides its iterations (frag:1406-1430, 802, 1461-1469)
FLOP_PER_HIT = 1800         # 4×pnoise + Phong per shaded pixel
PEAK_FP32_TFLOPS = 157.3    # MI355X_MICROARCH.md: peak FP32 vector
PEAK_HBM_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E spec peak


def cpu_baseline(settings):
    """Oracle (CPU port of the same frame) timed on a bounded, unbiased sample of the same workload:
    row-strided passes over the 4K frame (stride 8, offsets 4,0,1,…) until ≈10 s of wall time or the
    whole frame is done.  One OpenMP-free oracle call per row, `cores` rows in flight."""
    import ctypes as C
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as h
    from raymarcher_amd import scenes
    t = scenes.mandelbulb(W, H)
    cores = len(os.sched_getaffinity(0))
    lib = h.oracle()

    def one(r):
        buf = np.empty((1, W, 4), dtype=np.float32)
        st = lib.rmo_render(C.byref(t.camera), t.objects, t.num_objects, t.lights, t.num_lights, C.byref(t.globals_),
                            C.byref(settings), W, H, r, r + 1, h.fptr(buf), None, None, 1)
        assert st == 0
        return r

    done_rows, passes = 0, []
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        for off in (4, 0, 1, 2, 3, 5, 6, 7):
            rows = list(range(off, H, 8))
            list(ex.map(one, rows))
            done_rows += len(rows)
            passes.append(off)
            if time.perf_counter() - t0 > 10.0:
                break
    dt = time.perf_counter() - t0
    what = "the whole frame" if done_rows == H else f"rows ≡ {passes} (mod 8): {done_rows} of {H} rows"
    return {"value": round(done_rows * W / dt / 1e6, 4), "unit": "Mpixels/s", "cores": cores, "kind": "port",
            "sample": f"{what} of the same 3840x2160 Mandelbulb frame ({done_rows * W} px), {dt:.1f} s wall, "
                      f"scalar C oracle, one row per task on {cores} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--bulb-eval", choices=["reference", "algebraic"], default="reference",
                    help="reference: acos/atan/sin/cos/pow as the shader writes the step (the headline); algebraic: "
                         "RM_FEAT_BULB_POWER8_ALGEBRAIC, the same step by complex squarings (also reported as a variant)")
    ap.add_argument("--no-variants", action="store_true", help="skip the extra timing of the algebraic variant (profiling runs)")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the N > 1 code path (RCCL process group, pipelined gather, de-interleave) even with one rank: "
                         "a rehearsal of the multi-GPU path on a one-GPU box")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from raymarcher_amd import Renderer, abi, lib, scenes
    from raymarcher_amd.dist import FramePipeline, ShardPlan

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    distributed = world > 1 or args.force_dist
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    r = Renderer(local_rank)
    tables = scenes.mandelbulb(W, H)
    feats = abi.RM_FEAT_REFERENCE_DEFAULT | (abi.RM_FEAT_BULB_POWER8_ALGEBRAIC if args.bulb_eval == "algebraic" else 0)
    settings = abi.default_settings(fractalIters=FRACTAL_ITERS, features=feats)
    L = lib()
    plan = ShardPlan(H, TILE_ROWS, world)
    my_rows, slot_rows = plan.rows(rank), plan.slot_rows  # shard 0 owns the most rows → equal gather slots
    mine = torch.zeros((slot_rows, W, 4), dtype=torch.float32, device=r.device) if not distributed else None
    frame_holder = {}
    # N > 1: frames are independent, so frame i's gather (xGMI) runs under frame i+1's render; every frame still ends as a
    # complete float4 frame on rank 0 (rm_deinterleave), and the timed region ends only when the last one has.
    pipe = FramePipeline(plan, rank, (W, 4), torch.float32, r.device,
                         finish=lambda g: frame_holder.__setitem__("f", r.deinterleave(g, W, H, TILE_ROWS, world, slot_rows))) \
        if distributed else None

    def step():
        if not distributed:
            frame_holder["f"] = r.render(tables, settings, W, H, out=mine)
            return
        pipe.submit(lambda slot: r.render_tiles(tables, settings, W, H, TILE_ROWS, rank, world, out=slot[:my_rows]))

    def fence():
        if distributed:
            pipe.drain()
        torch.cuda.synchronize(r.device)
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(r.device)

    for _ in range(args.warmup):
        step()
    fence()
    L.rm_set_timing(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    import ctypes as C
    kms, kn = C.c_double(), C.c_int()
    stages = (C.c_double * 4)()
    L.rm_get_stage_timing(C.byref(kms), stages, C.byref(kn))
    L.rm_set_timing(0)
    tmax = torch.tensor([dt], dtype=torch.float64, device=r.device)
    kmax = torch.tensor([kms.value], dtype=torch.float64, device=r.device)
    if distributed:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(kmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    kernel_ms = float(kmax.item())

    # the opt-in evaluation scheme of the same step, timed beside the headline (single GPU only; never `value`)
    variant = None
    if not distributed and args.bulb_eval == "reference" and not args.no_variants:
        vs = abi.default_settings(fractalIters=FRACTAL_ITERS, features=feats | abi.RM_FEAT_BULB_POWER8_ALGEBRAIC)
        r.render(tables, vs, W, H, out=mine)
        fence()
        L.rm_set_timing(1)
        tv = time.perf_counter()
        nv = max(3, min(args.steps, 10))
        for _ in range(nv):
            r.render(tables, vs, W, H, out=mine)
        fence()
        dv = time.perf_counter() - tv
        vk, vn = C.c_double(), C.c_int()
        L.rm_get_timing(C.byref(vk), C.byref(vn))
        L.rm_set_timing(0)
        variant = {"value": round(W * H * nv / dv / 1e6, 2), "unit": "Mpixels/s", "ms_per_step": round(dv / nv * 1e3, 4),
                   "kernel_ms": round(vk.value, 4), "steps": nv,
                   "what": "RM_FEAT_BULB_POWER8_ALGEBRAIC: w^8 by complex squarings instead of acos/atan/sin/cos/pow; same "
                           "function, |ΔDE| median 4e-8, 0.08 % of frame pixels differ by > 1e-3 from the headline frame"}

    # algorithmic work of this rank's launch, from the frame's deterministic counters (outside the timed region)
    cnt = None
    if rank == 0:
        _, cnt = r.render_counted(tables, settings, W, H)
    if rank == 0:
        path = int(os.environ.get("RM_KERNEL_PATH", "0"))
        kernel_name = {0: "rm::render_kernel<BULB=true,COUNT=false,ENV=false,TEX=false> (one lane per pixel, 8x8 tile per wave)",
                       1: "rm::render_kernel<BULB=true,COUNT=false,ENV=false,TEX=false> (one lane per pixel, 8x8 tile per wave)",
                       2: "pipeline A: bulb_primary+surface+shadow+shade kernels (state machines + lane refill)",
                       3: "pipeline B: bulbB_primary+surface+shadow+shade kernels (compacted lists, plain loops)",
                       4: "pipeline C: pipeline B with step-budgeted march passes and re-compaction"}[path]
        mpix = W * H * args.steps / dt / 1e6
        flops_frame = cnt.bulbIters * FLOP_PER_ITER + cnt.sceneEvals * FLOP_PER_EVAL + cnt.hitPixels * FLOP_PER_HIT
        # the dominant kernel of one launch processes 1/world of the frame (interleaved tiles ≈ equal work)
        flops_launch = flops_frame / world
        achieved = flops_launch / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
        bytes_launch = W * H * 16 / world
        line = {
            "metric": "Mpixels/s at 3840x2160 Mandelbulb, 256 march steps",
            "value": round(mpix, 2), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Mandelbulb power 8, 12 iters, 3840x2160, 256 steps, 3 directional lights, "
                                   "Perlin bump, white background (unit_mandelbulb.json as constants)"
                                   + ("; step evaluated with RM_FEAT_BULB_POWER8_ALGEBRAIC" if args.bulb_eval == "algebraic" else ""),
                       "rows": "whole frame" if world == 1 else f"{TILE_ROWS}-row tiles round-robin over {world} GPUs; RCCL gather of "
                               "frame i to rank 0 overlapped with the render of frame i+1; every frame de-interleaved on rank 0",
                       "parity": "bit-exact vs CPU oracle (rm_math contract)"},
            "roofline": {"bound": "valu", "achieved": round(achieved, 3), "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_FP32_TFLOPS, 4),
                         # HBM bytes per launch measured with rocprofv3 PMC (WRITE_SIZE + 2·FETCH_SIZE, separate passes):
                         # exactly 16 B/pixel written + 0.55 MB read — profiles/r01_k_hbm_pmc.md
                         "traffic": int(bytes_launch + 552000) if path in (0, 1) else None,
                         "traffic_source": "profiles/r01_k_hbm_pmc.md (rocprofv3 PMC of this kernel and workload)",
                         "kernel": kernel_name, "kernel_ms": round(kernel_ms, 4),
                         "stage_ms": {"primary": round(stages[0], 4), "surface": round(stages[1], 4),
                                      "shadow": round(stages[2], 4), "shade": round(stages[3], 4)},
                         "algorithmic": {"flop_per_launch": flops_launch, "sceneEvals": cnt.sceneEvals,
                                         "bulbIters": cnt.bulbIters, "hitPixels": cnt.hitPixels},
                         "hbm": {"achieved": round(bytes_launch / (kernel_ms * 1e-3) / 1e9, 2) if kernel_ms > 0 else 0.0,
                                 "peak": PEAK_HBM_GBS, "unit": "GB/s", "bytes_per_pixel": 16}},
        }
        if variant is not None:
            line["variants"] = {"bulb_power8_algebraic": variant}
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(settings)
        print(json.dumps(line), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
