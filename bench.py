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
SFU_PER_ITER, SFU_PER_EVAL = 12, 3  # special-function ops inside the two figures above
PEAK_FP32_TFLOPS = 157.3    # MI355X_MICROARCH.md: peak FP32 vector
PEAK_LANE_SLOTS = 256 * 4 * 32 * 2.4e9  # VALU issue slots per second: 256 CUs x 4 SIMD-32 at 2.4 GHz (an fma = 1 slot = 2 flop)
PEAK_HBM_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E spec peak


def host_cores():
    """CPU cores this process may really use: the affinity mask capped by the cgroup CPU quota (a one-GPU box exposes
    every core of the host in the mask but grants a 16-core share)."""
    n = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = max(1, min(n, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, p_ = int(f.read()), int(g.read())
            if q > 0:
                n = max(1, min(n, -(-q // p_)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(settings, gpu_frame):
    """The oracle (the CPU port of the same frame, built -O3) on the host's cores: the WHOLE 3840x2160 frame in one
    OpenMP call (schedule(dynamic) over rows, every core), then a bounded single-thread sample (every 72nd row).  The
    oracle's frame is then compared bit for bit with `gpu_frame` — the frame the timed region produced — so the number
    this line reports is for a frame that was checked in the same run."""
    import ctypes as C
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as h
    from raymarcher_amd import scenes
    t = scenes.mandelbulb(W, H)
    cores = host_cores()
    lib = h.oracle()

    def render(rows0, rows1, threads, out):
        st = lib.rmo_render(C.byref(t.camera), t.objects, t.num_objects, t.lights, t.num_lights, C.byref(t.globals_),
                            C.byref(settings), W, H, rows0, rows1, h.fptr(out), None, None, threads)
        assert st == 0

    frame = np.empty((H, W, 4), dtype=np.float32)
    t0 = time.perf_counter()
    render(0, H, cores, frame)
    dt = time.perf_counter() - t0
    # single thread: rows 36, 108, ... (30 rows spread over the frame) — bounded to a few seconds
    rows1 = list(range(36, H, 72))
    buf = np.empty((1, W, 4), dtype=np.float32)
    t1 = time.perf_counter()
    for r in rows1:
        render(r, r + 1, 1, buf)
    dt1 = time.perf_counter() - t1
    got = gpu_frame.cpu().numpy()
    bad = int((got.view(np.uint32) != frame.view(np.uint32)).sum())
    parity = {"rows": H, "pixels": W * H, "mismatched_words": bad,
              "what": "the last frame of the timed region against the oracle's frame rendered in this run, 32-bit words"}
    base = {"value": round(W * H / dt / 1e6, 4), "unit": "Mpixels/s", "cores": cores, "kind": "port",
            "sample": f"the whole 3840x2160 Mandelbulb frame ({W * H} px) in {dt:.2f} s: scalar C oracle -O3, one OpenMP "
                      f"call, schedule(dynamic) over rows on {cores} threads",
            "single_thread": {"value": round(len(rows1) * W / dt1 / 1e6, 5), "unit": "Mpixels/s", "cores": 1,
                              "sample": f"{len(rows1)} rows (every 72nd) of the same frame, {len(rows1) * W} px in {dt1:.1f} s"}}
    return base, parity


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--bulb-eval", choices=["reference", "algebraic"], default="reference",
                    help="reference: acos/atan/sin/cos/pow as the shader writes the step (the headline); algebraic: "
                         "RM_FEAT_BULB_POWER8_ALGEBRAIC, the same step by complex squarings (also reported as a variant)")
    ap.add_argument("--no-variants", action="store_true", help="skip the extra timings (algebraic step, raster tile order): profiling runs")
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
        os.environ.setdefault("MASTER_PORT", "29517")  # only missing in the one-process rehearsal (--force-dist without a launcher)
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
    # three frames in flight, each on its own stream: the renders of consecutive frames overlap as well (a shard's frame
    # cannot end before its longest ray chain, ≈0.7–0.9 ms, which is 3× the shard's work at N = 8; dist.FramePipeline)
    pipe = FramePipeline(plan, rank, (W, 4), torch.float32, r.device, depth=3, multi_stream=True,
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
    timed_frame = frame_holder["f"].clone() if rank == 0 else None  # what the timed region produced (checked below)
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

    # the same frame without the tile-order feedback (raster order: what a first frame, or a frame after a change of size,
    # costs); the headline's timed frames all ran with the previous frame's tile costs
    raster = None
    if not distributed and not args.no_variants:
        L.rm_set_tile_order(0)
        r.render(tables, settings, W, H, out=mine)
        fence()
        L.rm_set_timing(1)
        tr = time.perf_counter()
        nr = max(3, min(args.steps, 10))
        for _ in range(nr):
            r.render(tables, settings, W, H, out=mine)
        fence()
        dr = time.perf_counter() - tr
        rk, rn = C.c_double(), C.c_int()
        L.rm_get_timing(C.byref(rk), C.byref(rn))
        L.rm_set_timing(0)
        L.rm_set_tile_order(-1)
        raster = {"value": round(W * H * nr / dr / 1e6, 2), "unit": "Mpixels/s", "ms_per_step": round(dr / nr * 1e3, 4),
                  "kernel_ms": round(rk.value, 4), "steps": nr,
                  "what": "rm_set_tile_order(0): tiles start in raster order — a frame with no history"}

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

    # three frames in flight, each on its own stream and into its own buffer: a frame's last straggler rays (a serial chain
    # of ≈0.7 ms) overlap the next frames' full waves — what a caller that renders a sequence gets; never `value`
    inflight = None
    if not distributed and not args.no_variants:
        streams = [torch.cuda.Stream(device=r.device) for _ in range(3)]
        bufs = [torch.empty((slot_rows, W, 4), dtype=torch.float32, device=r.device) for _ in range(3)]
        nf = max(6, min(args.steps, 30))
        for rep in range(2):  # first pass: every stream learns its own tile order
            fence()
            tf = time.perf_counter()
            for i in range(nf):
                with torch.cuda.stream(streams[i % 3]):
                    r.render(tables, settings, W, H, out=bufs[i % 3])
            fence()
            df = time.perf_counter() - tf
        same = all(bool(torch.equal(b.view(torch.int32), timed_frame.view(torch.int32))) for b in bufs)
        inflight = {"value": round(W * H * nf / df / 1e6, 2), "unit": "Mpixels/s", "ms_per_step": round(df / nf * 1e3, 4), "steps": nf,
                    "frames_identical_to_headline": same,
                    "what": "the same frames submitted round-robin on three HIP streams (three in flight)"}

    # work of this rank's launch from the frame's deterministic counters (outside the timed region): what the REFERENCE's
    # formulation does (algorithmic) and what the production kernel really executes (bit-identical shortcuts honoured);
    # then the shader clock the chip held under this kernel's own load (stamped diagnostic build, after the timed launches)
    cnt = cnt_exec = None
    clock_mhz = None
    if rank == 0:
        _, cnt = r.render_counted(tables, settings, W, H, abi.RM_COUNT_REFERENCE)
        _, cnt_exec = r.render_counted(tables, settings, W, H, abi.RM_COUNT_EXECUTED)
        for _ in range(5):
            r.render(tables, settings, W, H, out=mine)
        _, clock_mhz = r.render_clocked(tables, settings, W, H)
    if rank == 0:
        path = int(os.environ.get("RM_KERNEL_PATH", "0"))
        kernel_name = {0: "rm::render_kernel<BULB=true,COUNT=false,ENV=false,TEX=false> (one lane per pixel, 8x8 tile per wave)",
                       1: "rm::render_kernel<BULB=true,COUNT=false,ENV=false,TEX=false> (one lane per pixel, 8x8 tile per wave)",
                       2: "pipeline A: bulb_primary+surface+shadow+shade kernels (state machines + lane refill)",
                       3: "pipeline B: bulbB_primary+surface+shadow+shade kernels (compacted lists, plain loops)",
                       4: "pipeline C: pipeline B with step-budgeted march passes and re-compaction"}[path]
        mpix = W * H * args.steps / dt / 1e6
        def work(c):  # (flop, issue slots) of a frame with these counters; an SFU op is 1 flop-equivalent but 4 slots
            flop = c.bulbIters * FLOP_PER_ITER + c.sceneEvals * FLOP_PER_EVAL + c.hitPixels * FLOP_PER_HIT
            slots = (c.bulbIters * (FLOP_PER_ITER - SFU_PER_ITER) + c.sceneEvals * (FLOP_PER_EVAL - SFU_PER_EVAL)
                     + c.hitPixels * FLOP_PER_HIT) / 2 + 4 * (c.bulbIters * SFU_PER_ITER + c.sceneEvals * SFU_PER_EVAL)
            return flop, slots
        flops_frame, slots_frame = work(cnt)
        flops_exec, slots_exec = work(cnt_exec)
        # the dominant kernel of one launch processes 1/world of the frame (interleaved tiles ≈ equal work)
        flops_launch = flops_frame / world
        # with tile-order feedback a launch is two sort kernels (stage 0, ~0.02 ms) + the render kernel (stage 1): the roofline is
        # the render kernel's, `kernel_ms` stays the whole launch
        ordered = path in (0, 1) and stages[1] > 0.0
        render_ms = stages[1] if ordered else kernel_ms
        # with several frames in flight (N > 1) the launches overlap and their event spans are not kernel time: the roofline is
        # then taken over the wall time one frame of this rank's shard costs
        secs = (dt / args.steps if distributed else render_ms * 1e-3)
        achieved = flops_launch / secs / 1e12 if kernel_ms > 0 else 0.0
        executed = flops_exec / world / secs / 1e12 if kernel_ms > 0 else 0.0
        bytes_launch = W * H * 16 / world
        line = {
            "metric": "Mpixels/s at 3840x2160 Mandelbulb, 256 march steps",
            "value": round(mpix, 2), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Mandelbulb power 8, 12 iters, 3840x2160, 256 steps, 3 directional lights, "
                                   "Perlin bump, white background (unit_mandelbulb.json as constants)"
                                   + ("; step evaluated with RM_FEAT_BULB_POWER8_ALGEBRAIC" if args.bulb_eval == "algebraic" else ""),
                       "rows": "whole frame" if not distributed else f"{TILE_ROWS}-row tiles round-robin over {world} GPUs; three frames in "
                               "flight per GPU on three streams (renders of consecutive frames overlap, RCCL gather of frame i "
                               "to rank 0 runs under later renders); every frame de-interleaved on rank 0",
                       "tile_order": "feedback: each frame records its tiles' shader-cycle costs, the next frame starts heavy tiles "
                                     "first (same pixels, same work; variants.raster_tile_order = no history)",
                       "parity": "bit-exact vs CPU oracle (rm_math contract)"},
            "roofline": {"bound": "valu", "achieved": round(achieved, 3), "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_FP32_TFLOPS, 4),
                         # algorithmic = the reference's work ÷ this kernel's time; executed = the work this kernel really
                         # does (bounding-ball culls, no shadow march for dropped lights) ÷ the same time
                         "executed": {"achieved": round(executed, 3), "frac": round(executed / PEAK_FP32_TFLOPS, 4),
                                      "sceneEvals": cnt_exec.sceneEvals, "bulbIters": cnt_exec.bulbIters,
                                      "flop_per_launch": flops_exec / world},
                         # issue-slot view: an fma is one VALU slot, a special-function op four
                         "slots": {"algorithmic_frac": round(slots_frame / world / secs / PEAK_LANE_SLOTS, 4) if kernel_ms > 0 else 0.0,
                                   "executed_frac": round(slots_exec / world / secs / PEAK_LANE_SLOTS, 4) if kernel_ms > 0 else 0.0,
                                   "peak_lane_slots_per_s": PEAK_LANE_SLOTS},
                         "shader_clock_mhz": round(clock_mhz, 1) if clock_mhz else None,
                         "clock_source": "s_memtime / s_memrealtime stamps of every wave of one launch of the same kernel "
                                         "(diagnostic build, rm_render_clocked) after back-to-back launches",
                         # HBM bytes are not measured by this run: algorithmic 16 B/pixel written; the PMC measurement of this
                         # kernel (WRITE_SIZE + 2·FETCH_SIZE, separate passes) is in profiles/ (133.26 MB per 4K launch)
                         "traffic": None,
                         "traffic_source": "not collected by bench.py; rocprofv3 PMC: profiles/r02_r_hbm_pmc.md (218 MB per launch = 1.64 x the 16 B/pixel: register spills of the 5-waves-per-SIMD budget; 1.3 % of the HBM peak)",
                         "kernel": kernel_name, "kernel_ms": round(render_ms, 4), "launch_ms": round(kernel_ms, 4),
                         "time_base": ("wall time per frame (three overlapping frames in flight: event spans of single launches "
                                       "are longer than their share of the GPU)" if distributed else "HIP events around the render kernel"),
                         "stage_ms": ({"tile_order_sort": round(stages[0], 4), "render_kernel": round(stages[1], 4)} if ordered else
                                      {"primary": round(stages[0], 4), "surface": round(stages[1], 4),
                                       "shadow": round(stages[2], 4), "shade": round(stages[3], 4)}),
                         "algorithmic": {"flop_per_launch": flops_launch, "sceneEvals": cnt.sceneEvals,
                                         "bulbIters": cnt.bulbIters, "hitPixels": cnt.hitPixels},
                         "hbm": {"achieved": round(bytes_launch / (kernel_ms * 1e-3) / 1e9, 2) if kernel_ms > 0 else 0.0,
                                 "peak": PEAK_HBM_GBS, "unit": "GB/s", "bytes_per_pixel": 16}},
        }
        line["variants"] = {}
        if variant is not None:
            line["variants"]["bulb_power8_algebraic"] = variant
        if raster is not None:
            line["variants"]["raster_tile_order"] = raster
        if inflight is not None:
            line["variants"]["three_frames_in_flight"] = inflight
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"], line["parity_check"] = cpu_baseline(settings, timed_frame)
        print(json.dumps(line), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
