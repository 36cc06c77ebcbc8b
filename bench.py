#!/usr/bin/env python3
"""bench.py — Mpixels/s of the per-pixel raymarch on the BASELINE.json workloads.

  python bench.py --gpus 1 --steps K --warmup W [--config c1|c2|c3|c4|c5]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W [--config …]

A step = one frame of the chosen configuration rendered through the C-ABI into a float4 HBM framebuffer.  Default
(--config c3) is the north-star workload: the 3840×2160 Mandelbulb frame (scenefiles/simple/unit_mandelbulb.json as
constants, power 8, 256 march steps, 12 fractal iterations — BASELINE.json configs[2]).  c1, c2, c4, c5 are the other
BASELINE.json configurations, loaded from the reference's own scenefiles (tests/golden/scenes/, input data) through the
library's loader.  With N > 1 the frame is strong-scaled: rank r renders row tiles t ≡ r (mod N) (rm_render_tiles) and
rank 0 gathers them with RCCL (dist.gather) and de-interleaves; the gather is inside the timed region.  Rank 0 prints
ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
SCENES = os.path.join(ROOT, "tests", "golden", "scenes")

TILE_ROWS = 8
PEAK_FP32_TFLOPS = 157.3    # MI355X_MICROARCH.md: peak FP32 vector
PEAK_LANE_SLOTS = 256 * 4 * 32 * 2.4e9  # VALU issue slots per second: 256 CUs x 4 SIMD-32 at 2.4 GHz (an fma = 1 slot = 2 flop)
PEAK_HBM_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E spec peak

# ---- algorithmic work model (fma = 2 flop, one sqrt / division / exp2 / log2 / sin / cos = 1 flop-equivalent) ----
# SURVEY §8(d), the single-Mandelbulb class: per inner iteration (frag:786-798), per sdScene evaluation besides its
# iterations (frag:1406-1430, 802, 1461-1469), per shaded pixel (4 x pnoise + Phong with 3 lights)
FLOP_PER_ITER = 67 + 12
FLOP_PER_EVAL_BULB = 38 + 3
SFU_PER_ITER, SFU_PER_EVAL = 12, 3  # special-function ops inside the two figures above
# The table-walk classes, counted from the shader text the same way.  One unit primitive of sdMatch (frag:832-894,
# 991-1019, 1262-1293): sphere length − r; box abs, −, max, length, max max min, +; …
SDF_FLOP = {0: 19, 1: 28, 2: 17, 3: 7, 4: 30, 5: 10, 6: 10, 7: 34, 8: 19}  # cube cone cylinder sphere octahedron torus capsule deathstar rectangle
# sdMengerSponge (frag:1049-1071).  AS WRITTEN every evaluation also computes ani = smoothstep(−0.2, 0.2, −cos(0.5·iTime)) and
# off = 1.5·sin(0.01·iTime) (40 flop of launch-uniform arithmetic) and every level the rotation mix p ← mix(p, ma·(p + off), ani)
# (27 flop), which at iTime = 0 (ani = 0) is the identity.  No implementation executes either per lane, so the ALGORITHMIC model
# (roofline.frac) leaves the uniform prologue out and, at ani = 0, the mix; the shader-as-written price is reported beside it
# (roofline.shader_as_written), as round 3's `frac` was.
MENGER_BOX, MENGER_UNIFORM_PROLOGUE, MENGER_PER_LEVEL, MENGER_ROTATION_MIX = 19, 40, 42, 27
SIERPINSKI_FLOP = 14 * 18 + 7                 # 14 fold-scale iterations (frag:819-824), length, constant scale
FLOP_PER_OBJECT = 18 + 1 + 2                  # invModel·(p,1) (9 fma), ·scaleFactor, nearest-object compare / select (frag:1417-1423)
FLOP_PER_STEP = 10                            # ro + rd·t, hit test, t += d (frag:1459-1470 / 1708-1714)
FLOP_PER_PNOISE = 400                         # classic Perlin 3-D, arithmetic permute (frag:1610-1676); bumpNormal takes 4
FLOP_PER_SHADE, FLOP_PER_LIGHT = 50, 50       # view vector, ambient, material; one light of getPhong (frag:1864-1928)
FLOP_PER_FBM9 = 9 * (81 + 11) + 12            # fbm_9 = 9 x (noiseT 81 + octave update) (frag:493-502, 630-644) + sdTerrain's remap
FLOP_PER_FBMD8 = 8 * (164 + 75) + 4 * 21 + 30  # fbmd_8 = 8 x (noised 164 + octave update with the 3x3 derivative chain) (frag:536-567, 647-667) + cloudsMap

# HBM bytes per launch measured with rocprofv3 PMC counters (WRITE_SIZE, FETCH_SIZE in separate passes; the GPU box's
# counters cannot be collected from inside this run): see the named profile.  None = not measured yet.
TRAFFIC_MEASURED, PMC_MEASURED = {}, {}
for _name, _into in (("r04_hbm_traffic.json", TRAFFIC_MEASURED), ("r04_pmc.json", PMC_MEASURED)):
    _file = os.path.join(ROOT, "profiles", _name)
    if os.path.exists(_file):
        with open(_file) as _f:
            _into.update(json.load(_f))


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


# ---------------------------------------------------------------------------------------------------------- configurations
def build_config(name, algebraic=False):
    """(tables, settings, W, H, description dict) of a BASELINE.json configuration."""
    from raymarcher_amd import Scene, abi, scenes
    feats = abi.RM_FEAT_REFERENCE_DEFAULT
    if name == "c3":
        W, H = 3840, 2160
        t = scenes.mandelbulb(W, H)
        s = abi.default_settings(fractalIters=12, features=feats | (abi.RM_FEAT_BULB_POWER8_ALGEBRAIC if algebraic else 0))
        d = {"metric": "Mpixels/s at 3840x2160 Mandelbulb, 256 march steps",
             "workload": "Mandelbulb power 8, 12 iters, 3840x2160, 256 steps, 3 directional lights, Perlin bump, white background "
                         "(unit_mandelbulb.json as constants)" + ("; step evaluated with RM_FEAT_BULB_POWER8_ALGEBRAIC" if algebraic else ""),
             "baseline_config": "configs[2]", "cpu_rows": None}
    elif name == "c1":
        W, H = 256, 256
        t = Scene(path=os.path.join(SCENES, "simple", "unit_sphere.json")).tables(W, H)  # loads texture_store/blackmarble.png
        s = abi.default_settings(maxSteps=64)
        d = {"metric": "Mpixels/s at 256x256 unit_sphere.json, 64 march steps, Phong only",
             "workload": "scenefiles/simple/unit_sphere.json (sphere + textured floor cube, 3 spot lights), 256x256, 64 steps, Phong only",
             "baseline_config": "configs[0]", "cpu_rows": None}
    elif name == "c2":
        W, H = 1920, 1080
        t = Scene(path=os.path.join(SCENES, "lighting", "directional_light_2.json")).tables(W, H)
        s = abi.default_settings(enableSoftShadow=1, enableAmbientOcclusion=1)
        d = {"metric": "Mpixels/s at 1920x1080 directional_light_2.json, soft shadows + AO",
             "workload": "scenefiles/lighting/directional_light_2.json (5 primitives, 3 directional lights), 1920x1080, 256 steps, "
                         "soft shadows + ambient occlusion, Perlin bump",
             "baseline_config": "configs[1]", "cpu_rows": None}
    elif name == "c4":
        W, H = 3840, 2160
        t = Scene(path=os.path.join(SCENES, "simple", "volumetric.json")).tables(W, H, far=2000.0)
        env = abi.RM_FEAT_SKY_BACKGROUND | abi.RM_FEAT_TERRAIN | abi.RM_FEAT_CLOUD | abi.RM_FEAT_PERLIN_BUMP
        s = abi.default_settings(features=env)
        d = {"metric": "Mpixels/s at 3840x2160 FBM terrain + volumetric cloud march",
             "workload": "scenefiles/simple/volumetric.json (torus, 1 directional light) + TERRAIN, CLOUD, SKY_BACKGROUND, 3840x2160; the "
                         "file's own camera (0,500,5) sits below the terrain surface and looks down (variants.horizon_camera = the "
                         "same scene looking at the horizon)",
             "baseline_config": "configs[3]", "cpu_rows": None}
    elif name == "c5":
        W, H = 7680, 4320
        t = Scene(path=os.path.join(SCENES, "simple", "unit_mengersponge.json")).tables(W, H)
        s = abi.default_settings(mengerLevels=5, numReflection=2, enableReflection=1)
        d = {"metric": "Mpixels/s at 7680x4320 Menger sponge depth 5, reflection 2 bounces",
             "workload": "scenefiles/simple/unit_mengersponge.json (3 directional lights), 5 levels, reflection with 2 bounces, "
                         "7680x4320, 256 steps, Perlin bump",
             "baseline_config": "configs[4]", "cpu_rows": list(range(18, H, 36))}  # 120 bands of 8 rows ≈ 15 s on 16 cores
    else:
        raise SystemExit(f"unknown --config {name}")
    return t, s, W, H, d


def flop_model(t, s, cnt, executed=False, as_written=False):
    """Algorithmic flops of a frame from its deterministic counters and the scene table (the reference's formulation,
    independent of how the kernels evaluate it).  as_written=True also charges what the shader text computes per lane but no
    implementation would (the Menger sponge's launch-uniform prologue; its rotation mix where it is the identity).  executed=True
    takes the executed-work counters (culls, objects passed over) at the same unit prices.
    Returns (flop, breakdown dict, model dict)."""
    from raymarcher_amd import abi
    bulb_only = t.num_objects == 1 and t.objects[0].type == abi.RM_MANDELBULB
    per_eval = FLOP_PER_STEP
    for i in range(t.num_objects):
        ty = t.objects[i].type
        if ty == abi.RM_MANDELBULB:
            sdf = FLOP_PER_EVAL_BULB - FLOP_PER_OBJECT - FLOP_PER_STEP  # prologue + distance estimate (its iterations are counted apart)
        elif ty == abi.RM_MENGERSPONGE:
            still = t.globals_.iTime == 0.0  # ani = smoothstep(−0.2, 0.2, −cos 0) = 0: the rotation mix is the identity
            sdf = MENGER_BOX + (MENGER_PER_LEVEL + (0 if still else MENGER_ROTATION_MIX)) * s.mengerLevels
            if as_written:
                sdf = MENGER_BOX + MENGER_UNIFORM_PROLOGUE + (MENGER_PER_LEVEL + MENGER_ROTATION_MIX) * s.mengerLevels
        elif ty == abi.RM_SIERPINSKI:
            sdf = SIERPINSKI_FLOP
        else:
            sdf = SDF_FLOP.get(ty, 20)
        per_eval += FLOP_PER_OBJECT + sdf
    if bulb_only:
        per_eval = FLOP_PER_EVAL_BULB
    per_shade = FLOP_PER_SHADE + FLOP_PER_LIGHT * t.num_lights + (4 * FLOP_PER_PNOISE if s.features & abi.RM_FEAT_PERLIN_BUMP else 0)
    shades = cnt.shadedPoints if cnt.shadedPoints else cnt.hitPixels
    eval_flop = cnt.sceneEvals * per_eval
    shapes = getattr(cnt, "shapeEvals", 0)
    if executed and not bulb_only and t.num_objects > 0 and 0 < shapes < cnt.sceneEvals * t.num_objects:
        # the table walk passed over objects (or followed one alone): price the shapes really evaluated at the table's mean
        # cost per object, plus transform + test (27) for the ones passed over in full walks — an upper bound, since the
        # single-object fast path does not even touch the others
        mean_obj = (per_eval - FLOP_PER_STEP) / t.num_objects
        eval_flop = cnt.sceneEvals * FLOP_PER_STEP + shapes * mean_obj + (cnt.sceneEvals * t.num_objects - shapes) * 27
        model_note = "evaluations priced by shapeEvals (objects really evaluated) + 27 flop per object passed over"
    else:
        model_note = None
    parts = {"iterations": cnt.bulbIters * FLOP_PER_ITER, "evaluations": eval_flop, "shading": shades * per_shade,
             "terrain": cnt.terrainEvals * FLOP_PER_FBM9, "cloud": cnt.cloudEvals * FLOP_PER_FBMD8}
    model = {"flop_per_evaluation": per_eval, "flop_per_iteration": FLOP_PER_ITER, "flop_per_shaded_point": per_shade,
             "flop_per_terrain_eval": FLOP_PER_FBM9, "flop_per_cloud_eval": FLOP_PER_FBMD8}
    if model_note:
        model["note"] = model_note
    return float(sum(parts.values())), parts, model


def slots_model(cnt):
    """Issue-slot view of the single-bulb class: an fma is one VALU slot, a special-function op four."""
    return ((cnt.bulbIters * (FLOP_PER_ITER - SFU_PER_ITER) + cnt.sceneEvals * (FLOP_PER_EVAL_BULB - SFU_PER_EVAL) + cnt.hitPixels * 1800) / 2
            + 4 * (cnt.bulbIters * SFU_PER_ITER + cnt.sceneEvals * SFU_PER_EVAL))


def cpu_baseline(t, s, W, H, rows, gpu_frame):
    """The oracle (the CPU port of the same frame, built -O3) on the host's cores: the whole frame in one OpenMP call
    (schedule(dynamic) over rows, every core) where that takes seconds, a bounded sample of rows spread over the frame
    otherwise, then a single-thread sample.  The oracle's rows are compared bit for bit with the frame the timed region
    produced."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as h
    cores = host_cores()
    scene = (t.camera, t.objects, t.num_objects, t.lights, t.num_lights, t.globals_)
    res = {}
    if t.textures:
        res["textures"] = t.textures
    got = gpu_frame.cpu().numpy()
    bad = 0
    if rows is None:
        # the whole frame, repeated until about five seconds of CPU work have been timed
        t0 = time.perf_counter()
        reps = 0
        while reps == 0 or (time.perf_counter() - t0 < 5.0 and reps < 1000):
            frame = h.oracle_render(scene, s, W, H, 0, H, threads=cores, **res)
            reps += 1
        dt = time.perf_counter() - t0
        px = W * H * reps
        bad = int((got.view(np.uint32) != frame.view(np.uint32)).sum())
        sample = (f"the whole {W}x{H} frame ({W * H} px) {reps} time(s) in {dt:.2f} s: scalar C oracle -O3, one OpenMP call per frame, "
                  f"schedule(dynamic) over rows on {cores} threads")
        rows1 = list(range(H // 60 // 2, H, max(1, H // 30)))[:30]
        checked = H
    else:
        # bounded sample: bands of `band` consecutive rows at the listed positions, all cores busy inside each band
        band = max(1, min(8, cores // 2))
        t0 = time.perf_counter()
        px = 0
        for r in rows:
            ref = h.oracle_render(scene, s, W, H, r, r + band, threads=cores, **res)
            bad += int((got[r:r + band].view(np.uint32) != ref.view(np.uint32)).sum())
            px += band * W
        dt = time.perf_counter() - t0
        sample = (f"{len(rows)} bands of {band} rows spread over the {W}x{H} frame ({px} px) in {dt:.2f} s: scalar C oracle -O3, "
                  f"OpenMP schedule(dynamic) on {cores} threads")
        rows1 = rows[:: max(1, len(rows) // 4)][:4]
        checked = len(rows) * band
    t1 = time.perf_counter()
    for r in rows1:
        h.oracle_render(scene, s, W, H, r, r + 1, threads=1, **res)
    dt1 = time.perf_counter() - t1
    parity = {"rows": checked, "pixels": checked * W, "mismatched_words": bad,
              "what": "the last frame of the timed region against the oracle's rows rendered in this run, 32-bit words"}
    base = {"value": round(px / dt / 1e6, 4), "unit": "Mpixels/s", "cores": cores, "kind": "port", "sample": sample,
            "single_thread": {"value": round(len(rows1) * W / dt1 / 1e6, 5), "unit": "Mpixels/s", "cores": 1,
                              "sample": f"{len(rows1)} rows of the same frame, {len(rows1) * W} px in {dt1:.1f} s"}}
    return base, parity


def timed_frames(r, L, fence, render, n):
    """n frames after one untimed frame; returns (wall ms per frame, HIP-event ms per launch)."""
    import ctypes as C
    render()
    fence()
    L.rm_set_timing(1)
    t0 = time.perf_counter()
    for _ in range(n):
        render()
    fence()
    dt = time.perf_counter() - t0
    k, kn = C.c_double(), C.c_int()
    L.rm_get_timing(C.byref(k), C.byref(kn))
    L.rm_set_timing(0)
    return dt / n * 1e3, k.value


def orbit_variant(r, L, fence, t, s, W, H, frames=24, deg=1.0):
    """A moving sequence: the camera orbits the scene by `deg` degrees per frame (the reference's interactive use,
    src/realtime.cpp:235-281).  Every frame is a NEW picture, so its tiles are ordered by the geometric classification
    (tile_geom_kernel: rings of the objects' bounding balls first) combined with the previous frame's stale costs; timed against
    raster order, and two frames of the sequence are checked bit for bit against renders in raster order."""
    import math
    import torch
    from raymarcher_amd import Scene, abi, scenes
    from raymarcher_amd.render import build_camera

    def cam_at(i):
        a = math.radians(deg * i)
        pos = (4.5 * math.sin(a), 0.0, 4.5 * math.cos(a))
        cam, _, _ = build_camera(pos, tuple(-c for c in pos), (0.0, 1.0, 0.0), math.radians(30.0), W, H)
        return cam

    tabs = []
    for i in range(frames):
        ti = scenes.mandelbulb(W, H)
        ti.camera = cam_at(i)
        tabs.append(ti)
    out = torch.empty((H, W, 4), dtype=torch.float32, device=r.device)
    res = {}
    for mode, key in ((1, "feedback"), (0, "raster")):
        L.rm_set_tile_order(mode)
        r.render(tabs[0], s, W, H, out=out)  # history for frame 0 (feedback mode)
        fence()
        t0 = time.perf_counter()
        for ti in tabs:
            r.render(ti, s, W, H, out=out)
        fence()
        res[key] = (time.perf_counter() - t0) / frames * 1e3
    # frames rendered with a neighbour's tile costs equal frames rendered with no history at all
    L.rm_set_tile_order(1)
    r.render(tabs[6], s, W, H, out=out)
    a = r.render(tabs[7], s, W, H).clone()
    L.rm_set_tile_order(0)
    b = r.render(tabs[7], s, W, H)
    same = bool(torch.equal(a.view(torch.int32), b.view(torch.int32)))
    L.rm_set_tile_order(-1)
    return {"value": round(W * H / res["feedback"] / 1e3, 2), "unit": "Mpixels/s", "ms_per_step": round(res["feedback"], 4),
            "ms_per_step_raster_order": round(res["raster"], 4), "frames": frames, "degrees_per_frame": deg,
            "frame_identical_with_and_without_history": same,
            "what": "camera orbiting the bulb, every frame a new picture: tiles ordered by the geometric classification of the frame's own "
                    "scene and camera combined with the previous frame's (stale) costs, against raster order"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", choices=["c1", "c2", "c3", "c4", "c5"], default="c3",
                    help="BASELINE.json configuration: c3 (default) = the north-star Mandelbulb frame")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--bulb-eval", choices=["reference", "algebraic"], default="reference",
                    help="c3 only.  reference: acos/atan/sin/cos/pow as the shader writes the step (the headline); algebraic: "
                         "RM_FEAT_BULB_POWER8_ALGEBRAIC, the same step by complex squarings (also reported as a variant)")
    ap.add_argument("--no-variants", action="store_true", help="skip the extra timings: profiling runs")
    ap.add_argument("--gather", choices=["float4", "rgba8"], default="float4",
                    help="N > 1: what travels to rank 0 — the float4 tiles (16 B/pixel; every frame ends as a float4 frame on rank 0) or "
                         "their 8-bit conversion (4 B/pixel; every frame ends as the RGBA8 image saveViewportImage would write)")
    ap.add_argument("--gather-root", choices=["zero", "rotate"], default="zero",
                    help="N > 1: where frames end — every frame on rank 0 (default; SURVEY §8e), or frame i on rank i mod N, which spreads "
                         "the root's receive traffic and de-interleave pass over the ranks (also timed as variants.rotating_root)")
    ap.add_argument("--root-relief", default="auto",
                    help="N > 1, --gather-root zero: rm_set_root_relief(K) — rank 0, which also receives N − 1 slots and de-interleaves the "
                         "whole frame every frame, renders (K − 1)/K of a peer's tiles.  auto: 8 from four ranks, 16 for two or three, "
                         "0 (the plain t mod N deal) for one; or an integer (0 = off)")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the N > 1 code path (RCCL process group, pipelined gather, de-interleave) even with one rank: "
                         "a rehearsal of the multi-GPU path on a one-GPU box")
    args = ap.parse_args()
    cfg = args.config
    if args.steps is None:
        args.steps = {"c1": 200, "c2": 100, "c3": 100, "c4": 50, "c5": 20}[cfg]
    if args.warmup is None:
        # c2: the launcher's own measurements of this picture — tile shape (8 frames), settling of the tile order (4), light split
        # (4) — are over after 16 frames
        args.warmup = {"c5": 3, "c2": 24}.get(cfg, 10)

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
    tables, settings, W, H, desc = build_config(cfg, algebraic=(cfg == "c3" and args.bulb_eval == "algebraic"))
    L = lib()
    # the partition: with every frame ending on rank 0, rank 0 is relieved of some tiles (profiles/r04_i_submit_rate.md: its
    # de-interleave pass and N − 1 receives make it the longest pole of every frame); a rotating root needs no relief
    relief = 0
    if distributed and world > 1 and args.gather_root == "zero":
        relief = (8 if world >= 4 else 16) if args.root_relief == "auto" else int(args.root_relief)
    assert L.rm_set_root_relief(relief) == 0
    plan = ShardPlan(H, TILE_ROWS, world)
    my_rows, slot_rows = plan.rows(rank), plan.slot_rows  # shard 0 owns the most rows → equal gather slots
    mine = torch.zeros((slot_rows, W, 4), dtype=torch.float32, device=r.device) if not distributed else None
    frame_holder = {}
    # N > 1: frames are independent, so frame i's gather (xGMI) runs under frame i+1's render; every frame still ends as a
    # complete float4 frame on rank 0 (rm_deinterleave), and the timed region ends only when the last one has.
    # three frames in flight, each on its own stream: the renders of consecutive frames overlap as well (a shard's frame
    # cannot end before its longest ray chain, ≈0.7–0.9 ms, which is 3× the shard's work at N = 8; dist.FramePipeline)
    rgba8 = distributed and args.gather == "rgba8"
    tiles32 = [torch.zeros((slot_rows, W, 4), dtype=torch.float32, device=r.device) for _ in range(3)] if rgba8 else None

    def make_pipe(rotate, pl):
        """frame i is gathered to rank 0 — or, rotate: to rank i mod world (dist.FramePipeline) — and de-interleaved there"""
        sr = pl.slot_rows
        if rgba8:  # the shard's float4 tiles stay on its GPU (one buffer per frame slot); their 8-bit conversion travels
            return FramePipeline(pl, rank, (W, 4), torch.uint8, r.device, depth=3, multi_stream=True, rotate_root=rotate,
                                 finish=lambda g: frame_holder.__setitem__("f", r.deinterleave_rgba8(g, W, H, TILE_ROWS, world, sr)))
        return FramePipeline(pl, rank, (W, 4), torch.float32, r.device, depth=3, multi_stream=True, rotate_root=rotate,
                             finish=lambda g: frame_holder.__setitem__("f", r.deinterleave(g, W, H, TILE_ROWS, world, sr)))
    pipes = {"current": make_pipe(args.gather_root == "rotate", plan) if distributed else None, "rows": my_rows}
    submitted = [0]

    def step():
        if not distributed:
            frame_holder["f"] = r.render(tables, settings, W, H, out=mine)
            return
        pipe, rows = pipes["current"], pipes["rows"]
        if rgba8:
            buf = tiles32[submitted[0] % 3]
            submitted[0] += 1
            pipe.submit(lambda slot: r.tiles_to_rgba8(r.render_tiles(tables, settings, W, H, TILE_ROWS, rank, world, out=buf[:rows]),
                                                      out=slot[:rows]))
            return
        pipe.submit(lambda slot: r.render_tiles(tables, settings, W, H, TILE_ROWS, rank, world, out=slot[:rows]))

    def fence():
        if distributed:
            pipes["current"].drain()
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
    schedule = L.rm_debug_last_path()
    split_tiles = L.rm_debug_last_split()  # > 0: the launcher kept the light split for this picture (its heaviest tiles one light per workgroup)
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

    variants = {}
    if distributed and not args.no_variants and args.gather_root == "zero":
        # the same frames with the gather's root rotating over the ranks (frame i ends on rank i mod N; the plain t mod N deal): rank 0
        # is then no longer the one rank that receives N − 1 slots and de-interleaves the whole frame every frame; never `value`
        assert L.rm_set_root_relief(0) == 0
        plan0 = ShardPlan(H, TILE_ROWS, world)
        if rgba8 and plan0.slot_rows > tiles32[0].shape[0]:
            tiles32[:] = [torch.zeros((plan0.slot_rows, W, 4), dtype=torch.float32, device=r.device) for _ in range(3)]
        pipes["current"], pipes["rows"] = make_pipe(True, plan0), plan0.rows(rank)
        for _ in range(min(args.warmup, 3) + world):
            step()
        fence()
        t0r = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        dtr = torch.tensor([time.perf_counter() - t0r], dtype=torch.float64, device=r.device)
        dist.all_reduce(dtr, op=dist.ReduceOp.MAX)
        same = None
        if rank == 0 and "f" in frame_holder:  # the last frame rank 0 was the root of
            same = bool(torch.equal(frame_holder["f"], timed_frame))
        assert L.rm_set_root_relief(relief) == 0
        variants["rotating_root"] = {"value": round(W * H * args.steps / float(dtr.item()) / 1e6, 2), "unit": "Mpixels/s",
                                     "ms_per_step": round(float(dtr.item()) / args.steps * 1e3, 4), "steps": args.steps,
                                     "frame_identical_to_headline": same,
                                     "what": "FramePipeline(rotate_root=True), no root relief: frame i is gathered to and de-interleaved on rank i mod N"}
    single = not distributed and not args.no_variants
    nv = max(3, min(args.steps, 10))
    if single and schedule == 1 and cfg in ("c2", "c3", "c4", "c5"):
        # the same frame without the tile-order feedback (raster order: what a first frame, or a frame after a change of
        # size, costs); the headline's timed frames all ran with the previous frame's tile costs
        L.rm_set_tile_order(0)
        ms, k = timed_frames(r, L, fence, lambda: r.render(tables, settings, W, H, out=mine), nv)
        L.rm_set_tile_order(-1)
        variants["raster_tile_order"] = {"value": round(W * H / ms / 1e3, 2), "unit": "Mpixels/s", "ms_per_step": round(ms, 4),
                                         "kernel_ms": round(k, 4), "steps": nv,
                                         "what": "rm_set_tile_order(0): tiles start in raster order — a frame with no history"}
    if single and cfg == "c3" and args.bulb_eval == "reference":
        # the opt-in evaluation scheme of the same step, timed beside the headline (single GPU only; never `value`)
        vs = abi.default_settings(fractalIters=12, features=abi.RM_FEAT_REFERENCE_DEFAULT | abi.RM_FEAT_BULB_POWER8_ALGEBRAIC)
        ms, k = timed_frames(r, L, fence, lambda: r.render(tables, vs, W, H, out=mine), nv)
        variants["bulb_power8_algebraic"] = {
            "value": round(W * H / ms / 1e3, 2), "unit": "Mpixels/s", "ms_per_step": round(ms, 4), "kernel_ms": round(k, 4), "steps": nv,
            "what": "RM_FEAT_BULB_POWER8_ALGEBRAIC: w^8 by complex squarings instead of acos/atan/sin/cos/pow; same function, "
                    "|ΔDE| median 4e-8, 0.08 % of frame pixels differ by > 1e-3 from the headline frame"}
    if single and cfg in ("c1", "c2", "c3", "c4", "c5"):
        # three frames in flight, each on its own stream and into its own buffer: a frame's last straggler rays (a serial
        # chain of ≈0.7 ms; for the wavefront pipeline the tail of each of its dozen kernels) overlap the next frames' full
        # waves — what a caller that renders a sequence gets; never `value`
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
        variants["three_frames_in_flight"] = {"value": round(W * H * nf / df / 1e6, 2), "unit": "Mpixels/s", "ms_per_step": round(df / nf * 1e3, 4),
                                              "steps": nf, "frames_identical_to_headline": same,
                                              "what": "the same frames submitted round-robin on three HIP streams (three in flight)"}
        if cfg == "c3":
            variants["orbiting_camera"] = orbit_variant(r, L, fence, tables, settings, W, H, deg=float(os.environ.get("RM_ORBIT_DEG", "1.0")))
    if single and schedule == 5:
        # the same frame by the one-lane-per-pixel kernel (rm_set_kernel_path(1)); the headline ran the wavefront pipeline
        L.rm_set_kernel_path(1)
        ms, k = timed_frames(r, L, fence, lambda: r.render(tables, settings, W, H, out=mine), max(3, nv // 2))
        same = bool(torch.equal(mine.view(torch.int32), timed_frame.view(torch.int32)))
        L.rm_set_kernel_path(0)
        variants["one_lane_per_pixel_kernel"] = {"value": round(W * H / ms / 1e3, 2), "unit": "Mpixels/s", "ms_per_step": round(ms, 4),
                                                 "kernel_ms": round(k, 4), "frame_identical_to_headline": same,
                                                 "what": "rm_set_kernel_path(1): rm::render_kernel, an 8x8 pixel tile per wave"}
    if single and cfg == "c4":
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import helpers as h
        th = build_config("c4")[0]
        th.camera = h.make_camera((0, 500, 5), (0.3, 0.12, -1), (0, 1, 0), 70.0, W, H, far=2000.0)
        ms, k = timed_frames(r, L, fence, lambda: r.render(th, settings, W, H, out=mine), nv)
        variants["horizon_camera"] = {"value": round(W * H / ms / 1e3, 2), "unit": "Mpixels/s", "ms_per_step": round(ms, 4), "kernel_ms": round(k, 4),
                                      "what": "the same scene and layers with the camera turned to the horizon (the view the layers are made for)"}

    # work of a frame from its deterministic counters (outside the timed region): what the REFERENCE's formulation does
    # (algorithmic) and, for the plain classes, what the production kernel really executes (bit-identical shortcuts
    # honoured); then the shader clock the chip held under the bulb kernel's own load (stamped diagnostic build)
    if rank == 0:
        res_struct, _keep = r._resources(tables)
        frame_c = torch.empty((H, W, 4), dtype=torch.float32, device=r.device)

        def counted(mode):
            c = abi.RmCounters()
            torch.cuda.synchronize(r.device)
            st = L.rm_render_counted_res(*tables.args(settings), C.byref(res_struct), W, H, 0, H, C.c_void_p(frame_c.data_ptr()), None, mode, C.byref(c))
            assert st == 0, L.rm_last_error()
            return c
        cnt = counted(abi.RM_COUNT_REFERENCE)
        counted_same = (bool(torch.equal(r.to_rgba8(frame_c), timed_frame)) if rgba8 else
                        bool(torch.equal(frame_c.view(torch.int32), timed_frame.view(torch.int32))))
        plain = not (settings.features & (abi.RM_FEAT_TERRAIN | abi.RM_FEAT_CLOUD | abi.RM_FEAT_SEA | abi.RM_FEAT_SKY_BACKGROUND
                                          | abi.RM_FEAT_NIGHTSKY_BACKGROUND)) and not tables.textures
        cnt_exec = counted(abi.RM_COUNT_EXECUTED) if plain else None
        clock_mhz = None
        if cfg == "c3":
            for _ in range(5):
                r.render(tables, settings, W, H, out=mine)
            _, clock_mhz = r.render_clocked(tables, settings, W, H)

        mpix = W * H * args.steps / dt / 1e6
        flops_frame, parts, model = flop_model(tables, settings, cnt)
        flops_exec = flop_model(tables, settings, cnt_exec, executed=True)[0] if cnt_exec is not None else None
        flops_written = flop_model(tables, settings, cnt, as_written=True)[0]
        kernel_names = {1: "rm::render_kernel<BULB,COUNT=0,ENV,TEX> (one lane per pixel, 8x8 tile per wave)",
                        5: "wavefront pipeline: per ray generation rm::wf_march_kernel<0|1> (persistent waves, lanes = rays, refilled), "
                           "wf_surface_kernel, wf_march_kernel<2> (shadow rays), wf_light_kernel"}
        # with tile-order feedback a launch is two sort kernels (stage 0, ~0.02 ms) + the render kernel (stage 1): the roofline is
        # the render kernel's, `launch_ms` stays the whole launch
        ordered = schedule == 1 and stages[0] > 0.0
        render_ms = stages[1] if stages[1] > 0.0 else kernel_ms
        # with several frames in flight (N > 1) the launches overlap and their event spans are not kernel time: the roofline is
        # then taken over the wall time one frame of this rank's shard costs
        secs = (dt / args.steps if distributed else render_ms * 1e-3)
        flops_launch = flops_frame / world  # the dominant kernel(s) of one launch process 1/world of the frame (interleaved tiles ≈ equal work)
        achieved = flops_launch / secs / 1e12 if secs > 0 else 0.0
        bytes_launch = W * H * 16 / world
        roof = {"bound": "valu", "achieved": round(achieved, 3), "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_FP32_TFLOPS, 4),
                "kernel": kernel_names.get(schedule, str(schedule)), "kernel_ms": round(render_ms, 4), "launch_ms": round(kernel_ms, 4),
                "time_base": ("wall time per frame (three overlapping frames in flight: event spans of single launches are longer than "
                              "their share of the GPU)" if distributed else
                              ("HIP events around the render kernel" if schedule == 1 else "HIP events around the launch's kernels")),
                "algorithmic": {"flop_per_launch": flops_launch, "sceneEvals": cnt.sceneEvals, "bulbIters": cnt.bulbIters,
                                "hitPixels": cnt.hitPixels, "shadedPoints": cnt.shadedPoints, "terrainEvals": cnt.terrainEvals,
                                "cloudEvals": cnt.cloudEvals, "flop_by_part": parts, "model": model,
                                "counted_frame_identical_to_timed_frame": counted_same},
                "hbm": {"achieved": round(bytes_launch / secs / 1e9, 2) if secs > 0 else 0.0, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "bytes_per_pixel": 16, "what": "algorithmic bytes (one float4 store per pixel) over the same time"}}
        if ordered:
            roof["stage_ms"] = {"tile_order_sort": round(stages[0], 4), "render_kernel": round(stages[1], 4)}
        if split_tiles > 0:
            roof["light_split_tiles"] = split_tiles
        if flops_exec is not None:
            # executed = the work the one-lane-per-pixel kernel really does (bounding-ball culls, no shadow march for dropped lights)
            executed = flops_exec / world / secs / 1e12 if secs > 0 else 0.0
            roof["executed"] = {"achieved": round(executed, 3), "frac": round(executed / PEAK_FP32_TFLOPS, 4),
                                "sceneEvals": cnt_exec.sceneEvals, "bulbIters": cnt_exec.bulbIters, "shapeEvals": cnt_exec.shapeEvals,
                                "shapes_per_evaluation": round(cnt_exec.shapeEvals / max(cnt_exec.sceneEvals, 1), 3),
                                "flop_per_launch": flops_exec / world}
        if flops_written != flops_frame:
            w = flops_written / world / secs / 1e12 if secs > 0 else 0.0
            roof["shader_as_written"] = {"achieved": round(w, 3), "frac": round(w / PEAK_FP32_TFLOPS, 4), "flop_per_launch": flops_written / world,
                                         "what": "the same units priced with what the shader text computes per lane but no implementation would: the "
                                                 "Menger sponge's launch-uniform prologue (40 flop per evaluation) and, at iTime = 0, its identity "
                                                 "rotation mix (27 flop per level); round 3 reported this as `frac`"}
        pm = PMC_MEASURED.get(cfg)
        if pm:
            # a utilisation figure that needs no flop model: wave-level VALU instructions x live lanes over the issue slots of the
            # profiled launch (rocprofv3 PMC passes over this command; profiled clocks run a few percent lower)
            roof["valu_issue"] = {"frac": pm["valu_issue_frac"], "lanes_live": pm["lanes_live"], "cycles_per_valu_instr_per_simd": pm["cycles_per_valu"],
                                  "resident_waves_per_simd": pm["resident_waves"], "source": pm["source"]}
        if cfg == "c3":
            roof["slots"] = {"algorithmic_frac": round(slots_model(cnt) / world / secs / PEAK_LANE_SLOTS, 4) if secs > 0 else 0.0,
                             "executed_frac": round(slots_model(cnt_exec) / world / secs / PEAK_LANE_SLOTS, 4) if secs > 0 else 0.0,
                             "peak_lane_slots_per_s": PEAK_LANE_SLOTS}
            roof["shader_clock_mhz"] = round(clock_mhz, 1) if clock_mhz else None
            roof["clock_source"] = ("s_memtime / s_memrealtime stamps of every wave of one launch of the same kernel (diagnostic build, "
                                    "rm_render_clocked) after back-to-back launches")
        tm = TRAFFIC_MEASURED.get(cfg)
        roof["traffic"] = tm["bytes_per_launch"] if tm else None
        roof["traffic_source"] = (tm["source"] if tm else "not measured: HBM counters cannot be collected from inside this run")
        line = {
            "metric": desc["metric"],
            "value": round(mpix, 2), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc["workload"], "baseline_config": desc["baseline_config"], "name": cfg,
                       "rows": "whole frame" if not distributed else f"{TILE_ROWS}-row tiles round-robin over {world} GPUs; three frames in "
                               "flight per GPU on three streams (renders of consecutive frames overlap, RCCL gather of frame i "
                               + (f"to rank 0 runs under later renders); every frame de-interleaved on rank 0; root relief {relief} "
                                  "(rm_set_root_relief: rank 0 renders fewer tiles)" if args.gather_root == "zero" else
                                  "to rank i mod N runs under later renders); every frame de-interleaved on its root")
                               + (" as the RGBA8 image (4 B/pixel gathered)" if rgba8 else " as a float4 frame (16 B/pixel gathered)"),
                       "tile_order": ("feedback: each frame records its tiles' shader-cycle costs and the next frame of the SAME picture starts heavy "
                                      "tiles first; a new picture (first frame, moved camera) is ordered by a geometric classification instead "
                                      "(same pixels, same work; variants.raster_tile_order = plain raster order, variants.orbiting_camera = "
                                      "every frame a new picture)" if ordered else "not used by this schedule"),
                       "parity": "bit-exact vs CPU oracle (rm_math contract); vs the reference shader: unpinned, cross-checked (DESIGN §2)"},
            "roofline": roof,
            "variants": variants,
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"], line["parity_check"] = cpu_baseline(tables, settings, W, H, desc["cpu_rows"], timed_frame)
        print(json.dumps(line), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
