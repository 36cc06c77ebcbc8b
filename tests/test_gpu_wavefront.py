"""The wavefront pipeline (rm_wavefront.hip.h, rm_set_kernel_path(5)) against the CPU oracle and against the
one-lane-per-pixel kernel: a third schedule of the same per-ray arithmetic, so the bar is bit equality."""
import os

import numpy as np
import pytest

import helpers as h
import test_gpu_parity as tg
from raymarcher_amd import abi

pytestmark = pytest.mark.gpu


def render_path(renderer, path, t, s, W, H, **kw):
    """Render with a forced schedule and report the schedule that really ran."""
    from raymarcher_amd import lib
    try:
        assert lib().rm_set_kernel_path(path) == 0
        out = renderer.render(t, s, W, H, **kw)
        ran = lib().rm_debug_last_path()
    finally:
        lib().rm_set_kernel_path(0)
    return out, ran


def reflective_scene(W, H):
    """tg.reflect_refract_scene without the transparent material (refraction stays on rm::render_kernel)."""
    cam, objs, no, lights, nl, g = tg.reflect_refract_scene(W, H)
    for k in range(3):
        objs[1].cTransparent[k] = 0.0
        objs[1].cReflective[k] = 0.5
    return cam, objs, no, lights, nl, g


def primitives_scene(W, H):
    """tg.all_primitives_scene with a sphere in place of its Mandelbulb (data-dependent evaluation cost: not this pipeline's class)."""
    cam, objs, no, lights, nl, g = tg.all_primitives_scene(W, H)
    for i in range(no):
        if objs[i].type == abi.RM_MANDELBULB:
            objs[i].type = abi.RM_SPHERE
    return cam, objs, no, lights, nl, g


WF_CASES = {
    "primitives_phong_64steps": (primitives_scene, {"maxSteps": 64}, 96, 64),
    "primitives_softshadow_ao_nobump": (primitives_scene,
                                        {"enableSoftShadow": 1, "enableAmbientOcclusion": 1, "features": abi.RM_FEAT_DARK_BACKGROUND}, 80, 48),
    "reflective_3_bounces": (reflective_scene, {"enableReflection": 1, "numReflection": 3}, 96, 64),
    "reflective_soft_ao_ragged": (reflective_scene, {"enableReflection": 1, "numReflection": 2, "enableSoftShadow": 1,
                                                     "enableAmbientOcclusion": 1}, 101, 37),
    "reflection_enabled_zero_bounces": (reflective_scene, {"enableReflection": 1, "numReflection": 0}, 64, 40),
    "menger_5_levels_2_bounces": (lambda W, H: tg.menger_scene(W, H), {"mengerLevels": 5, "numReflection": 2, "enableReflection": 1}, 80, 60),
    "menger_animated": (lambda W, H: tg.menger_scene(W, H)[:5] + (h.make_globals(itime=7.5),),
                        {"mengerLevels": 4, "numReflection": 1, "enableReflection": 1}, 80, 60),
    "one_step": (primitives_scene, {"maxSteps": 1}, 40, 24),
}


@pytest.mark.parametrize("name", list(WF_CASES))
def test_wavefront_frames_bit_exact(renderer, name):
    build, over, W, H = WF_CASES[name]
    scene = build(W, H)
    s = abi.default_settings(**over)
    ref, ref_b = h.oracle_render(scene, s, W, H, bright=True)
    (out, br), ran = render_path(renderer, 5, tg.tables_of(scene), s, W, H, bright=True)
    assert ran == 5, "the wavefront pipeline must be the schedule that ran"
    tg.assert_bit_equal(out.cpu().numpy(), ref, f"{name} fragColor")
    tg.assert_bit_equal(br.cpu().numpy(), ref_b, f"{name} BrightColor")


def test_wavefront_falls_back_where_it_does_not_apply(renderer):
    """Refraction through a transparent object, a Mandelbulb in the table, procedural layers: rm_set_kernel_path(5) renders
    them with rm::render_kernel — and still the oracle's bits."""
    W, H = 64, 40
    cases = [(tg.reflect_refract_scene(W, H), abi.default_settings(enableReflection=1, enableRefraction=1)),
             (h.scene_mandelbulb(W, H), abi.default_settings(fractalIters=8)),
             (tg.env_scene(W, H), abi.default_settings(features=tg.ENV_ALL))]
    for scene, s in cases:
        out, ran = render_path(renderer, 5, tg.tables_of(scene), s, W, H)
        assert ran == 1
        tg.assert_bit_equal(out.cpu().numpy(), h.oracle_render(scene, s, W, H), "fallback")


def _random_wf_case(rng, W, H):
    """A random scene inside the wavefront pipeline's class: any table of constant-cost objects, the three plain light kinds,
    every shading option except refraction."""
    f = rng.uniform
    types = [abi.RM_CUBE, abi.RM_CONE, abi.RM_CYLINDER, abi.RM_SPHERE, abi.RM_OCTAHEDRON, abi.RM_TORUS, abi.RM_CAPSULE,
             abi.RM_DEATHSTAR, abi.RM_RECTANGLE, abi.RM_MENGERSPONGE, abi.RM_SIERPINSKI]
    objs = []
    for _ in range(int(rng.integers(1, 8))):
        ty = int(rng.choice(types))
        sc = float(f(0.6, 1.8))
        sx, sy, sz = (sc * float(f(0.8, 1.25)) for _ in range(3))
        M = h.translate(f(-2.2, 2.2), f(-1.0, 1.2), f(-2.5, 1.0)) @ tg.rot_x(f(-0.6, 0.6)) @ h.scale(sx, sy, sz)
        objs.append(h.make_object(ty, model=M, scale_factor=min(sx, sy, sz), ambient=tuple(f(0, .3, 3)), diffuse=tuple(f(.2, 1, 3)),
                                  specular=tuple(f(0, 1, 3)), shininess=float(rng.choice([0, 1, 7.5, 25, 100])),
                                  reflective=tuple(f(0, .8, 3)) if f() < 0.5 else (0, 0, 0),
                                  transparent=tuple(f(0, .8, 3)) if f() < 0.3 else (0, 0, 0), ior=float(f(1.05, 1.6))))
    lights = []
    for _ in range(int(rng.integers(0, 4))):
        kind = int(rng.integers(0, 3))
        col = tuple(f(.3, 1.6, 3))
        if kind == abi.RM_LIGHT_DIRECTIONAL:
            lights.append(h.make_light(kind, col, direction=(f(-1, 1), f(-1, -0.2), f(-1, 1))))
        elif kind == abi.RM_LIGHT_POINT:
            lights.append(h.make_light(kind, col, pos=(f(-4, 4), f(1, 5), f(-1, 5)), func=(f(.5, 1), f(0, .1), f(0, .02))))
        else:
            lights.append(h.make_light(kind, col, direction=(f(-.3, .3), -1, f(-.6, 0)), pos=(f(-2, 2), f(3, 5), f(0, 3)),
                                       func=(f(.5, 1), f(0, .1), 0), angle=float(f(.4, .9)), penumbra=float(f(.05, .3))))
    feats = int(rng.choice([abi.RM_FEAT_WHITE_BACKGROUND, abi.RM_FEAT_DARK_BACKGROUND, 0]))
    if f() < 0.6:
        feats |= abi.RM_FEAT_PERLIN_BUMP
    s = abi.default_settings(features=feats, enableSoftShadow=int(f() < 0.4), enableAmbientOcclusion=int(f() < 0.4),
                             enableReflection=int(f() < 0.7), enableRefraction=0,
                             maxSteps=int(rng.choice([32, 128, 256])), mengerLevels=int(rng.choice([3, 4, 5])),
                             numReflection=int(rng.choice([1, 2, 3, 5])))
    g = h.make_globals(ka=f(.2, .8), kd=f(.3, 1), ks=f(.2, 1), kt=f(.2, 1), itime=float(f(0, 9)))
    cam = h.make_camera((f(-1, 1), f(0.5, 2.5), f(4.5, 6.5)), (f(-.15, .15), f(-.45, -.05), -1), (0, 1, 0), float(f(35, 60)), W, H)
    nl = len(lights)
    lights = lights or [h.make_light(abi.RM_LIGHT_POINT)]
    return (cam, (abi.RmObject * len(objs))(*objs), len(objs), (abi.RmLight * len(lights))(*lights), nl, g), s


def test_wavefront_random_scenes_bit_exact(renderer):
    """Seeded random scenes of the pipeline's class: wavefront = oracle = one lane per pixel, fragColor and BrightColor."""
    W, H = 56, 40
    rng = np.random.default_rng(int(os.environ.get("RM_FUZZ_SEED", "20261004")))
    for i in range(int(os.environ.get("RM_FUZZ_CASES", "24"))):
        scene, s = _random_wf_case(rng, W, H)
        ref, ref_b = h.oracle_render(scene, s, W, H, bright=True)
        (out, br), ran = render_path(renderer, 5, tg.tables_of(scene), s, W, H, bright=True)
        assert ran == 5
        tg.assert_bit_equal(out.cpu().numpy(), ref, f"random wavefront scene {i}")
        tg.assert_bit_equal(br.cpu().numpy(), ref_b, f"random wavefront scene {i} bright")
        (mono, _), ran1 = render_path(renderer, 1, tg.tables_of(scene), s, W, H, bright=True)
        assert ran1 == 1 and tg._ieq(mono, out)


def test_wavefront_row_ranges_tiles_and_streams(renderer):
    """Row ranges and interleaved row tiles (the multi-GPU shards) through the pipeline, and two frames in flight on two
    streams (scratch is per stream): the same bits as the single whole-frame launch."""
    import torch
    from raymarcher_amd import lib
    W, H = 150, 83
    scene = tg.menger_scene(W, H)
    t = tg.tables_of(scene)
    s = abi.default_settings(mengerLevels=4, numReflection=2, enableReflection=1)
    full, ran = render_path(renderer, 5, t, s, W, H)
    assert ran == 5
    tg.assert_bit_equal(full.cpu().numpy(), h.oracle_render(scene, s, W, H), "whole frame")
    part, _ = render_path(renderer, 5, t, s, W, H, row_begin=17, row_end=60)
    assert tg._ieq(part, full[17:60])
    try:
        lib().rm_set_kernel_path(5)
        for N in (2, 3):
            for k in range(N):
                mine = renderer.render_tiles(t, s, W, H, 8, k, N)
                rows = [lib().rm_shard_row_to_frame(H, 8, k, N, i) for i in range(mine.shape[0])]
                assert tg._ieq(mine, full[torch.tensor(rows, device=full.device)])
        streams = [torch.cuda.Stream(device=renderer.device) for _ in range(2)]
        outs = []
        for st in streams * 2:
            with torch.cuda.stream(st):
                outs.append(renderer.render(t, s, W, H))
        torch.cuda.synchronize()
        assert all(tg._ieq(o, full) for o in outs)
    finally:
        lib().rm_set_kernel_path(0)


def test_wavefront_config5_bands_8k(renderer):
    """BASELINE configs[4] (unit_mengersponge.json, 5 levels, 2 bounces, 7680×4320) through the pipeline: identical to the
    one-lane-per-pixel frame everywhere, and to the oracle on two bands of rows."""
    from raymarcher_amd import Scene
    W, H = 7680, 4320
    t = Scene(path=os.path.join(tg.SCENES, "simple", "unit_mengersponge.json")).tables(W, H)
    s = abi.default_settings(mengerLevels=5, numReflection=2, enableReflection=1)
    wf, ran = render_path(renderer, 5, t, s, W, H)
    assert ran == 5
    mono, ran1 = render_path(renderer, 1, t, s, W, H)
    assert ran1 == 1 and tg._ieq(wf, mono)
    for r0 in (1000, 2164):
        ref = h.oracle_render(tg._scene_tuple(t), s, W, H, r0, r0 + 4, threads=16)
        tg.assert_bit_equal(wf[r0:r0 + 4].cpu().numpy(), ref, f"8K rows {r0}..{r0 + 4}")


def test_wavefront_workspace_refused_falls_back_to_the_pixel_kernel(renderer):
    """The auto-selected pipeline needs ≈175 B of grow-only scratch per pixel (per device and stream).  When that buffer may
    not be had — rm_set_workspace_limit here; a failing hipMalloc takes the same branch — a launch that chose the pipeline by
    itself renders with rm::render_kernel instead (identical bits, no workspace); only an explicit rm_set_kernel_path(5)
    reports the failure, and HIP's error state stays clean either way (rm_kernels.hip launch_render / stream_workspace)."""
    import ctypes as C
    import torch
    from raymarcher_amd import Scene, lib
    from raymarcher_amd._lib import RaymarcherError
    L = lib()
    W, H = 2560, 1664  # 4.26 M pixels: above the pipeline's 2^22-pixel threshold
    t = Scene(path=os.path.join(tg.SCENES, "simple", "unit_mengersponge.json")).tables(W, H)
    s = abi.default_settings(mengerLevels=5, numReflection=2, enableReflection=1)
    need = 150 * W * H  # a lower bound of what the pipeline asks for
    freed = C.c_ulonglong(0)
    assert L.rm_release_workspaces(C.byref(freed)) == 0  # whatever earlier tests left on this device
    try:
        assert L.rm_set_workspace_limit(512 << 20) == 0  # the frame needs 1.1 GB; a small frame's 0.36 GB (one slot chunk per persistent wave) fits
        capped, ran = render_path(renderer, 0, t, s, W, H)
        assert ran == 1, "a refused workspace must fall back to the one-lane-per-pixel kernel"
        again, ran2 = render_path(renderer, 0, t, s, W, H)  # the refusal is remembered: no second attempt, same frame
        assert ran2 == 1 and tg._ieq(again, capped)
        with pytest.raises(RaymarcherError):  # an explicit request reports instead of silently switching
            render_path(renderer, 5, t, s, W, H)
        torch.cuda.synchronize()  # no sticky HIP error: torch's own calls on this device still succeed
        small, ran3 = render_path(renderer, 5, t, s, 96, 64)  # a frame whose workspace fits the limit still takes the pipeline
        assert ran3 == 5
    finally:
        assert L.rm_set_workspace_limit(0) == 0
    wf, ran4 = render_path(renderer, 0, t, s, W, H)  # limit lifted: the refusal is forgotten, the pipeline runs
    assert ran4 == 5 and tg._ieq(wf, capped)
    for r0 in (400, 830):
        ref = h.oracle_render(tg._scene_tuple(t), s, W, H, r0, r0 + 4, threads=16)
        tg.assert_bit_equal(capped[r0:r0 + 4].cpu().numpy(), ref, f"capped frame rows {r0}..{r0 + 4}")
    assert L.rm_release_workspaces(C.byref(freed)) == 0 and freed.value >= need
    after, ran5 = render_path(renderer, 0, t, s, W, H)  # buffers come back on demand
    assert ran5 == 5 and tg._ieq(after, wf)
