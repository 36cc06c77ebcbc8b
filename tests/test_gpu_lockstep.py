"""The lockstep instantiation of the table-walk kernel (rm_device.hip.h shadowLockstep / sdSceneK; rm_set_lockstep): the shadow
rays of up to three lights of a shading point share one walk of the object table per step.  The same evaluations per ray and the
same sums in light order as getPhong's light loop (frag:1864-1930), so the bar is bit equality — with the CPU oracle and with
the kernel that marches light after light."""
import os

import numpy as np
import pytest

import helpers as h
import test_gpu_parity as tg
from raymarcher_amd import abi
from raymarcher_amd.render import Scene

pytestmark = pytest.mark.gpu


def render_lockstep(renderer, mode, t, s, W, H, **kw):
    """Render with lockstep forced on / off (one-lane-per-pixel kernel) and report whether the lockstep instantiation ran."""
    from raymarcher_amd import lib
    try:
        assert lib().rm_set_lockstep(mode) == 0
        assert lib().rm_set_kernel_path(1) == 0
        out = renderer.render(t, s, W, H, **kw)
        ran = lib().rm_debug_last_lockstep()
    finally:
        lib().rm_set_kernel_path(0)
        lib().rm_set_lockstep(-1)
    return out, ran


def scenefile(rel, W, H):
    return Scene(path=os.path.join(tg.SCENES, rel)).tables(W, H, load_textures=False)


# scenefile of the reference, settings, W, H: three directional lights (C2), three spot lights, three point lights, one
# directional + two point, four and five directional (groups of 3 + 1 and 3 + 2), nine lights (three groups)
FILE_CASES = {
    "c2_soft_ao": ("lighting/directional_light_2.json", {"enableSoftShadow": 1, "enableAmbientOcclusion": 1}, 160, 90),
    "c2_hard": ("lighting/directional_light_2.json", {}, 96, 54),
    "spot3_soft": ("lighting/spot_light_2.json", {"enableSoftShadow": 1}, 96, 54),
    "point3_ao": ("lighting/point_light_2.json", {"enableAmbientOcclusion": 1}, 96, 54),
    "dir_point_point_all_options": ("simple/phong_total.json", {"enableSoftShadow": 1, "enableAmbientOcclusion": 1,
                                                                "enableReflection": 1, "enableRefraction": 1}, 96, 54),
    "dir4_reflection": ("lighting/reflections_basic.json", {"enableReflection": 1, "numReflection": 2}, 96, 54),
    "dir5_reflection_soft": ("lighting/reflections_complex.json", {"enableReflection": 1, "enableSoftShadow": 1}, 96, 54),
    "nine_lights": ("lighting/shadow_test.json", {"enableSoftShadow": 1}, 80, 45),
    "two_lights_refraction": ("lighting/refract2.json", {"enableReflection": 1, "enableRefraction": 1}, 80, 45),
    "one_step": ("lighting/directional_light_2.json", {"maxSteps": 1, "enableSoftShadow": 1}, 48, 27),
    "no_steps": ("lighting/directional_light_2.json", {"maxSteps": 0}, 48, 27),
}


@pytest.mark.parametrize("name", list(FILE_CASES))
def test_lockstep_frames_bit_exact(renderer, name):
    rel, over, W, H = FILE_CASES[name]
    t = scenefile(rel, W, H)
    for i in range(t.num_objects):  # the images are not loaded: untextured objects (the sampler kernels are another class)
        t.objects[i].texLoc = -1
    s = abi.default_settings(**over)
    ref, ref_b = h.oracle_render(tg._scene_tuple(t), s, W, H, bright=True)
    (out, br), ran = render_lockstep(renderer, 1, t, s, W, H, bright=True)
    assert ran == 1, "the lockstep instantiation must be the kernel that ran"
    tg.assert_bit_equal(out.cpu().numpy(), ref, f"{name} fragColor")
    tg.assert_bit_equal(br.cpu().numpy(), ref_b, f"{name} BrightColor")
    (seq, seq_b), ran0 = render_lockstep(renderer, 0, t, s, W, H, bright=True)
    assert ran0 == 0 and tg._ieq(seq, out) and tg._ieq(seq_b, br)


def test_lockstep_class_boundaries(renderer):
    """One light, a fractal in the table, an area light, procedural layers, textures: rendered by the other instantiations."""
    from raymarcher_amd import lib
    W, H = 64, 40
    t1 = scenefile("lighting/directional_light_1.json", W, H)
    cases = [(t1, abi.default_settings()),
             (tg.tables_of(tg.menger_scene(W, H)), abi.default_settings()),
             (tg.tables_of(h.scene_mandelbulb(W, H)), abi.default_settings(fractalIters=8)),
             (tg.tables_of(tg.env_scene(W, H)), abi.default_settings(features=tg.ENV_ALL))]
    for t, s in cases:
        out, ran = render_lockstep(renderer, 1, t, s, W, H)
        assert ran == 0
        tg.assert_bit_equal(out.cpu().numpy(), h.oracle_render(tg._scene_tuple(t), s, W, H), "outside the class")
    al = tg.resource_case("area_light", W, H)
    ta = tg.tables_of(al[0])
    ta.ltc1, ta.ltc2 = al[2]["ltc1"], al[2]["ltc2"]
    _, ran = render_lockstep(renderer, 1, ta, al[1], W, H)
    assert ran == 0
    tt = Scene(path=os.path.join(tg.SCENES, "simple", "unit_sphere.json")).tables(W, H)  # three spot lights, textured floor
    out, ran = render_lockstep(renderer, 1, tt, abi.default_settings(maxSteps=64), W, H)
    assert ran == 0
    assert lib().rm_set_lockstep(2) != 0 and lib().rm_set_lockstep(-2) != 0


def _random_lockstep_case(rng, W, H):
    """A random scene of the class: constant-cost primitives, two to ten lights of the three plain kinds (some facing away from
    most of the scene, so that groups form from the lights a wave really marches), every shading option."""
    f = rng.uniform
    types = [abi.RM_CUBE, abi.RM_CONE, abi.RM_CYLINDER, abi.RM_SPHERE, abi.RM_OCTAHEDRON, abi.RM_TORUS, abi.RM_CAPSULE,
             abi.RM_DEATHSTAR, abi.RM_RECTANGLE]
    objs = []
    for _ in range(int(rng.integers(1, 9))):
        ty = int(rng.choice(types))
        sc = float(f(0.6, 1.8))
        sx, sy, sz = (sc * float(f(0.8, 1.25)) for _ in range(3))
        M = h.translate(f(-2.2, 2.2), f(-1.0, 1.2), f(-2.5, 1.0)) @ tg.rot_x(f(-0.6, 0.6)) @ h.scale(sx, sy, sz)
        objs.append(h.make_object(ty, model=M, scale_factor=min(sx, sy, sz), ambient=tuple(f(0, .3, 3)), diffuse=tuple(f(.2, 1, 3)),
                                  specular=tuple(f(0, 1, 3)), shininess=float(rng.choice([0, 1, 7.5, 25, 100])),
                                  reflective=tuple(f(0, .8, 3)) if f() < 0.4 else (0, 0, 0),
                                  transparent=tuple(f(0, .8, 3)) if f() < 0.3 else (0, 0, 0), ior=float(f(1.05, 1.6))))
    if f() < 0.5:  # a floor: long grazing shadow rays
        objs.append(h.make_object(abi.RM_CUBE, model=h.translate(0, -1.6, -1) @ h.scale(9, 0.2, 9), scale_factor=0.2,
                                  diffuse=(.7, .7, .7), ambient=(.1, .1, .1)))
    lights = []
    for _ in range(int(rng.choice([2, 2, 3, 3, 3, 4, 5, 6, 7, 10]))):
        kind = int(rng.integers(0, 3))
        col = tuple(f(.2, 1.2, 3))
        if kind == abi.RM_LIGHT_DIRECTIONAL:
            lights.append(h.make_light(kind, col, direction=(f(-1, 1), f(-1, 0.6), f(-1, 1))))
        elif kind == abi.RM_LIGHT_POINT:
            lights.append(h.make_light(kind, col, pos=(f(-4, 4), f(-1, 5), f(-3, 5)), func=(f(.5, 1), f(0, .1), f(0, .02))))
        else:
            lights.append(h.make_light(kind, col, direction=(f(-.3, .3), -1, f(-.6, 0)), pos=(f(-2, 2), f(3, 5), f(0, 3)),
                                       func=(f(.5, 1), f(0, .1), 0), angle=float(f(.4, .9)), penumbra=float(f(.05, .3))))
    feats = int(rng.choice([abi.RM_FEAT_WHITE_BACKGROUND, abi.RM_FEAT_DARK_BACKGROUND, 0]))
    if f() < 0.5:
        feats |= abi.RM_FEAT_PERLIN_BUMP
    s = abi.default_settings(features=feats, enableSoftShadow=int(f() < 0.5), enableAmbientOcclusion=int(f() < 0.4),
                             enableReflection=int(f() < 0.4), enableRefraction=int(f() < 0.3),
                             maxSteps=int(rng.choice([16, 64, 256])), numReflection=int(rng.choice([1, 2, 3])))
    g = h.make_globals(ka=f(.2, .8), kd=f(.3, 1), ks=f(.2, 1), kt=f(.2, 1))
    cam = h.make_camera((f(-1, 1), f(0.5, 2.5), f(4.5, 6.5)), (f(-.15, .15), f(-.45, -.05), -1), (0, 1, 0), float(f(35, 60)), W, H)
    return (cam, (abi.RmObject * len(objs))(*objs), len(objs), (abi.RmLight * len(lights))(*lights), len(lights), g), s


def test_lockstep_random_scenes_bit_exact(renderer):
    """Seeded random scenes of the class: lockstep = oracle = light after light, fragColor and BrightColor."""
    W, H = 56, 40
    rng = np.random.default_rng(int(os.environ.get("RM_FUZZ_SEED", "20261005")))
    for i in range(int(os.environ.get("RM_FUZZ_CASES", "32"))):
        scene, s = _random_lockstep_case(rng, W, H)
        ref, ref_b = h.oracle_render(scene, s, W, H, bright=True)
        (out, br), ran = render_lockstep(renderer, 1, tg.tables_of(scene), s, W, H, bright=True)
        assert ran == 1
        tg.assert_bit_equal(out.cpu().numpy(), ref, f"random lockstep scene {i}")
        tg.assert_bit_equal(br.cpu().numpy(), ref_b, f"random lockstep scene {i} bright")
        (seq, _), ran0 = render_lockstep(renderer, 0, tg.tables_of(scene), s, W, H, bright=True)
        assert ran0 == 0 and tg._ieq(seq, out)


def test_lockstep_counters_and_row_tiles(renderer):
    """The executed-work counters do not depend on the schedule (they come from the counting instantiation, which marches light
    after light), and row ranges / interleaved row tiles of a lockstep frame are the rows of the whole frame."""
    import torch
    from raymarcher_amd import lib
    W, H = 150, 83
    t = scenefile("lighting/directional_light_2.json", W, H)
    s = abi.default_settings(enableSoftShadow=1, enableAmbientOcclusion=1)
    full, ran = render_lockstep(renderer, 1, t, s, W, H)
    assert ran == 1
    tg.assert_bit_equal(full.cpu().numpy(), h.oracle_render(tg._scene_tuple(t), s, W, H), "whole frame")
    part, _ = render_lockstep(renderer, 1, t, s, W, H, row_begin=17, row_end=60)
    assert tg._ieq(part, full[17:60])
    try:
        lib().rm_set_lockstep(1)
        for k in range(3):
            mine = renderer.render_tiles(t, s, W, H, 8, k, 3)
            rows = [lib().rm_shard_row_to_frame(H, 8, k, 3, i) for i in range(mine.shape[0])]
            assert tg._ieq(mine, full[torch.tensor(rows, device=full.device)])
    finally:
        lib().rm_set_lockstep(-1)
    frame, c = renderer.render_counted(t, s, W, H, abi.RM_COUNT_EXECUTED)
    assert tg._ieq(frame, full) and c.sceneEvals > 0
