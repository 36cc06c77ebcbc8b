"""Known-answer tests that pin the CPU oracle's restatement of the shader (no GPU).

The reference has no tests or golden vectors, so these are analytic values of the distance
functions it implements (iquilezles.org SDF definitions at the unit sizes of sdMatch, frag:1262-1293) and
algebraic properties of the marchers."""
import ctypes as C

import os

import numpy as np
import pytest

import helpers as h
from raymarcher_amd import abi


def sd(type_, pts, model=None, scale_factor=1.0, settings=None, g=None):
    objs = (abi.RmObject * 1)(h.make_object(type_, model=model, scale_factor=scale_factor))
    pts = np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, 3)
    out = np.empty((len(pts), 4), dtype=np.float32)
    s = settings or abi.default_settings()
    g = g or h.make_globals()
    assert h.oracle().rmo_probe_sdscene(objs, 1, C.byref(g), C.byref(s), h.fptr(pts), h.fptr(out), len(pts)) == 0
    return out


def test_sphere_cube_analytic():
    d = sd(abi.RM_SPHERE, [[1, 0, 0], [0, 0, 0], [0, 2, 0], [0.3, 0.4, 0]])[:, 0]
    assert np.allclose(d, [0.5, -0.5, 1.5, 0.0], atol=1e-7)  # sdSphere((1,0,0), .5) = .5
    d = sd(abi.RM_CUBE, [[1, 0, 0], [0, 0, 0], [1.5, 1.5, 0], [0.5, 0.5, 0.5], [1.5, 1.5, 1.5]])[:, 0]
    assert np.allclose(d, [0.5, -0.5, np.sqrt(2.0), 0.0, np.sqrt(3.0)], atol=1e-6)
    d = sd(abi.RM_RECTANGLE, [[0, 0, 1], [1, 0, 0], [0, 0, 0]])[:, 0]
    assert np.allclose(d, [1.0, 0.5, 0.0], atol=1e-7)


def test_cylinder_torus_capsule_octahedron_cone_analytic():
    d = sd(abi.RM_CYLINDER, [[1, 0, 0], [0, 1, 0], [0, 0, 0], [1.5, 1.5, 0]])[:, 0]
    assert np.allclose(d, [0.5, 0.5, -0.5, np.sqrt(2.0)], atol=1e-6)
    d = sd(abi.RM_TORUS, [[0.5, 0, 0], [0, 0, 0], [0.5, 1, 0], [1.5, 0, 0]])[:, 0]
    assert np.allclose(d, [-0.125, 0.375, 0.875, 0.875], atol=1e-6)
    d = sd(abi.RM_CAPSULE, [[0, 0.25, 0], [0.5, 0.25, 0], [0, 1.0, 0], [0, -0.5, 0]])[:, 0]
    assert np.allclose(d, [-0.1, 0.4, 0.4, 0.4], atol=1e-6)
    d = sd(abi.RM_OCTAHEDRON, [[1, 0, 0], [0, 0, 0], [0, -2, 0]])[:, 0]
    assert np.allclose(d, [0.5, -0.5 * 0.57735027, 1.5], atol=1e-6)
    d = sd(abi.RM_CONE, [[0, 1.0, 0], [0, -1.0, 0], [2, -0.5, 0]])[:, 0]
    assert np.allclose(d, [0.5, 0.5, 1.5], atol=1e-6)  # above the apex, below the base disc, beside the base rim


def test_deathstar_is_sphere_minus_sphere():
    # far from the carved side it is the r = .5 sphere
    d = sd(abi.RM_DEATHSTAR, [[-2, 0, 0], [0, 2, 0]])[:, 0]
    assert np.allclose(d, [1.5, 1.5], atol=1e-6)
    # on the +x axis inside the carving sphere (centre x = .5, r = .35) the distance is to that sphere's surface
    d = sd(abi.RM_DEATHSTAR, [[0.5, 0, 0]])[:, 0]
    assert np.allclose(d, [0.35], atol=1e-6)


def test_object_transform_and_scale_factor():
    M = h.translate(1, 2, 3) @ h.scale(2, 2, 2)
    d = sd(abi.RM_SPHERE, [[1, 2, 3], [4, 2, 3], [1, 2, 6]], model=M, scale_factor=2.0)[:, 0]
    assert np.allclose(d, [-1.0, 2.0, 2.0], atol=1e-6)  # radius-1 sphere at (1,2,3): frag:1417-1419


def test_mandelbulb_power8_against_float64_iteration():
    """Distance estimator of frag:775-803 re-evaluated in float64 with libm trig."""
    rng = np.random.default_rng(5)
    pts = rng.normal(0, 0.9, (3000, 3)).astype(np.float32)
    out = sd(abi.RM_MANDELBULB, pts)
    P = pts.astype(np.float64)
    ref = np.empty(len(P))
    trapy = np.empty(len(P))
    stable = np.ones(len(P), bool)
    for n, pos in enumerate(P):
        w = pos.copy()
        m = w @ w
        trap = np.array([abs(w[0]), abs(w[1]), abs(w[2]), m])
        dz = 1.0
        for i in range(20):
            dz = 8.0 * m ** 3.5 * dz + 1.0
            r = np.sqrt(w @ w)
            b = 8.0 * np.arccos(w[1] / r)
            a = 8.0 * np.arctan2(w[0], w[2])
            w = pos + r ** 8 * np.array([np.sin(b) * np.sin(a), np.cos(b), np.sin(b) * np.cos(a)])
            trap = np.minimum(trap, np.array([abs(w[0]), abs(w[1]), abs(w[2]), m]))
            m = w @ w
            if abs(m - 2.0) < 1e-3:
                stable[n] = False  # bailout decision within rounding distance
            if m > 2.0:
                break
        ref[n] = 0.25 * np.log(m) * np.sqrt(m) / dz
        trapy[n] = trap[1]
    # the orbit is chaotic: errors grow with dz, but the DE divides by dz, so absolute agreement is tight
    err = np.abs(out[:, 0] - ref)[stable]
    assert np.percentile(err, 99) < 2e-5 and np.median(err) < 2e-7, (np.percentile(err, 99), np.median(err))
    assert np.percentile(np.abs(out[:, 2] - trapy)[stable], 95) < 1e-4
    assert (out[:, 1] == 0).all()


def test_menger_levels_and_trap():
    s = abi.default_settings()
    # centre of the unit sponge is inside the first cross hole; a corner region is solid
    out = sd(abi.RM_MENGERSPONGE, [[0, 0, 0], [0.99, 0.99, 0.99], [3, 0, 0]], settings=s)
    assert out[0, 0] > 0 and out[1, 0] < 0 and np.isclose(out[2, 0], 2.0, atol=1e-6)
    assert np.isclose(out[0, 3], 0.25)  # trap.z = (1+m)/4 of the first level that carved (frag:1067)
    deep = sd(abi.RM_MENGERSPONGE, [[0.99, 0.99, 0.99]], settings=abi.default_settings(mengerLevels=5))
    assert deep[0, 0] >= out[1, 0]  # more levels only remove material


def test_sierpinski_and_mandelbrot_prims_are_finite_and_signed():
    rng = np.random.default_rng(6)
    pts = rng.uniform(-2, 2, (500, 3)).astype(np.float32)
    for t in (abi.RM_SIERPINSKI, abi.RM_MANDELBROT):
        d = sd(t, pts)[:, 0]
        assert np.isfinite(d).all()
    assert (sd(abi.RM_SIERPINSKI, [[5, 5, 5]])[:, 0] > 0).all()


def test_scene_union_picks_nearest_and_reports_last_fractal_trap():
    objs = (abi.RmObject * 3)(
        h.make_object(abi.RM_SPHERE, model=h.translate(-2, 0, 0)),
        h.make_object(abi.RM_MANDELBULB, model=h.translate(2, 0, 0)),
        h.make_object(abi.RM_CUBE, model=h.translate(0, 3, 0)))
    pts = np.array([[-2, 0, 0], [2.0, 0.1, 0.2], [0, 3, 0], [-1.2, 0, 0]], dtype=np.float32)
    out = np.empty((4, 4), dtype=np.float32)
    g, s = h.make_globals(), abi.default_settings()
    assert h.oracle().rmo_probe_sdscene(objs, 3, C.byref(g), C.byref(s), h.fptr(pts), h.fptr(out), 4) == 0
    assert list(out[:, 1]) == [0, 1, 2, 0]
    # UB3: the trap belongs to the last fractal evaluated (the bulb), whichever object is nearest
    assert out[2, 1] == 2 and out[2, 2] > 0 and out[1, 2] > 0


def test_march_properties_on_a_sphere():
    """Primary hit depth (frag:1453-1484): |d| < 1e-3 at the reported point, miss → background."""
    W = Hh = 33
    cam = h.make_camera((0, 0, 3), (0, 0, -1), (0, 1, 0), 45.0, W, Hh)
    objs = (abi.RmObject * 1)(h.make_object(abi.RM_SPHERE, ambient=(1, 1, 1)))
    scene = (cam, objs, 1, (abi.RmLight * 1)(), 0, h.make_globals(ka=1.0))
    s = abi.default_settings(features=abi.RM_FEAT_DARK_BACKGROUND)
    img, br = h.oracle_render(scene, s, W, Hh, bright=True)
    centre = img[Hh // 2, W // 2]
    assert np.allclose(centre, [1, 1, 1, 1])  # ambient only: ka·cAmbient = 1
    assert (img[0, 0] == [0, 0, 0, 1]).all()  # DARK_BACKGROUND miss
    assert (br[..., :3] == 0).all() and (br[..., 3] == 1).all()  # luminance 1.0 is not > 1 (frag:1940)
    hit = img[..., 0] > 0
    # silhouette is a disc of angular radius asin(.5/3): pixel radius ≈ 33/2 · tan(9.59°)/tan(22.5°)
    assert abs(hit.sum() - np.pi * (16.5 * np.tan(np.arcsin(0.5 / 3)) / np.tan(np.deg2rad(22.5))) ** 2) < 40


def test_phong_directional_light_on_a_sphere_centre():
    W = Hh = 31
    cam = h.make_camera((0, 0, 3), (0, 0, -1), (0, 1, 0), 30.0, W, Hh)
    objs = (abi.RmObject * 1)(h.make_object(abi.RM_SPHERE, ambient=(.2, .2, .2), diffuse=(.5, .6, .7), specular=(1, 1, 1), shininess=10))
    lights = (abi.RmLight * 1)(h.make_light(abi.RM_LIGHT_DIRECTIONAL, (1, 1, 1), (0, 0, -1)))
    scene = (cam, objs, 1, lights, 1, h.make_globals(ka=0.5, kd=0.8, ks=0.3))
    s = abi.default_settings(features=abi.RM_FEAT_WHITE_BACKGROUND)  # no bump: N = (0,0,1) at the centre
    img = h.oracle_render(scene, s, W, Hh)
    c = img[Hh // 2, W // 2, :3]
    # N·L = 1, R·V = 1: ka·amb + kd·dif + ks·spec (frag:1860-1923)
    assert np.allclose(c, [0.1 + 0.4 + 0.3, 0.1 + 0.48 + 0.3, 0.1 + 0.56 + 0.3], atol=2e-3)


def test_shadow_reflection_refraction_change_the_image_in_the_expected_places():
    W, Hh = 64, 48
    import test_gpu_parity as tg
    scene = tg.reflect_refract_scene(W, Hh)
    base = h.oracle_render(scene, abi.default_settings(), W, Hh)
    refl = h.oracle_render(scene, abi.default_settings(enableReflection=1), W, Hh)
    refr = h.oracle_render(scene, abi.default_settings(enableRefraction=1), W, Hh)
    assert (refl[..., :3] >= base[..., :3] - 1e-6).all()  # reflection only adds light (frag:2520-2521)
    assert (refr[..., :3] >= base[..., :3] - 1e-6).all()
    assert (refl[..., 3] - base[..., 3]).max() == 1.0 and (refr[..., 3] - base[..., 3]).max() == 1.0
    soft = h.oracle_render(scene, abi.default_settings(enableSoftShadow=1), W, Hh)
    assert (soft[..., :3] <= base[..., :3] + 1e-6).all()  # penumbra factor ≤ 1 (frag:1711, 1928; UB1)
    assert np.abs(soft - base).max() > 0.01
    ao = h.oracle_render(scene, abi.default_settings(enableAmbientOcclusion=1), W, Hh)
    assert (ao[..., :3] <= base[..., :3] + 1e-6).all()


def test_textured_cube_face_samples_the_expected_texels():
    """A 2×2 texture on the +z face of a unit cube seen head-on, ambient off, one head-on light: the image
    quadrants take the four texel colours (uvMapCube frag:1299-1333, bilinear GL_REPEAT sampling)."""
    W = Hh = 64
    cam = h.make_camera((0, 0, 3), (0, 0, -1), (0, 1, 0), 20.0, W, Hh)
    o = h.make_object(abi.RM_CUBE, diffuse=(0, 0, 0))
    o.texLoc, o.repeatU, o.repeatV, o.blend = 0, 1.0, 1.0, 1.0
    objs = (abi.RmObject * 1)(o)
    lights = (abi.RmLight * 1)(h.make_light(abi.RM_LIGHT_DIRECTIONAL, (1, 1, 1), (0, 0, -1)))
    scene = (cam, objs, 1, lights, 1, h.make_globals(ka=0, kd=1, ks=0))
    tex = np.zeros((2, 2, 4), dtype=np.uint8)  # rows bottom-up
    tex[0, 0], tex[0, 1], tex[1, 0], tex[1, 1] = (255, 0, 0, 255), (0, 255, 0, 255), (0, 0, 255, 255), (255, 255, 0, 255)
    s = abi.default_settings(features=abi.RM_FEAT_DARK_BACKGROUND)
    img = h.oracle_render(scene, s, W, Hh, textures=[tex])
    # independent float64 restatement: the ray through pixel (x, y) meets the z = +0.5 face at
    # p = ndc · tan(10°) · 2.5 (the eye is 2.5 away), uvMapCube gives (p.x + .5, p.y + .5), then GL bilinear/REPEAT
    t64 = tex[..., :3].astype(np.float64) / 255.0
    k = np.tan(np.deg2rad(10.0)) * 2.5
    worst = 0.0
    for (x, y) in [(5, 7), (20, 20), (44, 20), (20, 44), (44, 44), (32, 32), (60, 3), (1, 62)]:
        u = ((x + 0.5) / W * 2 - 1) * k + 0.5
        v = ((y + 0.5) / Hh * 2 - 1) * k + 0.5
        fu, fv = u * 2 - 0.5, v * 2 - 0.5
        i0, j0 = int(np.floor(fu)), int(np.floor(fv))
        a_, b_ = fu - i0, fv - j0
        tx = lambda i, j: t64[j % 2, i % 2]
        exp = (tx(i0, j0) * (1 - a_) + tx(i0 + 1, j0) * a_) * (1 - b_) + (tx(i0, j0 + 1) * (1 - a_) + tx(i0 + 1, j0 + 1) * a_) * b_
        worst = max(worst, np.abs(img[y, x, :3] - exp).max())
    assert worst < 2e-3, worst  # hit point is within 1e-3 of the face (SURFACE_DIST) → uv within 1e-3
    assert np.allclose(img[Hh // 2, W // 2, :3], [0.5, 0.5, 0.25], atol=0.02)  # centre: average of the four texels


def test_post_pass_properties():
    """Known answers of the post passes (blur.frag, hdr.frag, fxaa.frag)."""
    rng = np.random.default_rng(9)
    Hh, W = 40, 56
    flat = np.full((Hh, W, 4), 0.25, dtype=np.float32)
    flat[..., 3] = 1
    # nothing enabled: identity
    assert (h.oracle_post(flat, None, abi.RmPostSettings()) == flat).all()
    # gamma: 0.25^(1/2.2)
    g = h.oracle_post(flat, None, abi.RmPostSettings(enableGammaCorrection=1))
    assert np.allclose(g[..., :3], 0.25 ** (1 / 2.2), atol=2e-4) and (g[..., 3] == 1).all()
    # hdr: 1 - exp(-c·exposure)
    e = h.oracle_post(flat, None, abi.RmPostSettings(enableHDR=1, exposure=2.0))
    assert np.allclose(e[..., :3], 1 - np.exp(-0.5), atol=2e-4)
    # bloom of a constant bright plane stays constant (weights sum to 1 within fp16 rounding) and is added before tone mapping
    bright = np.full((Hh, W, 4), 2.0, dtype=np.float32)
    b = h.oracle_post(flat, bright, abi.RmPostSettings(enableBloom=1, exposure=1.0))
    assert np.allclose(b[..., :3], 1 - np.exp(-(0.25 + 2.0)), atol=3e-3)
    # a single bright texel spreads along both axes (9 passes: 5 horizontal + 4 vertical) and sums to ≈ its energy
    spot = np.zeros((Hh, W, 4), dtype=np.float32)
    spot[20, 28, :3] = 8.0
    z = np.zeros((Hh, W, 4), dtype=np.float32)
    zb = h.oracle_post(z, spot, abi.RmPostSettings(enableBloom=1, enableHDR=1, exposure=1.0))
    energy = -np.log(1 - zb[..., 0].astype(np.float64))
    assert abs(energy.sum() - 8.0) < 0.2 and energy[20, 28] == energy.max() and energy[20, 40] > 0 and energy[33, 28] > 0
    # FXAA leaves flat regions untouched and softens a hard vertical edge
    edge = np.zeros((Hh, W, 4), dtype=np.float32)
    edge[:, W // 2:, :3] = 1.0
    edge[..., 3] = 1
    f = h.oracle_post(edge, None, abi.RmPostSettings(enableFXAA=1))
    # (the source wraps with GL_REPEAT — the reference never sets a wrap mode on it — so columns 0 / W-1 also form an edge)
    assert (f[:, 3:W // 2 - 3, :3] == 0).all() and (f[:, W // 2 + 3:W - 3, :3] == 1).all()
    mid = f[Hh // 2, W // 2 - 1:W // 2 + 1, 0]
    assert 0 < mid[0] < 1 or 0 < mid[1] < 1
    # the 8-bit FXAA source quantises its input
    noisy = rng.uniform(0, 1, (Hh, W, 4)).astype(np.float32)
    fx = h.oracle_post(noisy * 0 + 0.5004, None, abi.RmPostSettings(enableFXAA=1))
    assert np.allclose(fx[..., :3], round(0.5004 * 255) / 255, atol=1e-6)


def test_row_range_equals_full_frame_rows_and_is_thread_count_independent():
    W, Hh = 40, 30
    scene = h.scene_mandelbulb(W, Hh)
    s = abi.default_settings(fractalIters=12)
    full = h.oracle_render(scene, s, W, Hh, threads=8)
    part = h.oracle_render(scene, s, W, Hh, 7, 19, threads=1)
    assert (full[7:19].view(np.uint32) == part.view(np.uint32)).all()


def test_invalid_inputs_are_rejected():
    W, Hh = 8, 8
    cam, objs, no, lights, nl, g = h.scene_mandelbulb(W, Hh)
    out = np.zeros((Hh, W, 4), dtype=np.float32)
    s = abi.default_settings(features=abi.RM_FEAT_SEA)
    st = h.oracle().rmo_render(C.byref(cam), objs, no, lights, nl, C.byref(g), C.byref(s), W, Hh, 0, Hh, h.fptr(out), None, None, 1)
    assert st == abi.RM_ERR_UNSUPPORTED
    s = abi.default_settings()
    st = h.oracle().rmo_render(C.byref(cam), objs, 31, lights, nl, C.byref(g), C.byref(s), W, Hh, 0, Hh, h.fptr(out), None, None, 1)
    assert st == abi.RM_ERR_CAPACITY
    st = h.oracle().rmo_render(C.byref(cam), objs, no, lights, nl, C.byref(g), C.byref(s), W, Hh, 4, 12, h.fptr(out), None, None, 1)
    assert st == abi.RM_ERR_INVALID_ARGUMENT


def test_sea_height_matches_an_independent_float64_model():
    """seaMap (frag:2195-2217) restated in vectorised float64 numpy, with the hash argument rounded to binary32 as GLSL
    does: the oracle must agree point by point up to its binary32 arithmetic (the residual comes from hash values that
    sit next to a fract() wrap)."""
    import ctypes as C
    f32 = np.float32

    def hash2(x, y):
        a = (x.astype(f32) * f32(12.9898) + y.astype(f32) * f32(78.233)).astype(np.float64)
        v = np.sin(a) * 43758.5453
        return v - np.floor(v)

    def noise_w(px, py):
        ix, iy = np.floor(px), np.floor(py)
        fx, fy = px - ix, py - iy
        ux, uy = fx * fx * (3 - 2 * fx), fy * fy * (3 - 2 * fy)
        a, b, c, d = hash2(ix, iy), hash2(ix + 1, iy), hash2(ix, iy + 1), hash2(ix + 1, iy + 1)
        return 2 * ((a * (1 - ux) + b * ux) * (1 - uy) + (c * (1 - ux) + d * ux) * uy) - 1

    def sea_octave(ux, uy, choppy):
        n = noise_w(ux, uy)
        ux, uy = ux + n, uy + n
        wx, wy = 1 - np.abs(np.sin(ux)), 1 - np.abs(np.sin(uy))
        wx, wy = wx * (1 - wx) + np.abs(np.cos(ux)) * wx, wy * (1 - wy) + np.abs(np.cos(uy)) * wy
        return (1 - (wx * wy) ** 0.65) ** choppy

    def sea_map(p, iters, itime):
        st, freq, amp, ch = 1 + itime * 0.5, 0.16, 0.2, 1.0
        ux, uy, hh = p[:, 0].copy(), p[:, 2].copy(), 0.0
        for _ in range(iters):
            hh = hh + (sea_octave((ux + st) * freq, (uy + st) * freq, ch) + sea_octave((ux - st) * freq, (uy - st) * freq, ch)) * amp
            ux, uy = ux * 1.6 + uy * 1.2, ux * -1.2 + uy * 1.6
            freq, amp, ch = freq * 2, amp * 0.2, ch * 0.8 + 0.2
        return p[:, 1] - hh

    rng = np.random.default_rng(8)
    pts = np.ascontiguousarray(rng.uniform(-40, 40, (3000, 3)).astype(np.float32))
    out = np.zeros((len(pts), 4), np.float32)
    for itime in (0.0, 2.3):
        assert h.oracle().rmo_probe_env2(3, C.c_float(itime), None, h.fptr(pts), h.fptr(out), len(pts)) == 0
        for k, iters in ((0, 3), (1, 5)):
            d = np.abs(sea_map(pts.astype(np.float64), iters, itime) - out[:, k])
            # a binary32 hash has a granularity of ulp(43758) = 0.004, hence the floor on the agreement
            assert np.median(d) < 4e-3 and d.mean() < 1e-2, (itime, iters, np.median(d), d.mean())


def test_bulb_algebraic_power8_is_the_same_function():
    """RM_FEAT_BULB_POWER8_ALGEBRAIC evaluates the Mandelbulb step by complex squarings instead of acos/atan/sin/cos/pow.
    Same function: the distance estimate agrees with the trigonometric formulation to a few 1e-8 (median), and a frame
    differs on < 0.3 % of its pixels — two orders of magnitude closer than the reference shader on SwiftShader is to
    either (tests/test_oracle_vs_glsl.py: 14 % of bulb pixels > 1e-3)."""
    import ctypes as C
    rng = np.random.default_rng(21)
    pts = np.ascontiguousarray(rng.normal(0, 0.8, (20000, 3)).astype(np.float32))
    cam, objs, no, lights, nl, g = h.scene_mandelbulb(8, 8)
    outs = []
    for feat in (0, abi.RM_FEAT_BULB_POWER8_ALGEBRAIC):
        s = abi.default_settings(features=abi.RM_FEAT_REFERENCE_DEFAULT | feat)
        o = np.empty((len(pts), 4), np.float32)
        assert h.oracle().rmo_probe_sdscene(objs, no, C.byref(g), C.byref(s), h.fptr(pts), h.fptr(o), len(pts)) == 0
        outs.append(o)
    d = np.abs(outs[0][:, 0] - outs[1][:, 0])
    assert np.median(d) < 2e-7 and (d < 5e-5).mean() > 0.995, (np.median(d), (d < 5e-5).mean())
    W, H = 160, 90
    sc = h.scene_mandelbulb(W, H)
    fr = [h.oracle_render(sc, abi.default_settings(fractalIters=12, features=abi.RM_FEAT_REFERENCE_DEFAULT | f), W, H)
          for f in (0, abi.RM_FEAT_BULB_POWER8_ALGEBRAIC)]
    dd = np.abs(fr[0] - fr[1]).max(-1)
    assert (dd > 1e-3).mean() < 0.003 and dd.mean() < 1e-4, ((dd > 1e-3).mean(), dd.mean())


# ---------------------------------------------------------------- an INDEPENDENT binary64 evaluation of the Mandelbulb estimator
def _bulb_de_float64(pos, power, iters):
    """frag:775-803 transcribed from the SHADER TEXT into NumPy float64 (tests/arbiter_numpy.py, which carries the whole headline
    pixel the same way): not the oracle's source with wider types."""
    import arbiter_numpy as an
    d, res = an.sd_mandelbulb(pos, power, iters)
    return d, res[:, 0], res[:, 1:]


@pytest.mark.parametrize("fixture,iters", [("probe_sd_bulb_p8_12iters.npz", 12), ("probe_sd_bulb_p8.npz", 20)])
def test_bulb_estimator_against_an_independent_float64_transcription(fixture, iters):
    """The binary32 oracle's sdScene on the probe points of the reference-shader fixtures against the transcription above:
    the arbiter of rm_oracle_f64.c shares the oracle's source, so a mis-transcribed formula would pass there — not here.
    Points whose orbit passes within rounding distance of the bailout radius run a different number of iterations in the
    two evaluations (a legitimate flip, ≈0.5 %); all others agree to a few binary32 ulps of the estimate."""
    import ctypes as C
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "glsl", fixture))
    pts = np.ascontiguousarray(z["pts"], dtype=np.float32)
    objs = (abi.RmObject * 1).from_buffer_copy(z["objs"].tobytes()[:C.sizeof(abi.RmObject)])
    g = abi.RmGlobals.from_buffer_copy(z["globals"].tobytes())
    s = abi.RmSettings.from_buffer_copy(z["settings"].tobytes())
    assert s.fractalIters == iters and g.power == 8.0 and objs[0].type == abi.RM_MANDELBULB and objs[0].scaleFactor == 1.0
    out = np.empty((len(pts), 4), dtype=np.float32)
    assert h.oracle().rmo_probe_sdscene(objs, 1, C.byref(g), C.byref(s), h.fptr(pts), h.fptr(out), len(pts)) == 0
    d64, m64, trap64 = _bulb_de_float64(pts, 8.0, iters)
    err = np.abs(out[:, 0].astype(np.float64) - d64)
    rel = err / np.maximum(np.abs(d64), 1e-3)
    close = (err <= 5e-6) | (rel <= 2e-5)
    assert close.mean() >= 0.99, f"{(~close).sum()} of {len(pts)} points off (median {np.median(err):.2e})"
    assert np.median(err) < 2e-7
    # the orbit-trap components the palette reads (frag:2356-2360) on the agreeing points
    t_err = np.abs(out[close, 2:].astype(np.float64) - trap64[close][:, :2])
    assert np.percentile(t_err, 99) < 1e-4
