"""GPU parity tests proper: the HIP path, called through the C-ABI, against the CPU oracle.

The numeric contract makes every frame bit-reproducible, so the bar is BIT EQUALITY of
the float32 outputs (tolerance 0), far inside the 1e-3 per-channel L∞ the north star states.
"""
import ctypes as C
import os

import numpy as np
import pytest

import helpers as h
from raymarcher_amd import abi

pytestmark = pytest.mark.gpu


def tables_of(scene):
    from raymarcher_amd.render import SceneTables
    cam, objs, no, lights, nl, g = scene
    return SceneTables(cam, objs, no, lights, nl, g)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bit_equal(gpu, ref, what):
    gb, rb = bits(gpu), bits(ref)
    bad = gb != rb
    if bad.any():
        idx = np.argwhere(bad)[:5]
        diff = np.abs(gpu.astype(np.float64) - ref.astype(np.float64))
        raise AssertionError(f"{what}: {bad.sum()} of {bad.size} words differ; max |Δ| = {np.nanmax(diff):.3e}; "
                             f"first at {idx.tolist()}")


# ---------------------------------------------------------------- the numeric contract, function by function
def _math_inputs(fn, n, rng):
    f = np.float32
    if fn in (abi.RM_FN_SIN, abi.RM_FN_COS):
        x = np.concatenate([rng.uniform(-30, 30, n // 2), rng.uniform(-1e4, 1e4, n // 4), rng.normal(0, 1e-3, n // 4)])
        x = np.concatenate([x, [0.0, -0.0, np.pi, 1e7, -1e7, np.inf, -np.inf, np.nan, 4194303.5, 4194304.0]])
        return x.astype(f), None
    if fn in (abi.RM_FN_ACOS, abi.RM_FN_ASIN):
        x = np.concatenate([rng.uniform(-1, 1, n), [0, 1, -1, 0.5, -0.5, 1.0000001, -1.0000001, 2, -2, np.nan, 0.49999997]])
        return x.astype(f), None
    if fn == abi.RM_FN_ATAN2:
        y = np.concatenate([rng.normal(0, 2, n), [0, 0, -0.0, 1, -1, np.inf, np.inf, np.nan, 0.0, 1e-30]])
        x = np.concatenate([rng.normal(0, 2, n), [0, -1, -1, 0, 0, np.inf, -np.inf, 1, -0.0, 1e-30]])
        return y.astype(f), x.astype(f)
    if fn == abi.RM_FN_LOG2:
        x = np.concatenate([np.exp(rng.uniform(-80, 80, n)), rng.uniform(0.5, 2, n),
                            [0, -0.0, 1, -1, 1e-45, 1.17549435e-38, 1.1754942e-38, np.inf, np.nan, 2, 0.70710677, 0.70710683]])
        return x.astype(f), None
    if fn == abi.RM_FN_EXP2:
        x = np.concatenate([rng.uniform(-130, 130, n), rng.uniform(-1, 1, n),
                            [0, -125, -125.00001, -124.99999, 127.5, 127.99999, 128, 1e30, -1e30, np.inf, -np.inf, np.nan, 0.5, -0.5, 1.5, 2.5]])
        return x.astype(f), None
    if fn == abi.RM_FN_POW:
        # a third of the exponents are integers or half-integers (binary-exponentiation branch, |y| <= 128 and beyond),
        # with negative and zero bases among them
        k = n // 3
        yi = rng.integers(-300, 301, k) / 2.0
        xi = np.where(rng.uniform(size=k) < 0.2, -1.0, 1.0) * np.exp(rng.uniform(-1.5, 1.5, k))
        x = np.concatenate([np.exp(rng.uniform(-5, 5, n - k)), xi, [0, 0, 0, 1, 2, -1, np.inf, 0.9, 1e-40, 0, 0, -2, -2, 3, 3, 1.5, np.nan, 2]])
        y = np.concatenate([rng.uniform(-20, 100, n - k), yi, [0, 1, -1, 5, 0.5, 2, 2, 1e5, 2, -0.0, 3.5, 3, -3, 128, 128.5, -127.5, 0, np.nan]])
        return x.astype(f), y.astype(f)
    if fn == abi.RM_FN_Q16:
        x = np.concatenate([rng.normal(0, 1, n) * np.exp(rng.uniform(-25, 12, n)),
                            [0, -0.0, 6.1e-5, 6.0e-5, 5.96e-8, 2.98e-8, 2.99e-8, 1e-9, 65504, 65519.9, 65520, 1e6, np.inf, np.nan,
                             1.00048828125, 1.000244140625, 1.000732421875]])
        return x.astype(f), None
    if fn in (abi.RM_FN_SQRT, abi.RM_FN_SQRT_FAST):
        x = np.concatenate([np.exp(rng.uniform(-87, 88, n)), np.exp(rng.uniform(-68, -64, n // 4)), rng.uniform(0, 4, n),
                            [0, -0.0, 1e-45, 1e-40, 1.17549435e-38, 1.2621774e-29, 1.2621775e-29, 1.262177e-29, np.inf, -1, np.nan, 2, 4, 0.25, 2.9802322e-8]])
        return x.astype(f), None
    if fn == abi.RM_FN_DIV:
        x = np.concatenate([rng.normal(0, 100, n) * np.exp(rng.uniform(-40, 40, n)), [0, 1, 1, -1, 0, np.inf, 1e-40, 1e38, 1]])
        y = np.concatenate([rng.normal(0, 100, n) * np.exp(rng.uniform(-40, 40, n)), [1, 0, 3, 7, 0, np.inf, 1e3, 1e-5, 289]])
        return x.astype(f), y.astype(f)
    if fn in (abi.RM_FN_DIVR, abi.RM_FN_RCP):  # incl. the edges of the reciprocal's fast range, denormals, zeros, infinities
        edge = [0, -0.0, 1, -1, 3, 7, 289, 2000, 1.17549435e-38, 1.1754942e-38, 1e-40, 1e-45, 8.5070592e37, 8.507059e37, 1.7e38,
                3.4e38, np.inf, -np.inf, np.nan, 0.5, 2, 1e-20, 1e20]
        y = np.concatenate([rng.normal(0, 100, n) * np.exp(rng.uniform(-80, 80, n)), edge, edge]).astype(f)
        x = np.concatenate([rng.normal(0, 100, n) * np.exp(rng.uniform(-40, 40, n)), edge, edge[::-1]]).astype(f)
        return (y, None) if fn == abi.RM_FN_RCP else (x, y)
    raise ValueError(fn)


@pytest.mark.parametrize("fn", [abi.RM_FN_SIN, abi.RM_FN_COS, abi.RM_FN_ACOS, abi.RM_FN_ASIN, abi.RM_FN_ATAN2, abi.RM_FN_LOG2,
                                abi.RM_FN_EXP2, abi.RM_FN_POW, abi.RM_FN_SQRT, abi.RM_FN_DIV, abi.RM_FN_Q16, abi.RM_FN_SQRT_FAST,
                                abi.RM_FN_DIVR, abi.RM_FN_RCP])
def test_math_contract_bit_exact(renderer, fn):
    import torch
    rng = np.random.default_rng(1000 + fn)
    x, y = _math_inputs(fn, 200000, rng)
    ref = np.empty_like(x)
    st = h.oracle().rmo_probe_math(fn, h.fptr(x), h.fptr(y) if y is not None else None, None, h.fptr(ref), x.size)
    assert st == 0
    tx = torch.from_numpy(x).cuda()
    ty = torch.from_numpy(y).cuda() if y is not None else None
    got = renderer.probe_math(fn, tx, ty).cpu().numpy()
    # NaN payloads are compared as "both NaN"
    both_nan = np.isnan(got) & np.isnan(ref)
    gb, rb = bits(got), bits(ref)
    bad = (gb != rb) & ~both_nan
    assert not bad.any(), f"fn {fn}: {bad.sum()} mismatches, e.g. x={x[bad][:4]}, gpu={got[bad][:4]}, cpu={ref[bad][:4]}"


def test_cheap_reciprocal_and_square_root_are_the_ieee_results_for_every_input(renderer):
    """rcp_() — v_rcp_f32 + one Newton step wherever the whole wave is inside 2^-126 <= |y| < 2^126 — against 1.0f / y, and
    sqrt_fast_() / sqrt_noscale_() — v_sqrt_f32 + residual selection without the 2^32 pre-scaling — against sqrtf, for all
    2^32 inputs on the device (rm_debug_check_math): the guarded functions, and the bare fast forms over their whole ranges;
    and fract_() = v_fract_f32 against the contract's "x − floor(x), kept below 1"."""
    import ctypes as C
    from raymarcher_amd import lib
    out = (C.c_ulonglong * 5)()
    assert lib().rm_debug_check_math(out) == 0
    assert tuple(out) == (0, 0, 0, 0, 0)


def test_min_max_fract_bit_exact_on_every_pair_of_special_values(renderer):
    """min / max with the hardware's rule (signalling NaN → quieted, quiet NaN ignored, −0 < +0) and fract kept below 1: every
    ordered pair of ~1500 values (zeros, denormals, infinities, both kinds of NaN with payloads, neighbours) bit for bit —
    NaN payloads included, which the rule defines."""
    import torch
    rng = np.random.default_rng(9)
    special = np.array([0x00000000, 0x80000000, 0x00000001, 0x80000001, 0x007fffff, 0x807fffff, 0x00800000, 0x80800000,
                        0x3f800000, 0xbf800000, 0x3f7fffff, 0x3f800001, 0x7f7fffff, 0xff7fffff, 0x7f800000, 0xff800000,
                        0x7fc00000, 0xffc00000, 0x7fc00001, 0x7fffffff, 0x7f800001, 0xff800001, 0x7fa00000, 0x7fbfffff,
                        0xffa12345, 0x7fd12345], dtype=np.uint32)
    vals = np.concatenate([special, rng.integers(0, 2**32, 900, dtype=np.uint64).astype(np.uint32),
                           (rng.normal(0, 3, 600)).astype(np.float32).view(np.uint32)]).view(np.float32)
    a, b = np.meshgrid(vals, vals, indexing="ij")
    a, b = np.ascontiguousarray(a.ravel()), np.ascontiguousarray(b.ravel())
    for fn in (abi.RM_FN_MIN, abi.RM_FN_MAX):
        ref = np.empty_like(a)
        assert h.oracle().rmo_probe_math(fn, h.fptr(a), h.fptr(b), None, h.fptr(ref), a.size) == 0
        got = renderer.probe_math(fn, torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()).cpu().numpy()
        bad = bits(got) != bits(ref)
        assert not bad.any(), f"fn {fn}: {bad.sum()} mismatches, e.g. {bits(a[bad][:4])}, {bits(b[bad][:4])}: gpu {bits(got[bad][:4])} cpu {bits(ref[bad][:4])}"
    # the Menger level's median (rm_device.hip.h, mengerImpl): v_med3_f32(|x|,|y|,|z|) against the shader's min / max chain
    # (frag:1063-1064) through the oracle's contract min / max, on every triple of a set with zeros, denormals, infinities and quiet
    # NaNs (the operands are |fma(…)|: never signalling).  A NaN result only has to be a NaN: the level's `c > d` drops it.
    sp = np.array([0x00000000, 0x00000001, 0x007fffff, 0x00800000, 0x3f7fffff, 0x3f800000, 0x3f800001, 0x40000000, 0x7f7fffff,
                   0x7f800000, 0x7fc00000, 0x7fc12345, 0xffc00000, 0x80000000, 0xbf800000, 0xff800000], dtype=np.uint32)
    tv = np.concatenate([sp, rng.random(24, dtype=np.float32).view(np.uint32)]).view(np.float32)
    ta, tb, tc = (np.ascontiguousarray(g.ravel()) for g in np.meshgrid(tv, tv, tv, indexing="ij"))

    def omm(fn, p, q):
        out = np.empty_like(p)
        assert h.oracle().rmo_probe_math(fn, h.fptr(p), h.fptr(q), None, h.fptr(out), p.size) == 0
        return out
    aa, ab, ac = np.abs(ta), np.abs(tb), np.abs(tc)
    chain = omm(abi.RM_FN_MIN, omm(abi.RM_FN_MAX, aa, ab), omm(abi.RM_FN_MIN, omm(abi.RM_FN_MAX, ab, ac), omm(abi.RM_FN_MAX, ac, aa)))
    med = renderer.probe_math(abi.RM_FN_MEDIAN_ABS, torch.from_numpy(ta).cuda(), torch.from_numpy(tb).cuda(), torch.from_numpy(tc).cuda()).cpu().numpy()
    bad = (bits(med) != bits(chain)) & ~(np.isnan(med) & np.isnan(chain))
    assert not bad.any(), f"median: {bad.sum()} mismatches, e.g. {bits(ta[bad][:3])} {bits(tb[bad][:3])} {bits(tc[bad][:3])}: {bits(med[bad][:3])} vs {bits(chain[bad][:3])}"
    x = np.concatenate([vals, (rng.normal(0, 1e-7, 100000)).astype(np.float32), rng.normal(0, 100, 100000).astype(np.float32)])
    ref = np.empty_like(x)
    assert h.oracle().rmo_probe_math(abi.RM_FN_FRACT, h.fptr(x), None, None, h.fptr(ref), x.size) == 0
    got = renderer.probe_math(abi.RM_FN_FRACT, torch.from_numpy(x).cuda()).cpu().numpy()
    bad = (bits(got) != bits(ref)) & ~(np.isnan(got) & np.isnan(ref))
    assert not bad.any()


def test_library_loaded_before_torch_still_finds_the_device():
    """A host that loads the library first and imports torch afterwards (build() then smoke() in one process): both must end
    up on ONE HIP runtime (raymarcher_amd._lib shares the copy bundled with the PyTorch wheel)."""
    import subprocess
    import sys
    code = ("from raymarcher_amd import lib, abi; lib(); import torch; from raymarcher_amd import Renderer, scenes; "
            "r = Renderer(0); o = r.render(scenes.mandelbulb(32, 16), abi.default_settings(), 32, 16); "
            "print('frame', tuple(o.shape), float(o.sum()) == float(o.sum()))")
    p = subprocess.run([sys.executable, "-c", code], cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "frame (16, 32, 4) True" in p.stdout, p.stderr[-2000:]


def test_smoothstep_bit_exact(renderer):
    """smoothstep with the contract's x·RN(1/(e1 − e0)): random edges (also equal, reversed, tiny and non-finite ones)."""
    import torch
    rng = np.random.default_rng(77)
    n = 200000
    e0 = rng.normal(0, 10, n).astype(np.float32)
    e1 = (e0 + rng.normal(0, 5, n) * np.exp(rng.uniform(-30, 3, n))).astype(np.float32)
    x = (e0 + (e1 - e0) * rng.uniform(-0.5, 1.5, n)).astype(np.float32)
    e1[:50] = e0[:50]
    e1[50:60] = np.inf
    e0[60:70] = np.nan
    ref = np.empty_like(x)
    assert h.oracle().rmo_probe_math(abi.RM_FN_SMOOTHSTEP, h.fptr(e0), h.fptr(e1), h.fptr(x), h.fptr(ref), n) == 0
    got = renderer.probe_math(abi.RM_FN_SMOOTHSTEP, torch.from_numpy(e0).cuda(), torch.from_numpy(e1).cuda(), torch.from_numpy(x).cuda()).cpu().numpy()
    bad = (bits(got) != bits(ref)) & ~(np.isnan(got) & np.isnan(ref))
    assert not bad.any(), f"{bad.sum()} mismatches, e.g. {e0[bad][:3]}, {e1[bad][:3]}, {x[bad][:3]}: gpu {got[bad][:3]} cpu {ref[bad][:3]}"


def test_pnoise_bit_exact(renderer):
    import torch
    rng = np.random.default_rng(7)
    n = 100000
    p = rng.uniform(-40, 40, (3, n)).astype(np.float32)
    p[:, :8] = np.array([[0, 0, 0], [1, 2, 3], [-1, -2, -3], [255.5, 0.25, -0.75], [256, 256, 256], [-0.0, 0.0, 5.5],
                         [1e-8, -1e-8, 0.5], [12.999999, 3.0000002, -7.0]], dtype=np.float32).T
    ref = np.empty(n, dtype=np.float32)
    x, y, z = (np.ascontiguousarray(p[i]) for i in range(3))
    assert h.oracle().rmo_probe_math(abi.RM_FN_PNOISE3, h.fptr(x), h.fptr(y), h.fptr(z), h.fptr(ref), n) == 0
    got = renderer.probe_math(abi.RM_FN_PNOISE3, torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(),
                              torch.from_numpy(z).cuda()).cpu().numpy()
    assert_bit_equal(got, ref, "pnoise")
    assert np.abs(ref).max() <= 1.5 and np.abs(ref).max() > 0.3


# ---------------------------------------------------------------- sdScene on random points, every primitive type
def all_primitives_scene(W=64, H=64):
    cam = h.make_camera((0, 0, 6), (0, 0, -1), (0, 1, 0), 45.0, W, H)
    types = [abi.RM_CUBE, abi.RM_CONE, abi.RM_CYLINDER, abi.RM_SPHERE, abi.RM_OCTAHEDRON, abi.RM_TORUS, abi.RM_CAPSULE,
             abi.RM_DEATHSTAR, abi.RM_RECTANGLE, abi.RM_SIERPINSKI, abi.RM_MENGERSPONGE, abi.RM_MANDELBULB]
    objs = (abi.RmObject * len(types))()
    for i, t in enumerate(types):
        gx, gy = (i % 4) - 1.5, (i // 4) - 1.0
        M = h.translate(1.6 * gx, 1.6 * gy, 0.0) @ h.scale(0.9, 0.8 + 0.05 * i, 0.9)
        objs[i] = h.make_object(t, model=M, scale_factor=min(0.9, 0.8 + 0.05 * i), ambient=(.2, .2, .2),
                                diffuse=(0.3 + 0.05 * i, 0.8, 1.0 - 0.05 * i), specular=(1, 1, 1), shininess=15.0 + i)
    lights = (abi.RmLight * 3)(
        h.make_light(abi.RM_LIGHT_DIRECTIONAL, (1, 1, 1), (-0.3, -1, -0.6)),
        h.make_light(abi.RM_LIGHT_POINT, (1, 0.8, 0.6), pos=(3, 3, 4), func=(0.5, 0.1, 0.01)),
        h.make_light(abi.RM_LIGHT_SPOT, (0.7, 0.8, 1), direction=(0, -1, -1), pos=(0, 5, 5), func=(0.8, 0.02, 0.0),
                     angle=np.deg2rad(35.0), penumbra=np.deg2rad(12.0)))
    return cam, objs, len(types), lights, 3, h.make_globals()


def test_sdscene_bit_exact_all_primitives(renderer):
    import torch
    scene = all_primitives_scene()
    s = abi.default_settings()
    rng = np.random.default_rng(3)
    pts = rng.uniform(-3.5, 3.5, (50000, 3)).astype(np.float32)
    pts[:, 2] *= 0.4
    ref = np.empty((pts.shape[0], 4), dtype=np.float32)
    cam, objs, no, lights, nl, g = scene
    assert h.oracle().rmo_probe_sdscene(objs, no, C.byref(g), C.byref(s), h.fptr(pts), h.fptr(ref), pts.shape[0]) == 0
    got = renderer.probe_sdscene(tables_of(scene), s, torch.from_numpy(pts).cuda()).cpu().numpy()
    assert_bit_equal(got, ref, "sdScene")
    assert len(np.unique(ref[:, 1])) >= 10  # the points really exercise most object types


def test_sdscene_algebraic_power8(renderer):
    """RM_FEAT_BULB_POWER8_ALGEBRAIC: bit-exact CPU↔GPU like everything else, within 5e-5 of the trigonometric
    formulation on ≥ 99.5 % of points (the rest sit on a bailout boundary), and ignored for any other power."""
    import torch
    rng = np.random.default_rng(12)
    pts = np.ascontiguousarray(rng.normal(0, 0.8, (60000, 3)).astype(np.float32))
    pts[:64, 0] = 0.0
    pts[:64, 2] = 0.0  # the ρ = 0 axis (atan(0,0) branch)
    alg = abi.default_settings(features=abi.RM_FEAT_REFERENCE_DEFAULT | abi.RM_FEAT_BULB_POWER8_ALGEBRAIC)
    trig = abi.default_settings()

    def both(scene, s):
        cam, objs, no, lights, nl, g = scene
        ref = np.empty((len(pts), 4), dtype=np.float32)
        assert h.oracle().rmo_probe_sdscene(objs, no, C.byref(g), C.byref(s), h.fptr(pts), h.fptr(ref), len(pts)) == 0
        got = renderer.probe_sdscene(tables_of(scene), s, torch.from_numpy(pts).cuda()).cpu().numpy()
        assert_bit_equal(got, ref, "sdScene")
        return ref

    bulb = h.scene_mandelbulb(8, 8)
    a, t = both(bulb, alg), both(bulb, trig)
    assert np.isfinite(a[:, 0]).all()
    d = np.abs(a[:, 0] - t[:, 0])
    assert np.median(d) < 5e-7 and (d < 5e-5).mean() > 0.995
    assert (a != t).any()
    p6 = bulb[:5] + (h.make_globals(power=6.0),)
    assert_bit_equal(both(p6, alg), both(p6, trig), "the bit is ignored unless power == 8")


def env_scene(W, H, pos=(0, 500, 5), look=(0.3, 0.12, -1)):
    """Terrain + volumetric cloud + sky (the shader's TERRAIN / CLOUD / SKY_BACKGROUND defines), with a reflective
    torus floating in front of the camera so secondary rays also see the layers (frag:2506-2518)."""
    cam = h.make_camera(pos, look, (0, 1, 0), 70.0, W, H, far=2000.0)
    objs = (abi.RmObject * 1)(h.make_object(abi.RM_TORUS, model=h.translate(8, pos[1] + 3, -30) @ h.scale(12, 12, 12),
                                            scale_factor=12, ambient=(.3, .3, .3), specular=(1, 1, 1), shininess=50,
                                            reflective=(.6, .6, .6), transparent=(.5, .5, .5), ior=1.3))
    lights = (abi.RmLight * 1)(h.make_light(abi.RM_LIGHT_DIRECTIONAL, (3, 2.6, 2.0), (-0.577, -0.577, 0.577)))
    return cam, objs, 1, lights, 1, h.make_globals()


ENV_ALL = abi.RM_FEAT_SKY_BACKGROUND | abi.RM_FEAT_TERRAIN | abi.RM_FEAT_CLOUD | abi.RM_FEAT_PERLIN_BUMP

# ---------------------------------------------------------------- whole frames
FRAME_CASES = {
    "env_terrain_cloud_sky_reflect": (lambda W, H: env_scene(W, H), {"features": ENV_ALL, "enableReflection": 1}, 96, 54),
    "env_terrain_cloud_refract_time": (lambda W, H: env_scene(W, H, (0, 560, 0), (0.2, 0.3, -1)),
                                       {"features": ENV_ALL, "enableRefraction": 1, "enableReflection": 1}, 80, 45),
    "env_terrain_sky_only": (lambda W, H: env_scene(W, H), {"features": abi.RM_FEAT_SKY_BACKGROUND | abi.RM_FEAT_TERRAIN}, 64, 36),
    "env_cloud_only_dark": (lambda W, H: env_scene(W, H), {"features": abi.RM_FEAT_CLOUD | abi.RM_FEAT_DARK_BACKGROUND}, 64, 36),
    "bulb_reference_consts": (lambda W, H: h.scene_mandelbulb(W, H), {}, 96, 54),
    "bulb_bench_consts_12iters": (lambda W, H: h.scene_mandelbulb(W, H), {"fractalIters": 12}, 96, 54),
    "bulb_softshadow_ao": (lambda W, H: h.scene_mandelbulb(W, H), {"enableSoftShadow": 1, "enableAmbientOcclusion": 1}, 64, 36),
    "bulb_algebraic_power8": (lambda W, H: h.scene_mandelbulb(W, H),
                              {"fractalIters": 12, "features": abi.RM_FEAT_REFERENCE_DEFAULT | abi.RM_FEAT_BULB_POWER8_ALGEBRAIC}, 96, 54),
    "bulb_algebraic_julia_ao": (lambda W, H: h.scene_mandelbulb(W, H)[:5] + (h.make_globals(julia=(0.35, -0.2)),),
                                     {"enableAmbientOcclusion": 1,
                                      "features": abi.RM_FEAT_REFERENCE_DEFAULT | abi.RM_FEAT_BULB_POWER8_ALGEBRAIC}, 64, 36),
    "primitives_phong": (lambda W, H: all_primitives_scene(W, H), {"maxSteps": 64}, 96, 64),
    "primitives_softshadow_ao_nobump": (lambda W, H: all_primitives_scene(W, H),
                                        {"enableSoftShadow": 1, "enableAmbientOcclusion": 1,
                                         "features": abi.RM_FEAT_DARK_BACKGROUND}, 80, 48),
}


@pytest.mark.parametrize("name", list(FRAME_CASES))
def test_frame_bit_exact(renderer, name):
    build, over, W, H = FRAME_CASES[name]
    scene = build(W, H)
    s = abi.default_settings(**over)
    ref, ref_b = h.oracle_render(scene, s, W, H, bright=True)
    out, br = renderer.render(tables_of(scene), s, W, H, bright=True)
    assert_bit_equal(out.cpu().numpy(), ref, f"{name} fragColor")
    assert_bit_equal(br.cpu().numpy(), ref_b, f"{name} BrightColor")
    assert np.isfinite(ref).all()
    if name == "env_terrain_cloud_refract_time":
        scene[5].iTime = 12.5  # clouds drift with iTime (frag:1951)
        ref2 = h.oracle_render(scene, s, W, H)
        assert np.abs(ref2 - ref).max() > 1e-3
        assert_bit_equal(renderer.render(tables_of(scene), s, W, H).cpu().numpy(), ref2, f"{name} at iTime 12.5")
    hit = (ref[..., :3] != ref[0, 0, :3]).any(axis=-1).mean()
    assert 0.05 < hit <= 1.0, "frame should not be constant"


def reflect_refract_scene(W, H):
    cam = h.make_camera((0, 1.2, 5), (0, -0.2, -1), (0, 1, 0), 40.0, W, H)
    objs = (abi.RmObject * 4)(
        h.make_object(abi.RM_SPHERE, model=h.translate(-1.1, 0, 0) @ h.scale(1.6, 1.6, 1.6), scale_factor=1.6,
                      ambient=(.1, .1, .1), diffuse=(.8, .2, .2), specular=(1, 1, 1), shininess=30, reflective=(.8, .8, .8)),
        h.make_object(abi.RM_SPHERE, model=h.translate(1.1, 0, 0.3) @ h.scale(1.5, 1.5, 1.5), scale_factor=1.5,
                      ambient=(.1, .1, .1), diffuse=(.2, .3, .8), specular=(1, 1, 1), shininess=50,
                      transparent=(.9, .9, .9), ior=1.4),
        h.make_object(abi.RM_CUBE, model=h.translate(0, -1.3, 0) @ h.scale(8, 1, 8), scale_factor=1.0,
                      ambient=(.2, .2, .2), diffuse=(.6, .6, .5), specular=(.3, .3, .3), shininess=5, reflective=(.3, .3, .3)),
        h.make_object(abi.RM_TORUS, model=h.translate(0.2, 0.4, -2.0) @ h.scale(2, 2, 2), scale_factor=2.0,
                      ambient=(.1, .2, .1), diffuse=(.3, .9, .3), specular=(1, 1, 1), shininess=10))
    lights = (abi.RmLight * 2)(
        h.make_light(abi.RM_LIGHT_DIRECTIONAL, (1, 1, 1), (-0.5, -1, -0.4)),
        h.make_light(abi.RM_LIGHT_POINT, (.8, .8, 1), pos=(-3, 4, 3), func=(0.6, 0.05, 0.0)))
    return cam, objs, 4, lights, 2, h.make_globals(kt=0.8)


def test_kernels_without_secondary_rays_pick_up_exactly_where_they_may(renderer):
    """The launcher compiles main's reflection loop and refraction out (render_kernel<…, SEC = false>) when they cannot fire for
    any pixel: reflection off, or on with zero bounces, or no reflective material; refraction off or no transparent material.
    Every combination around that decision — for the table walk, the bulb class, the sampler kernel and the layer kernel — is the
    oracle's frame bit for bit, and frames that differ only in a switch that cannot matter are identical to each other."""
    W, H = 80, 48
    cam, objs, no, lights, nl, g = reflect_refract_scene(W, H)

    def variant(reflective, transparent):
        o = (abi.RmObject * no)(*[abi.RmObject.from_buffer_copy(bytes(objs[i])) for i in range(no)])
        for i in range(no):
            for k in range(3):
                if not reflective:
                    o[i].cReflective[k] = 0.0
                if not transparent:
                    o[i].cTransparent[k] = 0.0
        return (cam, o, no, lights, nl, g)

    frames = {}
    for mat in ((1, 1), (1, 0), (0, 1), (0, 0)):
        scene = variant(*mat)
        for refl, nb, refr in ((0, 1, 0), (1, 0, 0), (1, 1, 0), (0, 1, 1), (1, 2, 1), (1, 0, 1)):
            s = abi.default_settings(enableReflection=refl, numReflection=nb, enableRefraction=refr)
            ref, ref_b = h.oracle_render(scene, s, W, H, bright=True)
            out, br = renderer.render(tables_of(scene), s, W, H, bright=True)
            assert_bit_equal(out.cpu().numpy(), ref, f"materials {mat}, reflection {refl} x{nb}, refraction {refr}")
            assert_bit_equal(br.cpu().numpy(), ref_b, f"materials {mat}, reflection {refl} x{nb}, refraction {refr} bright")
            frames[(mat, refl, nb, refr)] = out
    # switches that cannot matter: reflection on with zero bounces or without a reflective material adds 0 to rgb (alpha aside)
    assert _ieq(frames[((0, 0), 0, 1, 0)][..., :3].contiguous(), frames[((0, 0), 1, 2, 1)][..., :3].contiguous())
    assert _ieq(frames[((1, 0), 0, 1, 0)][..., :3].contiguous(), frames[((1, 0), 1, 0, 0)][..., :3].contiguous())
    # the other kernel classes: a bulb (reflective material, reflection off / on), a textured scene, the procedural layers
    bulb = h.scene_mandelbulb(W, H)
    for k in range(3):
        bulb[1][0].cReflective[k] = 0.5
    for s in (abi.default_settings(fractalIters=8), abi.default_settings(fractalIters=8, enableReflection=1),
              abi.default_settings(fractalIters=8, enableReflection=1, numReflection=0)):
        assert_bit_equal(renderer.render(tables_of(bulb), s, W, H).cpu().numpy(), h.oracle_render(bulb, s, W, H), "bulb class")
    for name in ("skybox_reflect", "sea_sky"):
        scene, s, res = resource_case(name, W, H)
        for refl in (0, 1):
            s.enableReflection, s.enableRefraction = refl, refl
            t = tables_of(scene)
            for k, v in res.items():
                setattr(t, k, v)
            assert_bit_equal(renderer.render(t, s, W, H).cpu().numpy(), h.oracle_render(scene, s, W, H, **res), f"{name} secondary {refl}")


@pytest.mark.parametrize("bounces", [1, 2])
def test_reflection_refraction_bit_exact(renderer, bounces):
    W, H = 96, 64
    scene = reflect_refract_scene(W, H)
    s = abi.default_settings(enableReflection=1, enableRefraction=1, numReflection=bounces)
    ref = h.oracle_render(scene, s, W, H)
    plain = h.oracle_render(scene, abi.default_settings(), W, H)
    assert np.abs(ref - plain).max() > 0.05, "secondary rays must change the image"
    out = renderer.render(tables_of(scene), s, W, H)
    assert_bit_equal(out.cpu().numpy(), ref, "reflection+refraction")
    assert ref[..., 3].max() >= 2.0  # alpha accumulates per bounce (frag:2520, 2568, 2572)


def menger_scene(W, H):
    cam = h.make_camera((2.6, 2.2, 3.0), (-2.6, -2.2, -3.0), (0, 1, 0), 30.0, W, H)
    objs = (abi.RmObject * 1)(h.make_object(abi.RM_MENGERSPONGE, ambient=(.3, .3, .3), diffuse=(1, 1, 1),
                                            specular=(1, 1, 1), shininess=25.0, reflective=(.4, .4, .4)))
    lights = (abi.RmLight * 2)(h.make_light(abi.RM_LIGHT_DIRECTIONAL, (1, 1, 1), (-1, -1.5, -0.7)),
                               h.make_light(abi.RM_LIGHT_DIRECTIONAL, (.5, .5, .6), (1, -0.5, 0.3)))
    return cam, objs, 1, lights, 2, h.make_globals()


@pytest.mark.parametrize("levels,bounces,itime", [(4, 1, 0.0), (5, 2, 0.0), (4, 1, 7.5)])
def test_menger_bit_exact(renderer, levels, bounces, itime):
    W, H = 80, 60
    scene = menger_scene(W, H)
    scene[5].iTime = itime
    s = abi.default_settings(mengerLevels=levels, numReflection=bounces, enableReflection=1)
    ref = h.oracle_render(scene, s, W, H)
    out = renderer.render(tables_of(scene), s, W, H)
    assert_bit_equal(out.cpu().numpy(), ref, "menger")


@pytest.mark.parametrize("order", ["bulb_first", "sponge_first"])
@pytest.mark.parametrize("itime", [0.0, 4.2])
def test_bulb_hit_reads_the_trap_of_the_last_fractal_in_the_table(renderer, order, itime):
    """sdScene hands back the trap of the LAST fractal it evaluated, whichever object is nearest (UB3): a Mandelbulb hit in a table
    with a Menger sponge after it is coloured by the sponge's trap — all of it, .y (the product of the pairwise maxima) included.
    (Found by the round-4 soak: a sponge evaluation that kept only the .z a sponge hit reads differed on such pixels.)"""
    W, H = 72, 54
    cam = h.make_camera((1.6, 1.3, 4.0), (-1.6, -1.3, -4.0), (0, 1, 0), 40.0, W, H)
    # the two INTERSECT (the bulb pokes out of the sponge's faces): on most of the bulb's visible surface the sponge's level loop has
    # raised its running maximum at least once, so its trap — .y included — is not the initial one
    bulb = h.make_object(abi.RM_MANDELBULB, ambient=(.3, .2, .3), diffuse=(.9, 1, .8), specular=(1, 1, 1), shininess=30.0,
                         reflective=(.3, .3, .3))
    sponge = h.make_object(abi.RM_MENGERSPONGE, model=h.translate(0.1, -0.05, 0.0) @ h.scale(0.95, 0.95, 0.95), scale_factor=0.95,
                           ambient=(.3, .3, .3), diffuse=(1, 1, 1), specular=(1, 1, 1), shininess=25.0, reflective=(.4, .4, .4))
    objs = (abi.RmObject * 2)(*((bulb, sponge) if order == "bulb_first" else (sponge, bulb)))
    lights = (abi.RmLight * 2)(h.make_light(abi.RM_LIGHT_DIRECTIONAL, (1, 1, 1), (-1, -1.5, -0.7)),
                               h.make_light(abi.RM_LIGHT_POINT, (.8, .8, 1), pos=(2, 3, 4), func=(0.8, 0.05, 0.0)))
    scene = (cam, objs, 2, lights, 2, h.make_globals(itime=itime))
    for over in ({}, {"enableReflection": 1, "numReflection": 2, "mengerLevels": 5, "fractalIters": 8}):
        s = abi.default_settings(**over)
        ref = h.oracle_render(scene, s, W, H)
        assert_bit_equal(renderer.render(tables_of(scene), s, W, H).cpu().numpy(), ref, f"bulb + sponge, {order}, iTime {itime}")
    assert ref[..., :3].std() > 0.05


def test_mandelbrot_2d_and_julia(renderer):
    W, H = 64, 48
    scene = list(h.scene_mandelbulb(W, H))
    scene[5] = h.make_globals(two_d=1, itime=3.0)
    s = abi.default_settings()
    ref = h.oracle_render(tuple(scene), s, W, H)
    assert_bit_equal(renderer.render(tables_of(tuple(scene)), s, W, H).cpu().numpy(), ref, "2-D mandelbrot")
    assert ref[..., :3].std() > 0.01
    scene[5] = h.make_globals(julia=(0.35, -0.2), power=6.0)
    ref = h.oracle_render(tuple(scene), s, W, H)
    assert_bit_equal(renderer.render(tables_of(tuple(scene)), s, W, H).cpu().numpy(), ref, "julia bulb power 6")


def test_bulb_class_with_every_light_kind_and_kernel_path_requests(renderer):
    """A transformed bulb under a directional, a point and a spot light, with AO and soft shadows: the oracle's bits whichever
    schedule is requested — 1, 0 (auto) or 5 (does not apply to the bulb class: runs 1); the removed paths 2-4 are refused."""
    from raymarcher_amd import lib
    W, H = 150, 83
    cam = h.make_camera((0.5, 0.8, 4.0), (-0.5, -0.8, -4.0), (0, 1, 0), 35.0, W, H)
    M = h.translate(0.2, -0.1, 0.3) @ h.scale(1.3, 1.3, 1.3)
    objs = (abi.RmObject * 1)(h.make_object(abi.RM_MANDELBULB, model=M, scale_factor=1.3, ambient=(.3, .2, .3),
                                            diffuse=(.9, 1, .8), specular=(1, 1, 1), shininess=40.0))
    lights = (abi.RmLight * 3)(
        h.make_light(abi.RM_LIGHT_DIRECTIONAL, (1, 1, 1), (-0.4, -1, -0.5)),
        h.make_light(abi.RM_LIGHT_POINT, (1, .9, .7), pos=(3, 2, 4), func=(0.7, 0.05, 0.01)),
        h.make_light(abi.RM_LIGHT_SPOT, (.6, .8, 1), direction=(0, -1, -0.3), pos=(0, 5, 1.5), func=(1, 0, 0),
                     angle=np.deg2rad(30.0), penumbra=np.deg2rad(10.0)))
    scene = (cam, objs, 1, lights, 3, h.make_globals())
    for over in ({}, {"enableSoftShadow": 1, "enableAmbientOcclusion": 1, "fractalIters": 9, "maxSteps": 100},
                 {"features": abi.RM_FEAT_DARK_BACKGROUND, "fractalIters": 1}):
        s = abi.default_settings(**over)
        ref, ref_b = h.oracle_render(scene, s, W, H, bright=True)
        for path in (1, 5, 0):
            try:
                assert lib().rm_set_kernel_path(path) == 0
                a, ab = renderer.render(tables_of(scene), s, W, H, bright=True)
                assert lib().rm_debug_last_path() == 1
            finally:
                lib().rm_set_kernel_path(0)
            assert_bit_equal(a.cpu().numpy(), ref, f"path {path} vs oracle {over}")
            assert_bit_equal(ab.cpu().numpy(), ref_b, f"path {path} bright vs oracle {over}")
    for gone in (2, 3, 4, 6, -1):
        assert lib().rm_set_kernel_path(gone) == abi.RM_ERR_INVALID_ARGUMENT


def synthetic_textures():
    """Two procedural RGBA8 textures (rows bottom-up): a 37×23 colour gradient with a grid and a 64×64 checker."""
    rng = np.random.default_rng(11)
    yy, xx = np.mgrid[0:23, 0:37]
    a = np.stack([xx * 255 // 36, yy * 255 // 22, (xx * 7 + yy * 13) % 256, np.full_like(xx, 255)], -1).astype(np.uint8)
    a[::4, :, :3] //= 2
    yy, xx = np.mgrid[0:64, 0:64]
    b = np.where(((xx // 8 + yy // 8) % 2)[..., None] == 0, np.array([230, 40, 40, 255]), np.array([30, 60, 220, 255])).astype(np.uint8)
    b[..., :3] = np.clip(b[..., :3].astype(int) + rng.integers(-20, 20, (64, 64, 3)), 0, 255).astype(np.uint8)
    return [np.ascontiguousarray(a), np.ascontiguousarray(b)]


def textured_scene(W, H):
    """scenefiles/textures_tests in one frame: textured cube (floor), sphere, cone and cylinder + an untextured torus."""
    cam = h.make_camera((0.4, 2.2, 5.5), (-0.05, -0.35, -1), (0, 1, 0), 42.0, W, H)
    def tex(o, loc, ru, rv, blend):
        o.texLoc, o.repeatU, o.repeatV, o.blend = loc, ru, rv, blend
        return o
    objs = (abi.RmObject * 5)(
        tex(h.make_object(abi.RM_CUBE, model=h.translate(0, -0.8, 0) @ h.scale(7, 0.5, 7), scale_factor=0.5, ambient=(.2, .2, .2),
                          diffuse=(.9, .9, .9), specular=(.4, .4, .4), shininess=8), 1, 6.0, 6.0, 0.8),
        tex(h.make_object(abi.RM_SPHERE, model=h.translate(-1.5, 0.3, 0) @ h.scale(1.6, 1.6, 1.6), scale_factor=1.6,
                          ambient=(.1, .1, .1), diffuse=(1, 1, 1), specular=(1, 1, 1), shininess=30), 0, 2.0, 1.0, 1.0),
        tex(h.make_object(abi.RM_CONE, model=h.translate(0.3, 0.2, 0.8) @ h.scale(1.2, 1.5, 1.2), scale_factor=1.2,
                          ambient=(.1, .1, .1), diffuse=(.7, .9, .7), specular=(.5, .5, .5), shininess=12), 0, 3.0, 2.0, 0.5),
        tex(h.make_object(abi.RM_CYLINDER, model=h.translate(1.9, 0.2, -0.4) @ h.scale(1.1, 1.5, 1.1), scale_factor=1.1,
                          ambient=(.1, .1, .1), diffuse=(.9, .8, .6), specular=(.5, .5, .5), shininess=12), 1, 2.0, 1.0, 0.9),
        h.make_object(abi.RM_TORUS, model=h.translate(0, 1.6, -1.5) @ h.scale(2, 2, 2), scale_factor=2.0, ambient=(.1, .1, .2),
                      diffuse=(.3, .4, .9), specular=(1, 1, 1), shininess=20))
    lights = (abi.RmLight * 2)(h.make_light(abi.RM_LIGHT_DIRECTIONAL, (1, 1, 1), (-0.4, -1, -0.5)),
                               h.make_light(abi.RM_LIGHT_POINT, (.9, .8, .7), pos=(3, 4, 4), func=(0.7, 0.04, 0.0)))
    return cam, objs, 5, lights, 2, h.make_globals()


@pytest.mark.parametrize("over", [{}, {"features": abi.RM_FEAT_WHITE_BACKGROUND, "enableSoftShadow": 1, "enableAmbientOcclusion": 1},
                                  {"features": ENV_ALL}])
def test_textured_frames_bit_exact(renderer, over):
    W, H = 112, 72
    scene = textured_scene(W, H)
    texs = synthetic_textures()
    s = abi.default_settings(**over)
    ref = h.oracle_render(scene, s, W, H, textures=texs)
    plain = [o for o in scene[1]]
    t = tables_of(scene)
    t.textures = texs
    out = renderer.render(t, s, W, H)
    assert_bit_equal(out.cpu().numpy(), ref, f"textured {over}")
    # the textures really contribute
    for o in scene[1]:
        o.texLoc = -1
    assert np.abs(h.oracle_render(scene, s, W, H) - ref).max() > 0.05


def test_texture_errors(renderer):
    from raymarcher_amd import RaymarcherError
    W, H = 16, 16
    scene = textured_scene(W, H)
    t = tables_of(scene)
    with pytest.raises(RaymarcherError) as e:  # texLoc set but no textures supplied
        renderer.render(t, abi.default_settings(), W, H)
    assert e.value.status == abi.RM_ERR_UNSUPPORTED
    t.textures = synthetic_textures()
    t.objects[4].texLoc = 0  # torus: the reference has no uv map for it
    with pytest.raises(RaymarcherError) as e:
        renderer.render(t, abi.default_settings(), W, H)
    assert e.value.status == abi.RM_ERR_UNSUPPORTED


def test_scenefile_with_texture_end_to_end(renderer, tmp_path):
    """Scenefile → loader → PNG decode → upload → render, against the oracle fed the same decoded pixels."""
    from PIL import Image
    from raymarcher_amd.render import Scene
    (tmp_path / "scenes").mkdir()
    (tmp_path / "texture_store").mkdir()
    tex = synthetic_textures()[0]
    Image.fromarray(tex[::-1, :, :3]).save(tmp_path / "texture_store" / "grad.png")  # file rows are top-down
    scene_json = """{"globalData": {"ambientCoeff": 0.5, "diffuseCoeff": 0.5, "specularCoeff": 0.5},
      "cameraData": {"position": [0, 1.5, 4], "up": [0, 1, 0], "heightAngle": 40, "focus": [0, 0, 0]},
      "groups": [{"lights": [{"type": "directional", "color": [1, 1, 1], "direction": [-0.3, -1, -0.6]}]},
                 {"translate": [0, 0, 0], "scale": [2, 2, 2], "primitives": [{"type": "sphere", "diffuse": [1, 1, 1],
                   "ambient": [0.2, 0.2, 0.2], "specular": [1, 1, 1], "shininess": 20, "blend": 0.9,
                   "textureFile": "texture_store/grad.png", "textureU": 3, "textureV": 2}]}]}"""
    (tmp_path / "scenes" / "s.json").write_text(scene_json)
    W, H = 64, 48
    t = Scene(path=tmp_path / "scenes" / "s.json").tables(W, H)
    assert t.textures is not None and (t.textures[0] == tex[..., :4]).all()
    s = abi.default_settings()
    ref = h.oracle_render((t.camera, t.objects, t.num_objects, t.lights, t.num_lights, t.globals_), s, W, H, textures=t.textures)
    assert_bit_equal(renderer.render(t, s, W, H).cpu().numpy(), ref, "scenefile with texture")


POST_CASES = {
    "gamma": dict(enableGammaCorrection=1),
    "hdr": dict(enableHDR=1, exposure=1.7),
    "bloom": dict(enableBloom=1, exposure=1.0),
    "bloom_hdr_fxaa": dict(enableBloom=1, enableHDR=1, enableFXAA=1, exposure=0.8),
    "fxaa_only": dict(enableFXAA=1),
    "gamma_fxaa": dict(enableGammaCorrection=1, enableFXAA=1),
    "none": dict(),
}


@pytest.mark.parametrize("name", list(POST_CASES))
def test_post_passes_bit_exact(renderer, name):
    """applyLightEffects + applyFXAA (realtimerender.cpp:78-165) on a rendered frame with bright pixels."""
    W, H = 150, 90  # not multiples of the 256-wide blocks
    scene = reflect_refract_scene(W, H)
    for li in scene[3]:
        li.color[0] *= 2.5; li.color[1] *= 2.5; li.color[2] *= 2.5  # over-exposed: BrightColor is populated
    s = abi.default_settings(enableReflection=1)
    frag, bright = renderer.render(tables_of(scene), s, W, H, bright=True)
    assert float(bright[..., :3].max()) > 1.0
    post = abi.RmPostSettings(**{"exposure": 1.0, **POST_CASES[name]})
    got = renderer.post_process(frag, bright, post).cpu().numpy()
    ref = h.oracle_post(frag.cpu().numpy(), bright.cpu().numpy(), post)
    assert_bit_equal(got, ref, f"post {name}")
    if name != "none":
        assert np.abs(got[..., :3] - frag.cpu().numpy()[..., :3]).max() > 0.01
    img = renderer.to_rgba8(renderer.post_process(frag, bright, post)).cpu().numpy()
    exp = (np.clip(ref[::-1], 0, 1) * np.float32(255.0) + np.float32(0.5)).astype(np.uint8)
    assert (img == exp).all()


@pytest.mark.parametrize("W,H", [(700, 45), (257, 33), (64, 32), (1, 1), (3, 70)])
def test_post_passes_bit_exact_on_wide_synthetic_frames(renderer, W, H):
    """Several 64×32 blur tiles and 256-pixel row blocks with ragged right / bottom edges (and frames smaller than one apron): the
    fused first pass (float BrightColor staged as binary16) and the fused last pass (horizontal blur inside the composite) against
    the oracle's ten-pass loop (realtimerender.cpp:92-108), on random frames with sparse bright pixels."""
    import torch
    rng = np.random.default_rng(W * 1000 + H)
    frag = rng.random((H, W, 4), dtype=np.float32) * np.float32(1.6)
    frag[..., 3] = 1.0
    luma = (frag[..., :3] * np.array([0.2126, 0.7152, 0.0722], dtype=np.float32)).sum(-1, keepdims=True)
    bright = np.where(luma > 1.0, frag, np.float32(0.0)).astype(np.float32)
    bright[..., 3] = 1.0
    fd, bd = torch.from_numpy(frag).to(renderer.device), torch.from_numpy(bright).to(renderer.device)
    for name in ("bloom", "bloom_hdr_fxaa", "hdr", "gamma_fxaa"):
        post = abi.RmPostSettings(**{"exposure": 1.0, **POST_CASES[name]})
        got = renderer.post_process(fd, bd, post).cpu().numpy()
        assert_bit_equal(got, h.oracle_post(frag, bright, post), f"post {name} {W}x{H}")


def test_post_full_size_4k(renderer):
    """3840×2160: determinism, and bit equality with the oracle on a horizontal band that contains its whole
    9-tap / FXAA neighbourhood (bloom off so rows do not depend on far rows)."""
    import torch
    W, H = 3840, 2160
    from raymarcher_amd import scenes
    t = scenes.mandelbulb(W, H)
    s = abi.default_settings(fractalIters=12)
    frag, bright = renderer.render(t, s, W, H, bright=True)
    post = abi.RmPostSettings(enableFXAA=1, enableHDR=1, exposure=1.3)
    a = renderer.post_process(frag, bright, post)
    b = renderer.post_process(frag, bright, post)
    assert (a.view(dtype=torch.int32) == b.view(dtype=torch.int32)).all()
    post_b = abi.RmPostSettings(enableBloom=1, enableHDR=1, enableFXAA=1, exposure=1.0)
    c = renderer.post_process(frag, bright, post_b)
    assert torch.isfinite(c).all() and float(c[..., :3].max()) <= 1.0 and float(c[..., :3].min()) >= 0.0


# ---------------------------------------------------------------- samplers beyond object textures: noise (night sky, sea),
# sky box, LTC tables (area lights).  All inputs are synthetic — the ABI takes them as data.
def synthetic_noise():
    """256×256 RGBA8 with DIFFERENT channels (the reference's noise_texture_1.png is grey; distinct channels also
    exercise noiseV's .yx swizzle).  A few texels are pushed to 255 so that stars (noise > 0.99) exist."""
    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, (256, 256, 4), dtype=np.uint8)
    a[rng.integers(0, 256, 900), rng.integers(0, 256, 900), :2] = 255
    a[..., 3] = 255
    return np.ascontiguousarray(a)


def synthetic_skybox(n=24):
    """Six n×n RGBA8 faces, each a different two-colour gradient with a bright spot (bloom source)."""
    yy, xx = np.mgrid[0:n, 0:n]
    faces = []
    for f in range(6):
        c0 = np.array([(f * 40) % 256, (255 - f * 30) % 256, (f * 90 + 30) % 256])
        c1 = np.array([(200 + f * 10) % 256, (f * 50) % 256, (120 + f * 20) % 256])
        t = ((xx + (f + 1) * yy) / ((f + 2) * (n - 1.0)))[..., None]
        img = c0 * (1 - t) + c1 * t
        img[(xx - n // 3) ** 2 + (yy - n // 2) ** 2 < 6] = 255
        faces.append(np.ascontiguousarray(np.concatenate([img, np.full((n, n, 1), 255)], -1).astype(np.uint8)))
    return faces


def synthetic_ltc():
    """Smooth stand-ins for the LTC tables (float, 64×64×4; u = column): t1 ≈ the inverse-matrix parameters,
    t2 = (fresnel scale, fresnel bias, unused, horizon-clipping form factor)."""
    v, u = np.mgrid[0:64, 0:64] / 63.0
    t1 = np.stack([0.55 + 0.45 * v, 0.25 * u * v, 0.15 * (1 - v), 0.5 + 0.5 * np.sqrt(v)], -1)
    t2 = np.stack([0.9 - 0.5 * v, 0.1 + 0.3 * u, 0 * u, np.clip(0.35 + 0.65 * u + 0.1 * v, 0, 1.2)], -1)
    return t1.astype(np.float32), t2.astype(np.float32)


def night_scene(W, H):
    cam = h.make_camera((1.6, 0.4, -5), (-0.42, 0.36, 1), (0, 1, 0), 60.0, W, H)  # looks toward MOON (frag:107)
    objs = (abi.RmObject * 2)(
        h.make_object(abi.RM_SPHERE, model=h.translate(-0.9, 0, 0) @ h.scale(1.5, 1.5, 1.5), scale_factor=1.5, ambient=(.1, .1, .15),
                      diffuse=(.5, .5, .7), specular=(1, 1, 1), shininess=25, reflective=(.9, .9, .9)),
        h.make_object(abi.RM_CUBE, model=h.translate(1.3, -0.2, 0.4) @ h.scale(1.1, 1.1, 1.1), scale_factor=1.1, ambient=(.1, .1, .1),
                      diffuse=(.7, .4, .3), specular=(.5, .5, .5), shininess=10))
    lights = (abi.RmLight * 1)(h.make_light(abi.RM_LIGHT_DIRECTIONAL, (.9, .9, 1), (0.4, -0.4, -0.3)))
    return cam, objs, 2, lights, 1, h.make_globals(itime=1.3)


def sea_scene(W, H):
    cam = h.make_camera((0, 3.5, 6), (0, -0.35, -1), (0, 1, 0), 50.0, W, H, far=100.0)
    objs = (abi.RmObject * 1)(
        h.make_object(abi.RM_SPHERE, model=h.translate(0, 1.8, -1.5) @ h.scale(2, 2, 2), scale_factor=2.0, ambient=(.2, .2, .2),
                      diffuse=(.8, .3, .2), specular=(1, 1, 1), shininess=20, reflective=(.6, .6, .6)))
    lights = (abi.RmLight * 1)(h.make_light(abi.RM_LIGHT_DIRECTIONAL, (1, 1, 1), (-0.4, -1, -0.3)))
    return cam, objs, 1, lights, 1, h.make_globals(itime=0.7)


def area_light_scene(W, H):
    """A floor, a sphere and a torus under one rectangular area light with its emissive rectangle
    (RayMarchScene::initScene appends one per area light, raymarchscene.cpp:121-133) plus a point light."""
    cam = h.make_camera((0, 1.6, 5.5), (0, -0.2, -1), (0, 1, 0), 45.0, W, H)
    ctm = h.translate(0.3, 2.2, -1.0) @ rot_x(np.deg2rad(65.0)) @ h.scale(2.4, 1.4, 1.0)
    rect = h.make_object(abi.RM_RECTANGLE, model=ctm, scale_factor=1.0)
    rect.isEmissive, rect.lightIdx = 1, 0
    rect.color[0], rect.color[1], rect.color[2] = 1.0, 0.9, 0.6
    objs = (abi.RmObject * 4)(
        h.make_object(abi.RM_CUBE, model=h.translate(0, -1.0, 0) @ h.scale(9, 0.4, 9), scale_factor=0.4, ambient=(.1, .1, .1),
                      diffuse=(.7, .7, .7), specular=(.6, .6, .6), shininess=12, reflective=(.25, .25, .25)),
        h.make_object(abi.RM_SPHERE, model=h.translate(-1.2, 0, 0.2) @ h.scale(1.5, 1.5, 1.5), scale_factor=1.5, ambient=(.1, .1, .1),
                      diffuse=(.3, .5, .9), specular=(1, 1, 1), shininess=40),
        h.make_object(abi.RM_TORUS, model=h.translate(1.4, -0.2, 0) @ h.scale(1.8, 1.8, 1.8), scale_factor=1.8, ambient=(.1, .1, .1),
                      diffuse=(.9, .5, .2), specular=(.8, .8, .8), shininess=20),
        rect)
    area = h.make_light(abi.RM_LIGHT_AREA, (1.0, 0.9, 0.6), func=(1, 0, 0))
    area.intensity, area.twoSided = 0.0, 1  # sceneparser.cpp:18-30 drops the parsed intensity; twoSided is always set
    corners = [(-0.5, 0.5, 0), (0.5, 0.5, 0), (0.5, -0.5, 0), (-0.5, -0.5, 0)]  # realtime.h:136-141
    for k, c in enumerate(corners):
        w = ctm @ np.array([*c, 1.0])
        for j in range(3):
            area.points[k][j] = float(np.float32(w[j]))
    lights = (abi.RmLight * 2)(area, h.make_light(abi.RM_LIGHT_POINT, (.5, .5, .6), pos=(-3, 3, 3), func=(0.8, 0.05, 0)))
    return cam, objs, 4, lights, 2, h.make_globals()


def rot_x(a):
    M = np.eye(4)
    M[1, 1], M[1, 2], M[2, 1], M[2, 2] = np.cos(a), -np.sin(a), np.sin(a), np.cos(a)
    return M


def resource_case(name, W, H):
    """name → (scene, settings, resources dict) of the sampler-driven cases."""
    WB = abi.RM_FEAT_WHITE_BACKGROUND
    if name == "night_sky":
        return night_scene(W, H), abi.default_settings(features=abi.RM_FEAT_NIGHTSKY_BACKGROUND, enableReflection=1), {"noise": synthetic_noise()}
    if name == "sea_sky":
        return sea_scene(W, H), abi.default_settings(features=abi.RM_FEAT_SEA | abi.RM_FEAT_SKY_BACKGROUND, enableReflection=1), \
            {"noise": synthetic_noise()}
    if name == "sea_terrain_cloud":
        sc = sea_scene(W, H)
        sc = (h.make_camera((0, 700, 6), (0, -0.2, -1), (0, 1, 0), 50.0, W, H),) + sc[1:]
        return sc, abi.default_settings(features=ENV_ALL | abi.RM_FEAT_SEA), {"noise": synthetic_noise()}
    if name == "sea_terrain":
        sc = sea_scene(W, H)
        sc = (h.make_camera((0, 700, 6), (0, -0.2, -1), (0, 1, 0), 50.0, W, H, far=2000.0),) + sc[1:]
        return sc, abi.default_settings(features=abi.RM_FEAT_SKY_BACKGROUND | abi.RM_FEAT_TERRAIN | abi.RM_FEAT_SEA), {"noise": synthetic_noise()}
    if name == "skybox_reflect":
        return reflect_refract_scene(W, H), abi.default_settings(features=WB, enableSkyBox=1, enableReflection=1, enableRefraction=1), \
            {"skybox": synthetic_skybox()}
    if name == "area_light":
        t1, t2 = synthetic_ltc()
        return area_light_scene(W, H), abi.default_settings(features=WB, enableReflection=1), \
            {"ltc1": h.oracle_ltc_quantise(t1), "ltc2": h.oracle_ltc_quantise(t2)}
    if name == "area_light_soft_bump":
        t1, t2 = synthetic_ltc()
        return area_light_scene(W, H), abi.default_settings(enableSoftShadow=1, enableAmbientOcclusion=1), \
            {"ltc1": h.oracle_ltc_quantise(t1), "ltc2": h.oracle_ltc_quantise(t2)}
    if name == "area_light_bump_ao":
        t1, t2 = synthetic_ltc()
        return area_light_scene(W, H), abi.default_settings(enableAmbientOcclusion=1), \
            {"ltc1": h.oracle_ltc_quantise(t1), "ltc2": h.oracle_ltc_quantise(t2)}
    raise KeyError(name)


RESOURCE_CASES = ["night_sky", "sea_sky", "sea_terrain", "sea_terrain_cloud", "skybox_reflect", "area_light", "area_light_soft_bump",
                  "area_light_bump_ao"]


@pytest.mark.parametrize("name", RESOURCE_CASES)
def test_resource_frames_bit_exact(renderer, name):
    W, H = 96, 64
    scene, s, res = resource_case(name, W, H)
    ref, ref_b = h.oracle_render(scene, s, W, H, bright=True, **res)
    t = tables_of(scene)
    for k, v in res.items():
        setattr(t, k, v)
    out, br = renderer.render(t, s, W, H, bright=True)
    assert_bit_equal(out.cpu().numpy(), ref, name)
    assert_bit_equal(br.cpu().numpy(), ref_b, name + " bright")
    assert np.isfinite(ref).all()


def test_resource_frames_in_row_tiles(renderer):
    """rm_render_tiles_res: the sharded render of sampler-driven scenes equals the single launch."""
    import torch
    from raymarcher_amd import lib
    W, H, T, N = 96, 70, 8, 3
    for name in ("sea_sky", "area_light", "skybox_reflect"):
        scene, s, res = resource_case(name, W, H)
        t = tables_of(scene)
        for k, v in res.items():
            setattr(t, k, v)
        full = renderer.render(t, s, W, H)
        slot = lib().rm_shard_rows(H, T, 0, N)
        gathered = torch.zeros((N * slot, W, 4), dtype=torch.float32, device=full.device)
        for k in range(N):
            mine = renderer.render_tiles(t, s, W, H, T, k, N)
            gathered[k * slot:k * slot + mine.shape[0]] = mine
        frame = renderer.deinterleave(gathered, W, H, T, N, slot)
        assert (frame.view(dtype=torch.int32) == full.view(dtype=torch.int32)).all(), name


def test_host_pointers_are_refused(renderer):
    """A host pointer where device memory is expected must come back as an error, never reach a kernel."""
    from raymarcher_amd import RaymarcherError, lib
    from raymarcher_amd._lib import check
    W, H = 16, 16
    t = tables_of(h.scene_mandelbulb(W, H))
    s = abi.default_settings()
    host = np.zeros((H, W, 4), np.float32)
    with pytest.raises(RaymarcherError) as e:
        check(lib().rm_render(*t.args(s), W, H, 0, H, C.c_void_p(host.ctypes.data), None, None))
    assert e.value.status == abi.RM_ERR_INVALID_ARGUMENT and "device" in str(e.value)
    scene, s2, res = resource_case("night_sky", W, H)
    hres, _keep = h.host_resources(**res)          # HOST pixel pointers
    out = renderer.torch.empty((H, W, 4), dtype=renderer.torch.float32, device=renderer.device)
    with pytest.raises(RaymarcherError) as e:
        check(lib().rm_render_res(*tables_of(scene).args(s2), C.byref(hres), W, H, 0, H, C.c_void_p(out.data_ptr()), None, None))
    assert e.value.status == abi.RM_ERR_INVALID_ARGUMENT


def test_resource_errors(renderer):
    """A feature whose sampler was not supplied is refused, on both sides, with RM_ERR_UNSUPPORTED."""
    from raymarcher_amd import RaymarcherError
    W, H = 16, 16
    for name in ("night_sky", "sea_sky", "skybox_reflect", "area_light"):
        scene, s, _res = resource_case(name, W, H)
        with pytest.raises(RaymarcherError) as e:
            renderer.render(tables_of(scene), s, W, H)
        assert e.value.status == abi.RM_ERR_UNSUPPORTED, name
        h.oracle_render(scene, s, W, H, expect=abi.RM_ERR_UNSUPPORTED)
    assert (renderer.torch.from_numpy(h.oracle_ltc_quantise(synthetic_ltc()[0])).numpy() ==
            __import__("raymarcher_amd").render.ltc_quantise(synthetic_ltc()[0])).all()


# ---------------------------------------------------------------- edge cases of the boundary
def test_row_ranges_and_ragged_sizes(renderer):
    W, H = 37, 29  # not multiples of the 8×8 wave tile
    scene = all_primitives_scene(W, H)
    s = abi.default_settings(maxSteps=48)
    ref = h.oracle_render(scene, s, W, H)
    full = renderer.render(tables_of(scene), s, W, H).cpu().numpy()
    assert_bit_equal(full, ref, "ragged frame")
    part = renderer.render(tables_of(scene), s, W, H, 5, 18).cpu().numpy()
    assert_bit_equal(part, ref[5:18], "row range")
    empty = renderer.render(tables_of(scene), s, W, H, 7, 7)
    assert empty.shape[0] == 0


def test_empty_scene_is_background(renderer):
    W, H = 16, 8
    cam = h.make_camera((0, 0, 3), (0, 0, -1), (0, 1, 0), 45.0, W, H)
    scene = (cam, (abi.RmObject * 1)(), 0, (abi.RmLight * 1)(), 0, h.make_globals())
    out = renderer.render(tables_of(scene), abi.default_settings(), W, H).cpu().numpy()
    assert (out == 1.0).all()  # WHITE_BACKGROUND, alpha 1
    assert_bit_equal(out, h.oracle_render(scene, abi.default_settings(), W, H), "empty scene")


def test_tiles_gather_roundtrip(renderer):
    import torch
    from raymarcher_amd import lib
    W, H, T = 48, 50, 8  # last tile is partial (50 = 6·8 + 2)
    scene = h.scene_mandelbulb(W, H)
    s = abi.default_settings(fractalIters=12)
    ref = h.oracle_render(scene, s, W, H)
    for shards in (1, 2, 3, 4):
        parts = [renderer.render_tiles(tables_of(scene), s, W, H, T, k, shards) for k in range(shards)]
        assert sum(p.shape[0] for p in parts) == H
        for k, p in enumerate(parts):
            rows = [lib().rm_shard_row_to_frame(H, T, k, shards, r) for r in range(p.shape[0])]
            assert_bit_equal(p.cpu().numpy(), ref[rows], f"shard {k}/{shards}")
        frame = renderer.deinterleave(torch.cat(parts, 0).contiguous(), W, H, T, shards)
        assert_bit_equal(frame.cpu().numpy(), ref, f"deinterleave {shards}")


def test_tiles_gather_roundtrip_with_root_relief(renderer):
    """rm_set_root_relief(K): shard 0 (the gather's root) owns (K − 1)/K of a peer's tiles.  Every shard's rows are the frame's, the
    packed concatenation and the slotted gather layout (equal slots of the LARGEST shard — shard 1 now) de-interleave to the frame,
    as float4 and as RGBA8 (flipped and not)."""
    import torch
    from raymarcher_amd import lib
    L = lib()
    W, H, T = 48, 210, 8  # 27 tiles, the last one partial
    scene = h.scene_mandelbulb(W, H)
    t = tables_of(scene)
    s = abi.default_settings(fractalIters=10)
    ref = h.oracle_render(scene, s, W, H)
    full = renderer.render(t, s, W, H)
    img = renderer.to_rgba8(full)  # flipped: row 0 = top
    try:
        for shards, K in ((2, 2), (3, 4), (4, 8), (8, 8), (8, 2)):
            assert L.rm_set_root_relief(K) == 0
            parts = [renderer.render_tiles(t, s, W, H, T, k, shards) for k in range(shards)]
            assert [p.shape[0] for p in parts] == [L.rm_shard_rows(H, T, k, shards) for k in range(shards)] and sum(p.shape[0] for p in parts) == H
            assert parts[0].shape[0] <= parts[1].shape[0]  # the root is relieved (a frame shorter than one cycle of the deal may not show it)
            if K == 2:
                assert parts[0].shape[0] < parts[1].shape[0]
            for k, p in enumerate(parts):
                rows = [L.rm_shard_row_to_frame(H, T, k, shards, r) for r in range(p.shape[0])]
                assert_bit_equal(p.cpu().numpy(), ref[rows], f"shard {k}/{shards}, relief {K}")
            assert _ieq(renderer.deinterleave(torch.cat(parts, 0).contiguous(), W, H, T, shards), full)
            slot = L.rm_gather_slot_rows(H, T, shards)
            assert slot == max(p.shape[0] for p in parts)
            gathered = torch.zeros((shards * slot, W, 4), dtype=torch.float32, device=full.device)
            g8 = torch.zeros((shards * slot, W, 4), dtype=torch.uint8, device=full.device)
            for k, p in enumerate(parts):
                gathered[k * slot:k * slot + p.shape[0]] = p
                g8[k * slot:k * slot + p.shape[0]] = renderer.tiles_to_rgba8(p)
            assert _ieq(renderer.deinterleave(gathered, W, H, T, shards, slot), full)
            assert bool((renderer.deinterleave_rgba8(g8, W, H, T, shards, slot, flip=True) == img).all())
            assert bool((renderer.deinterleave_rgba8(g8, W, H, T, shards, slot, flip=False) == img.flip(0)).all())
    finally:
        L.rm_set_root_relief(0)


def test_unsupported_and_invalid_inputs(renderer):
    from raymarcher_amd import RaymarcherError
    W, H = 8, 8
    scene = h.scene_mandelbulb(W, H)
    t = tables_of(scene)
    with pytest.raises(RaymarcherError) as e:
        renderer.render(t, abi.default_settings(features=abi.RM_FEAT_SEA), W, H)
    assert e.value.status == abi.RM_ERR_UNSUPPORTED
    with pytest.raises(RaymarcherError) as e:
        renderer.render(t, abi.default_settings(), W, H, 4, 12)
    assert e.value.status == abi.RM_ERR_INVALID_ARGUMENT
    t.objects[0].texLoc = 0
    with pytest.raises(RaymarcherError) as e:
        renderer.render(t, abi.default_settings(), W, H)
    assert e.value.status == abi.RM_ERR_UNSUPPORTED
    t.objects[0].texLoc = -1
    t.num_objects = 31
    with pytest.raises(RaymarcherError) as e:
        renderer.render(t, abi.default_settings(), W, H)
    assert e.value.status == abi.RM_ERR_CAPACITY


def test_counters_match_oracle(renderer):
    W, H = 64, 36
    scene = h.scene_mandelbulb(W, H)
    s = abi.default_settings()
    _, cnt = h.oracle_render(scene, s, W, H, counters=True)
    out, gcnt = renderer.render_counted(tables_of(scene), s, W, H)
    assert (gcnt.sceneEvals, gcnt.bulbIters, gcnt.hitPixels) == (cnt.sceneEvals, cnt.bulbIters, cnt.hitPixels)
    assert gcnt.shadedPoints == cnt.shadedPoints == cnt.hitPixels and gcnt.terrainEvals == gcnt.cloudEvals == 0


def test_counters_of_every_kernel_class_match_oracle(renderer):
    """The units of bench.py's work model — sdScene evaluations, shaded points (bounce hits included), terrain / cloud noise
    evaluations — from the counting instantiations of the plain, layer and sampler kernels against the oracle's own count, and
    the counted frames against the production frames."""
    W, H = 64, 40
    cases = [(reflect_refract_scene(W, H), abi.default_settings(enableReflection=1, enableRefraction=1, numReflection=2), {}),
             (env_scene(W, H), abi.default_settings(features=ENV_ALL, enableReflection=1), {}),
             (textured_scene(W, H), abi.default_settings(), {"textures": synthetic_textures()})]
    for scene, s, res in cases:
        ref, cnt = h.oracle_render(scene, s, W, H, counters=True, **res)
        t = tables_of(scene)
        for k, v in res.items():
            setattr(t, k, v)
        out, g = renderer.render_counted(t, s, W, H)
        assert (g.sceneEvals, g.bulbIters, g.hitPixels, g.shadedPoints, g.terrainEvals, g.cloudEvals, g.shapeEvals) == \
               (cnt.sceneEvals, cnt.bulbIters, cnt.hitPixels, cnt.shadedPoints, cnt.terrainEvals, cnt.cloudEvals, cnt.shapeEvals)
        assert_bit_equal(out.cpu().numpy(), ref, "counted frame")
        assert g.shapeEvals == g.sceneEvals * scene[2]  # as the shader is written: every object at every evaluation
    assert cnt.shadedPoints > 0
    # the executed-work count of the plain class: the table walk passes over objects, the same frame
    scene, s, _ = cases[0]
    out, g1 = renderer.render_counted(tables_of(scene), s, W, H)
    out2, g2 = renderer.render_counted(tables_of(scene), s, W, H, abi.RM_COUNT_EXECUTED)
    assert _ieq(out, out2) and g2.sceneEvals <= g1.sceneEvals and 0 < g2.shapeEvals < g1.shapeEvals


def test_rgba8_flip_and_png(renderer, tmp_path):
    W, H = 40, 24
    scene = h.scene_mandelbulb(W, H)
    s = abi.default_settings(fractalIters=12)
    frame = renderer.render(tables_of(scene), s, W, H)
    img = renderer.to_rgba8(frame).cpu().numpy()
    ref = frame.cpu().numpy()[::-1]
    exp = (np.clip(ref, 0, 1) * np.float32(255.0) + np.float32(0.5)).astype(np.uint8)
    assert (img == exp).all()
    path = tmp_path / "bulb.png"
    renderer.save_png(frame, path)
    from PIL import Image
    assert (np.asarray(Image.open(path)) == img).all()


# ---------------------------------------------------------------- full-size, size-independent properties
def test_full_size_properties_4k_bulb(renderer):
    """BASELINE.json's size (3840×2160), the benchmark frame itself: (i) determinism — also across the tile-order modes,
    which only change WHEN a tile starts; (ii) row-range renders of arbitrary bands equal the same rows of the full frame;
    (iii) the WHOLE frame equals the oracle's frame bit for bit (the oracle needs ≈2 s on the GPU box's 16 cores)."""
    import torch
    from raymarcher_amd import lib, scenes
    W, H = 3840, 2160
    t = scenes.mandelbulb(W, H)
    s = abi.default_settings(fractalIters=12)
    a = renderer.render(t, s, W, H).clone()
    b = renderer.render(t, s, W, H)  # second frame: tiles start heaviest-first by the first frame's costs
    assert _ieq(a, b)
    lib().rm_set_tile_order(0)
    try:
        assert _ieq(a, renderer.render(t, s, W, H))  # raster order
    finally:
        lib().rm_set_tile_order(-1)
    band = renderer.render(t, s, W, H, 1000, 1100)
    assert _ieq(band, a[1000:1100])
    ref = h.oracle_render(_scene_tuple(t), s, W, H, threads=16)
    assert_bit_equal(a.cpu().numpy(), ref, "the whole 4K Mandelbulb frame")
    hit = float((a[..., 0] != 1.0).float().mean())
    assert 0.25 < hit < 0.40  # ≈0.33 of the pixels hit the bulb


def test_tile_order_of_new_and_repeated_pictures_never_changes_a_pixel(renderer):
    """Frames of at least 2048 tiles start their tiles in an order the launcher derives: measured costs of the previous frame for
    the SAME picture, a geometric classification (tile_geom_kernel: objects' bounding balls) — combined with stale costs — for a
    NEW picture (first frame, moved camera, changed scene, another row range or shard), raster order where the scene gives no
    balls (an object without a bound, procedural layers).  Whatever the order: the frame of rm_set_tile_order(0), bit for bit,
    and the oracle's."""
    import torch
    from raymarcher_amd import lib
    W, H = 640, 384  # 80 x 48 = 3840 tiles
    L = lib()

    def frames(seq, **kw):
        outs = []
        for scene, s in seq:
            outs.append(renderer.render(tables_of(scene), s, W, H, **kw).clone())
        return outs

    prim = all_primitives_scene(W, H)
    moved = (h.make_camera((0.6, 2.2, 6.5), (-0.1, -0.3, -1), (0, 1, 0), 45.0, W, H),) + prim[1:]
    bulb = h.scene_mandelbulb(W, H)
    sier = (prim[0], (abi.RmObject * 2)(h.make_object(abi.RM_SIERPINSKI, model=h.scale(0.8, 0.8, 0.8), scale_factor=0.8, diffuse=(.8, .6, .3)),
                                        h.make_object(abi.RM_SPHERE, model=h.translate(1.5, 0, 0))), 2) + prim[3:]
    s0 = abi.default_settings(maxSteps=96)
    s1 = abi.default_settings(maxSteps=96, enableAmbientOcclusion=1)
    sb = abi.default_settings(fractalIters=8, maxSteps=96)
    # new, same, same, moved camera, back, changed settings, another class, an unbounded object (raster fallback), the layers
    seq = [(prim, s0), (prim, s0), (prim, s0), (moved, s0), (prim, s0), (prim, s1), (bulb, sb), (bulb, sb), (sier, s0), (sier, s0),
           (env_scene(W, H), abi.default_settings(features=ENV_ALL, maxSteps=64))]
    try:
        assert L.rm_set_tile_order(0) == 0
        want = frames(seq)
        assert L.rm_set_tile_order(1) == 0
        got = frames(seq)
        for k, (a, b) in enumerate(zip(got, want)):
            assert _ieq(a, b), f"frame {k} of the sequence differs from its raster-order render"
        for k in (0, 3, 6):
            scene, s = seq[k]
            assert_bit_equal(got[k].cpu().numpy(), h.oracle_render(scene, s, W, H), f"frame {k} vs oracle")
        # a tall frame so that a row range and a shard still have >= 2048 tiles: other row maps are other pictures
        H2 = 1024
        scene = (h.make_camera((0, 1.2, 6), (0, -0.2, -1), (0, 1, 0), 45.0, W, H2),) + prim[1:]
        t = tables_of(scene)
        L.rm_set_tile_order(0)
        full = renderer.render(t, s0, W, H2).clone()
        L.rm_set_tile_order(1)
        for _ in range(2):
            assert _ieq(renderer.render(t, s0, W, H2), full)
            assert _ieq(renderer.render(t, s0, W, H2, row_begin=200, row_end=904), full[200:904])
            mine = renderer.render_tiles(t, s0, W, H2, 8, 1, 2)
            rows = [L.rm_shard_row_to_frame(H2, 8, 1, 2, i) for i in range(mine.shape[0])]
            assert _ieq(mine, full[torch.tensor(rows, device=full.device)])
        # a picture that repeats settles (the fourth cost-ordered frame on reuses the third's order: no ordering launches, no cost
        # atomics); the picture after a settled one is ordered by geometry + the costs the last sort kept
        settle = [(bulb, sb)] * 7 + [(prim, s0)] + [(bulb, sb)] * 6 + [(moved, s0)] * 14 + [(prim, s0), (moved, s0)]
        L.rm_set_tile_order(0)
        want = {id(sc): renderer.render(tables_of(sc), st, W, H).clone() for sc, st in ((bulb, sb), (prim, s0), (moved, s0))}
        L.rm_set_tile_order(1)
        for k, (sc, st) in enumerate(settle):
            assert _ieq(renderer.render(tables_of(sc), st, W, H), want[id(sc)]), f"frame {k} of the settling sequence"
    finally:
        L.rm_set_tile_order(-1)


@pytest.mark.parametrize("soft,ao,shape", [(1, 1, 3), (0, 0, 3), (1, 0, 2)])
def test_light_split_of_a_settled_picture_never_changes_a_pixel(renderer, soft, ao, shape):
    """A settled picture of the plain table-walk class with several lights renders its heaviest tiles one light per workgroup; the
    last of a tile's workgroups to arrive finishes it from the stored results (rm_kernels.hip, "light split"; forced here — by
    default the launcher measures per picture whether it pays).  Every frame of the sequence —
    new picture, cost-ordered repeats, settled and split — is the first frame, which is the oracle's; lights of every plain kind,
    one of them facing away from most of the scene (dropped by N·L on many pixels), soft and hard shadows, both tile shapes, a
    ragged frame size, and a row range."""
    from raymarcher_amd import lib
    L = lib()
    W, H = 636, 388  # 80 x 49 8×8 tiles, ragged on both edges
    prim = all_primitives_scene(W, H)
    lights = (abi.RmLight * 4)(
        h.make_light(abi.RM_LIGHT_DIRECTIONAL, (1, 1, 1), (-0.4, -1, -0.5)),
        h.make_light(abi.RM_LIGHT_POINT, (1, .9, .7), pos=(3, 2, 4), func=(0.7, 0.05, 0.01)),
        h.make_light(abi.RM_LIGHT_SPOT, (.6, .8, 1), direction=(0, -1, -0.3), pos=(0, 5, 1.5), func=(1, 0, 0),
                     angle=np.deg2rad(30.0), penumbra=np.deg2rad(10.0)),
        h.make_light(abi.RM_LIGHT_DIRECTIONAL, (.5, .5, .4), (0.9, 0.2, 0.1)))
    scene = (prim[0], prim[1], prim[2], lights, 4, prim[5])
    s = abi.default_settings(maxSteps=96, enableSoftShadow=soft, enableAmbientOcclusion=ao)
    t = tables_of(scene)
    try:
        assert L.rm_debug_set_tile_shape(shape) == 0
        assert L.rm_debug_set_light_split(32) == 0  # the heaviest 1/32 of the tiles (default 1/256)
        first = renderer.render(t, s, W, H).clone()
        assert_bit_equal(first.cpu().numpy(), h.oracle_render(scene, s, W, H), "first frame vs oracle")
        assert L.rm_debug_last_split() == 0
        split = 0
        for k in range(8):
            assert _ieq(renderer.render(t, s, W, H), first), f"repeat {k + 1} differs"
            split = max(split, L.rm_debug_last_split())
        tiles = (-(-W // 8)) * (-(-H // 8)) if shape == 3 else (-(-W // 4)) * (-(-H // 16))
        assert split == tiles // 32, "the settled picture was not split"
        # a row range is another picture: it settles and splits on its own
        for k in range(7):
            assert _ieq(renderer.render(t, s, W, H, row_begin=40, row_end=364), first[40:364]), f"row range, frame {k}"
        assert L.rm_debug_last_split() > 0
        # a frame whose pixels can spawn secondary rays is not eligible
        prim2 = all_primitives_scene(W, H)
        for c in range(3):
            prim2[1][3].cReflective[c] = 0.4
        t2 = tables_of((prim2[0], prim2[1], prim2[2], lights, 4, prim2[5]))
        s2 = abi.default_settings(maxSteps=96, enableSoftShadow=soft, enableAmbientOcclusion=ao, enableReflection=1)
        ref2 = renderer.render(t2, s2, W, H).clone()
        for k in range(6):
            assert _ieq(renderer.render(t2, s2, W, H), ref2)
        assert L.rm_debug_last_split() == 0
        # every tile split — 200 frames of it: the tiles' workgroups hand their results over through memory inside one launch
        # (release / acquire around a per-tile arrival counter), and whichever of them arrives last finishes the tile
        assert L.rm_debug_set_light_split(1) == 0
        for k in range(200):
            assert _ieq(renderer.render(t, s, W, H), first), f"all tiles split, frame {k}"
        assert L.rm_debug_last_split() == tiles
        assert L.rm_debug_set_light_split(0) == 0
        assert _ieq(renderer.render(t, s, W, H), first) and L.rm_debug_last_split() == 0
        # the default: the launcher measures (two plain frames, two split, then the better) — whatever it decides, the same frame
        assert L.rm_debug_set_light_split(-1) == 0
        seen = set()
        for k in range(20):
            assert _ieq(renderer.render(t, s, W, H), first), f"measured mode, frame {k}"
            seen.add(L.rm_debug_last_split())
        assert seen <= {0, tiles // 256} and tiles // 256 in seen  # its two split frames ran
    finally:
        L.rm_debug_set_tile_shape(-1)
        L.rm_debug_set_light_split(-1)


def test_tile_shape_tuner_never_changes_a_pixel(renderer):
    """The launcher measures, per stream and picture, whether 8×8 or 4 wide × 16 tall pixel tiles are faster (frames 0-7 of a
    picture alternate the two shapes in pairs, then the choice sticks; rm_kernels.hip "tile shape").  Every frame of such a
    sequence — and the frame with either shape forced, whole, as a row range and as a shard — is the same frame, the oracle's."""
    import torch
    from raymarcher_amd import lib
    L = lib()
    W, H = 640, 400  # 4000 8×8 tiles, 4000 4×16 tiles
    scene = all_primitives_scene(W, H)
    t = tables_of(scene)
    s = abi.default_settings(maxSteps=96, enableSoftShadow=1)
    try:
        assert L.rm_debug_set_tile_shape(3) == 0
        ref = renderer.render(t, s, W, H).clone()
        assert_bit_equal(ref.cpu().numpy(), h.oracle_render(scene, s, W, H), "8×8 tiles vs oracle")
        assert L.rm_debug_set_tile_shape(2) == 0
        for _ in range(2):
            assert _ieq(renderer.render(t, s, W, H), ref)
        assert _ieq(renderer.render(t, s, W, H, row_begin=33, row_end=377), ref[33:377])
        mine = renderer.render_tiles(t, s, W, H, 8, 1, 3)
        rows = [L.rm_shard_row_to_frame(H, 8, 1, 3, i) for i in range(mine.shape[0])]
        assert _ieq(mine, ref[torch.tensor(rows, device=ref.device)])
        assert L.rm_debug_set_tile_shape(0) == 0
        for k in range(14):  # the tuner's eight frames, its decision, and frames after it
            assert _ieq(renderer.render(t, s, W, H), ref), f"frame {k} of the tuned sequence differs"
        moved = (h.make_camera((0.5, 2.0, 6.5), (-0.1, -0.3, -1), (0, 1, 0), 45.0, W, H),) + scene[1:]
        want = h.oracle_render(moved, s, W, H)
        for k in range(3):  # a new picture restarts the measurement
            assert_bit_equal(renderer.render(tables_of(moved), s, W, H).cpu().numpy(), want, f"moved camera, frame {k}")
        assert L.rm_debug_set_tile_shape(5) == abi.RM_ERR_INVALID_ARGUMENT
    finally:
        L.rm_debug_set_tile_shape(-1)


# ---------------------------------------------------------------- the other BASELINE.json configurations, at their full sizes
SCENES = os.path.join(os.path.dirname(__file__), "golden", "scenes")


def _scene_tuple(t):
    return t.camera, t.objects, t.num_objects, t.lights, t.num_lights, t.globals_


def _ieq(a, b):
    import torch
    return bool((a.view(dtype=torch.int32) == b.view(dtype=torch.int32)).all())


def test_config1_unit_sphere_256(renderer):
    """configs[0]: scenefiles/simple/unit_sphere.json through the product's loader and PNG reader, 256×256, 64 steps, Phong
    only — the whole frame against the oracle, the floor textured with the reference's own texture_store/blackmarble.png
    (tests/golden/scenes/texture_store/, input data)."""
    from raymarcher_amd import Scene
    W = H = 256
    t = Scene(path=os.path.join(SCENES, "simple", "unit_sphere.json")).tables(W, H)
    assert t.num_objects == 2 and t.num_lights == 3 and sum(t.objects[i].texLoc == 0 for i in range(2)) == 1
    assert len(t.textures) == 1 and t.textures[0].shape == (1320, 1990, 4)  # blackmarble.png, decoded by rm_image_load
    s = abi.default_settings(maxSteps=64)
    ref = h.oracle_render(_scene_tuple(t), s, W, H, textures=t.textures)
    assert_bit_equal(renderer.render(t, s, W, H).cpu().numpy(), ref, "unit_sphere 256²")
    assert ref[..., :3].std() > 0.05 and np.isfinite(ref).all()  # the floor fills the view behind the sphere


def test_config2_lighting_1080p_softshadow_ao(renderer):
    """configs[1]: scenefiles/lighting/directional_light_2.json (5 primitives, 3 directional lights), 1920×1080, soft
    shadows + AO: the WHOLE frame against the oracle, a row-range render against the same rows of the full frame."""
    from raymarcher_amd import Scene
    W, H = 1920, 1080
    t = Scene(path=os.path.join(SCENES, "lighting", "directional_light_2.json")).tables(W, H)
    assert t.num_objects == 5 and t.num_lights == 3
    s = abi.default_settings(enableSoftShadow=1, enableAmbientOcclusion=1)
    full = renderer.render(t, s, W, H)
    ref = h.oracle_render(_scene_tuple(t), s, W, H, threads=16)
    assert_bit_equal(full.cpu().numpy(), ref, "the whole 1080p directional_light_2 frame")
    assert _ieq(renderer.render(t, s, W, H, 411, 623), full[411:623])
    assert 0.2 < float((full[..., :3] != 1.0).any(-1).float().mean()) < 0.95


def test_config4_terrain_cloud_4k_row_tiles(renderer):
    """configs[3]: scenefiles/simple/volumetric.json + TERRAIN | CLOUD | SKY_BACKGROUND at 3840×2160, as the 8 row-tile
    shards the 8-GPU job renders: every shard, gathered and de-interleaved, equals the single-launch frame; sampled
    rows equal the oracle."""
    import torch
    from raymarcher_amd import Scene, lib
    W, H, T, N = 3840, 2160, 8, 8
    t = Scene(path=os.path.join(SCENES, "simple", "volumetric.json")).tables(W, H, far=2000.0)
    s = abi.default_settings(features=ENV_ALL)
    # (i) the scenefile exactly as it is: its camera (0,500,5) sits below the terrain surface and looks straight down, so
    # every ray starts inside the height field — a nearly black frame, but the reference's own; rows against the oracle
    asis = renderer.render(t, s, W, H)
    for r0 in (7, 1080, 2100):
        ref = h.oracle_render(_scene_tuple(t), s, W, H, r0, r0 + 8, threads=16)
        assert_bit_equal(asis[r0:r0 + 8].cpu().numpy(), ref, f"4K volumetric.json as is, rows {r0}..{r0 + 8}")
    # (ii) the view the layers are made for — the reference's user flies the camera: same position, looking at the
    # horizon, as in env_scene() (DESIGN.md §6 lists both)
    t.camera = env_scene(W, H)[0]
    full = renderer.render(t, s, W, H)
    assert 0.1 < float(torch.nan_to_num(full[..., :3]).mean()) < 1.5
    slot = lib().rm_shard_rows(H, T, 0, N)
    gathered = torch.zeros((N * slot, W, 4), dtype=torch.float32, device=full.device)
    for k in range(N):
        mine = renderer.render_tiles(t, s, W, H, T, k, N)
        assert mine.shape[0] == lib().rm_shard_rows(H, T, k, N)
        gathered[k * slot:k * slot + mine.shape[0]] = mine
    assert _ieq(renderer.deinterleave(gathered, W, H, T, N, slot), full)
    for r0 in (40, 1400):
        ref = h.oracle_render(_scene_tuple(t), s, W, H, r0, r0 + 16, threads=16)
        assert_bit_equal(full[r0:r0 + 16].cpu().numpy(), ref, f"4K env rows {r0}..{r0 + 16}")
    # a cloud sample exactly at y = 900 has gradient sign(0) = 0 and the reference normalises it (frag:1996): NaN there too
    assert float((~torch.isfinite(full).all(-1)).float().mean()) < 1e-4


def test_config5_menger_8k_reflection(renderer):
    """configs[4]: scenefiles/simple/unit_mengersponge.json through the product's loader, 5 levels, reflection with 2
    bounces, 7680×4320 (530 MB of float4): one of the 8 shards against the full frame, sampled rows against the oracle."""
    import torch
    from raymarcher_amd import Scene, lib
    W, H, T, N = 7680, 4320, 8, 8
    t = Scene(path=os.path.join(SCENES, "simple", "unit_mengersponge.json")).tables(W, H)  # the reference's own scenefile
    assert t.num_objects == 1 and t.num_lights == 3 and t.objects[0].type == abi.RM_MENGERSPONGE
    s = abi.default_settings(mengerLevels=5, numReflection=2, enableReflection=1)
    full = renderer.render(t, s, W, H)
    k = 5
    assert lib().rm_debug_last_path() == 5
    mine = renderer.render_tiles(t, s, W, H, T, k, N)
    # a row-TILE shard of 4.1 M pixels takes the wavefront pipeline too (multi-GPU hosts keep frames in flight: threshold 2^21)
    assert lib().rm_debug_last_path() == 5
    rows = [lib().rm_shard_row_to_frame(H, T, k, N, i) for i in range(mine.shape[0])]
    assert _ieq(mine, full[torch.tensor(rows, device=full.device)])
    band = renderer.render(t, s, W, H, row_begin=1000, row_end=1000 + mine.shape[0])  # the same pixel count as a plain row range: 2^22 applies
    assert lib().rm_debug_last_path() == 1 and _ieq(band, full[1000:1000 + mine.shape[0]])
    for r0 in (2160, 3000):
        ref = h.oracle_render(_scene_tuple(t), s, W, H, r0, r0 + 8, threads=16)
        assert_bit_equal(full[r0:r0 + 8].cpu().numpy(), ref, f"8K rows {r0}..{r0 + 8}")
    hit = float((full[..., :3] != 1.0).any(-1).float().mean())
    assert 0.1 < hit < 0.9


RCCL_WORKER = r'''
import os, sys
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
from raymarcher_amd import Renderer, abi, scenes
from raymarcher_amd.dist import FramePipeline, ShardPlan
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
r = Renderer(0)
W, H, T = 320, 180, 8
plan = ShardPlan(H, T, 1)
frames = []
pipe = FramePipeline(plan, 0, (W, 4), torch.float32, r.device, depth=3, multi_stream=True,   # as bench.py runs it for N > 1
                     finish=lambda g: frames.append(r.deinterleave(g, W, H, T, 1, plan.slot_rows).clone()))
want = []
for k in range(7):
    t = scenes.mandelbulb(W, H)
    t.globals_.power = 8.0 - 0.5 * k                # a different image per frame
    s = abi.default_settings(fractalIters=12)
    want.append(r.render(t, s, W, H).clone())
    pipe.submit(lambda slot, t=t, s=s: r.render_tiles(t, s, W, H, T, 0, 1, out=slot[:plan.rows(0)]))
pipe.drain()
torch.cuda.synchronize()
ok = len(frames) == 7 and all(bool((a.view(dtype=torch.int32) == b.view(dtype=torch.int32)).all()) for a, b in zip(frames, want))
ok = ok and not bool((want[0] == want[1]).all())
dist.destroy_process_group()
sys.exit(0 if ok else 3)
'''


def test_frame_pipeline_over_rccl_single_rank(renderer, tmp_path):
    """The N > 1 machinery of bench.py (RCCL process group, asynchronous gather into slot views, stream-ordered join,
    rm_deinterleave) with one rank on this GPU, in a child process: every pipelined frame equals the direct render."""
    import subprocess
    import sys
    script = tmp_path / "rccl_worker.py"
    script.write_text(RCCL_WORKER.format(root=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29700 + os.getpid() % 200), HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]


# ---------------------------------------------------------------- randomised scenes
def _random_case(rng, W, H):
    """A random scene + settings + resources drawn from everything the ABI accepts."""
    f = rng.uniform
    n_obj = int(rng.integers(1, 7))
    types = [abi.RM_CUBE, abi.RM_CONE, abi.RM_CYLINDER, abi.RM_SPHERE, abi.RM_OCTAHEDRON, abi.RM_TORUS, abi.RM_CAPSULE,
             abi.RM_DEATHSTAR, abi.RM_RECTANGLE, abi.RM_MANDELBULB, abi.RM_MENGERSPONGE, abi.RM_SIERPINSKI]
    texs = synthetic_textures()
    objs = []
    for _ in range(n_obj):
        ty = int(rng.choice(types))
        sc = float(f(0.6, 1.8))
        sx, sy, sz = (sc * float(f(0.8, 1.25)) for _ in range(3))
        M = h.translate(f(-2.2, 2.2), f(-1.0, 1.2), f(-2.5, 1.0)) @ rot_x(f(-0.6, 0.6)) @ h.scale(sx, sy, sz)
        o = h.make_object(ty, model=M, scale_factor=min(sx, sy, sz), ambient=tuple(f(0, .3, 3)), diffuse=tuple(f(.2, 1, 3)),
                          specular=tuple(f(0, 1, 3)), shininess=float(rng.choice([0, 1, 7.5, 25, 100])),
                          reflective=tuple(f(0, .8, 3)) if f() < 0.4 else (0, 0, 0),
                          transparent=tuple(f(0, .8, 3)) if f() < 0.3 else (0, 0, 0), ior=float(f(1.05, 1.6)))
        if ty in (abi.RM_CUBE, abi.RM_CONE, abi.RM_CYLINDER, abi.RM_SPHERE) and f() < 0.4:
            o.texLoc, o.repeatU, o.repeatV, o.blend = int(rng.integers(0, 2)), float(f(0.5, 4)), float(f(0.5, 4)), float(f(0, 1))
        objs.append(o)
    lights = []
    for _ in range(int(rng.integers(1, 4))):
        kind = int(rng.integers(0, 3))
        col = tuple(f(.3, 1.6, 3))
        if kind == abi.RM_LIGHT_DIRECTIONAL:
            lights.append(h.make_light(kind, col, direction=(f(-1, 1), f(-1, -0.2), f(-1, 1))))
        elif kind == abi.RM_LIGHT_POINT:
            lights.append(h.make_light(kind, col, pos=(f(-4, 4), f(1, 5), f(-1, 5)), func=(f(.5, 1), f(0, .1), f(0, .02))))
        else:
            lights.append(h.make_light(kind, col, direction=(f(-.3, .3), -1, f(-.6, 0)), pos=(f(-2, 2), f(3, 5), f(0, 3)),
                                       func=(f(.5, 1), f(0, .1), 0), angle=float(f(.4, .9)), penumbra=float(f(.05, .3))))
    res = {}
    if any(o.texLoc >= 0 for o in objs):
        res["textures"] = texs
    if f() < 0.3:  # an area light with its emissive rectangle
        ctm = h.translate(f(-1, 1), f(2, 3), f(-2, 0)) @ rot_x(f(0.8, 1.5)) @ h.scale(f(1, 3), f(1, 2), 1.0)
        rect = h.make_object(abi.RM_RECTANGLE, model=ctm, scale_factor=1.0)
        rect.isEmissive, rect.lightIdx = 1, len(lights)
        area = h.make_light(abi.RM_LIGHT_AREA, tuple(f(.5, 1.2, 3)))
        area.intensity, area.twoSided = float(rng.choice([0.0, 0.7])), int(rng.integers(0, 2))
        for k, c in enumerate([(-0.5, 0.5, 0), (0.5, 0.5, 0), (0.5, -0.5, 0), (-0.5, -0.5, 0)]):
            w = ctm @ np.array([*c, 1.0])
            for j in range(3):
                area.points[k][j], rect.color[j] = float(np.float32(w[j])), area.color[j]
        objs.append(rect)
        lights.append(area)
        t1, t2 = synthetic_ltc()
        res["ltc1"], res["ltc2"] = h.oracle_ltc_quantise(t1), h.oracle_ltc_quantise(t2)
    feats = int(rng.choice([abi.RM_FEAT_WHITE_BACKGROUND, abi.RM_FEAT_DARK_BACKGROUND, 0, abi.RM_FEAT_SKY_BACKGROUND,
                            abi.RM_FEAT_NIGHTSKY_BACKGROUND]))
    if f() < 0.6:
        feats |= abi.RM_FEAT_PERLIN_BUMP
    if f() < 0.15:
        feats |= abi.RM_FEAT_SEA
    if f() < 0.3:
        feats |= abi.RM_FEAT_BULB_POWER8_ALGEBRAIC
    if feats & (abi.RM_FEAT_NIGHTSKY_BACKGROUND | abi.RM_FEAT_SEA):
        res["noise"] = synthetic_noise()
    sky = f() < 0.25
    if sky:
        res["skybox"] = synthetic_skybox(12)
    s = abi.default_settings(features=feats, enableSoftShadow=int(f() < 0.3), enableAmbientOcclusion=int(f() < 0.4),
                             enableReflection=int(f() < 0.5), enableRefraction=int(f() < 0.4), enableSkyBox=int(sky),
                             maxSteps=int(rng.choice([64, 128, 256])), fractalIters=int(rng.choice([6, 12, 20])),
                             mengerLevels=int(rng.choice([3, 4, 5])), numReflection=int(rng.choice([1, 2, 3])))
    g = h.make_globals(ka=f(.2, .8), kd=f(.3, 1), ks=f(.2, 1), kt=f(.2, 1), power=float(rng.choice([8.0, 8.0, 6.0, 5.5, 2.0])),
                       julia=(f(-.5, .5), f(-.5, .5)) if f() < 0.2 else (0, 0), itime=float(f(0, 9)))
    cam = h.make_camera((f(-1, 1), f(0.5, 2.5), f(4.5, 6.5)), (f(-.15, .15), f(-.45, -.05), -1), (0, 1, 0), float(f(35, 60)), W, H)
    scene = (cam, (abi.RmObject * len(objs))(*objs), len(objs), (abi.RmLight * len(lights))(*lights), len(lights), g)
    return scene, s, res


def test_random_scenes_bit_exact(renderer):
    """24 seeded random scenes over the whole ABI surface (every primitive and fractal type, all four light kinds, object
    textures, sky box, night sky, sea, every option and loop-bound knob): the GPU equals the oracle in every bit."""
    W, H = (int(v) for v in os.environ.get("RM_FUZZ_SIZE", "56x40").split("x"))  # 512x320: ordered tiles, see the wide test below
    rng = np.random.default_rng(int(os.environ.get("RM_FUZZ_SEED", "20261003")))
    kinds = set()
    for i in range(int(os.environ.get("RM_FUZZ_CASES", "24"))):  # a long soak: RM_FUZZ_CASES=400 RM_FUZZ_SEED=…
        scene, s, res = _random_case(rng, W, H)
        ref, ref_b = h.oracle_render(scene, s, W, H, bright=True, **res)
        t = tables_of(scene)
        for k, v in res.items():
            setattr(t, k, v)
        out, br = renderer.render(t, s, W, H, bright=True)
        assert_bit_equal(out.cpu().numpy(), ref, f"random scene {i}")
        assert_bit_equal(br.cpu().numpy(), ref_b, f"random scene {i} bright")
        if (W // 8) * (H // 8) >= 2048:
            for rep in range(3):
                assert _ieq(renderer.render(t, s, W, H), out), f"random scene {i}: repeat {rep + 1} differs"
        kinds |= {scene[1][k].type for k in range(scene[2])}
    assert len(kinds) >= 10


def _random_primitive_case(rng, W, H):
    """A random all-primitive table (the class of the table walk's pass-over test AND of the march loops' single-object fast
    path), one to eight objects, sometimes over a floor slab, two to ten lights of the three plain kinds, every shading option."""
    f = rng.uniform
    types = [abi.RM_CUBE, abi.RM_CONE, abi.RM_CYLINDER, abi.RM_SPHERE, abi.RM_OCTAHEDRON, abi.RM_TORUS, abi.RM_CAPSULE,
             abi.RM_DEATHSTAR, abi.RM_RECTANGLE]
    objs = []
    for _ in range(int(rng.integers(1, 9))):
        ty = int(rng.choice(types))
        sc = float(f(0.6, 1.8))
        sx, sy, sz = (sc * float(f(0.8, 1.25)) for _ in range(3))
        M = h.translate(f(-2.2, 2.2), f(-1.0, 1.2), f(-2.5, 1.0)) @ rot_x(f(-0.6, 0.6)) @ h.scale(sx, sy, sz)
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


def test_random_primitive_scenes_bit_exact(renderer):
    """Seeded random all-primitive scenes (the general random test draws a fractal into most tables, which switches the
    single-object fast path off): the GPU equals the oracle in every bit, through both schedules."""
    from raymarcher_amd import lib
    W, H = 56, 40
    rng = np.random.default_rng(int(os.environ.get("RM_FUZZ_SEED", "20261007")))
    for i in range(int(os.environ.get("RM_FUZZ_CASES", "32"))):
        scene, s = _random_primitive_case(rng, W, H)
        ref, ref_b = h.oracle_render(scene, s, W, H, bright=True)
        out, br = renderer.render(tables_of(scene), s, W, H, bright=True)
        assert_bit_equal(out.cpu().numpy(), ref, f"random primitive scene {i}")
        assert_bit_equal(br.cpu().numpy(), ref_b, f"random primitive scene {i} bright")
        if not s.enableRefraction and i % 4 == 0:
            try:
                lib().rm_set_kernel_path(5)
                wf = renderer.render(tables_of(scene), s, W, H)
                assert lib().rm_debug_last_path() == 5 and _ieq(wf, out)
            finally:
                lib().rm_set_kernel_path(0)


def _random_tablewalk_case(rng, W, H):
    """A WIDE random all-primitive scene (helpers.random_tablewalk_objects: arbitrary-axis rotations, shear, anisotropy 0.2–5,
    scaleFactors that are not the smallest scale, nested and coincident objects, tables of up to 30), sometimes over a floor slab
    or inside an enclosing box, one to ten lights of the three plain kinds, every shading option, cameras outside, inside an
    object, or on an object's surface."""
    f = rng.uniform
    objs = h.random_tablewalk_objects(rng, max_objects=28)
    if f() < 0.4:  # a floor: long grazing shadow rays
        objs.append(h.make_object(abi.RM_CUBE, model=h.translate(0, -1.8, -1) @ h.scale(11, 0.2, 11), scale_factor=0.2,
                                  diffuse=(.7, .7, .7), ambient=(.1, .1, .1), reflective=(.3, .3, .3) if f() < 0.3 else (0, 0, 0)))
    if f() < 0.15:  # everything (camera too) inside one big cube: every ray hits, negative distances never occur but no ray leaves
        objs.append(h.make_object(abi.RM_CUBE, model=h.scale(24, 24, 24), scale_factor=24, diffuse=(.4, .5, .4), ambient=(.1, .1, .1)))
    lights = []
    for _ in range(int(rng.choice([1, 2, 2, 3, 3, 3, 4, 5, 7, 10]))):
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
                             maxSteps=int(rng.choice([16, 64, 256, 256])), numReflection=int(rng.choice([1, 2, 3])))
    g = h.make_globals(ka=f(.2, .8), kd=f(.3, 1), ks=f(.2, 1), kt=f(.2, 1))
    where = f()
    if where < 0.2:  # the camera inside an object (its near plane, where rays start, may still be outside a small one)
        o = objs[int(rng.integers(0, len(objs)))]
        M = np.linalg.inv(np.array(list(o.invModel), dtype=np.float64).reshape(4, 4).T)
        pos = tuple((M @ np.array([*f(-0.15, 0.15, 3), 1.0]))[:3])
        look = tuple(f(-1, 1, 3) + np.array([0, 0, -0.3]))
    elif where < 0.3:  # on (about) the bounding ball of an object, looking along it
        o = objs[int(rng.integers(0, len(objs)))]
        M = np.linalg.inv(np.array(list(o.invModel), dtype=np.float64).reshape(4, 4).T)
        d = rng.normal(size=3)
        pos = tuple((M @ np.array([*(d / np.linalg.norm(d) * 0.6), 1.0]))[:3])
        look = tuple(np.cross(d, rng.normal(size=3)))
    else:
        pos, look = (f(-1, 1), f(0.5, 2.5), f(4.5, 6.5)), (f(-.15, .15), f(-.45, -.05), -1)
    if np.linalg.norm(look) < 1e-3 or abs(np.dot(look, (0, 1, 0))) > 0.98 * np.linalg.norm(look):
        look = (0.1, -0.2, -1)
    cam = h.make_camera(pos, look, (0, 1, 0), float(f(35, 70)), W, H)
    return (cam, (abi.RmObject * len(objs))(*objs), len(objs), (abi.RmLight * len(lights))(*lights), len(lights), g), s


def test_random_tablewalk_scenes_bit_exact(renderer):
    """The exactness evidence of the table walk's shortcuts — pass-over test, runner-up tracking / single-object fast path, ball ∩
    box march bounds (rm_device.hip.h sdSceneImpl / march; their slack constants 1.00003, 1e-4, nextMinBound) — on scenes drawn
    WIDE: 128 seeded cases by default; RM_FUZZ_CASES=1000 with three RM_FUZZ_SEEDs is the soak
    (scripts/gpu_fuzz_soak.sh, summary in profiles/).  Every bit of fragColor and BrightColor equals the oracle's; every
    fourth refraction-free scene also runs through the wavefront pipeline's instantiations of the same tests."""
    from raymarcher_amd import lib
    # RM_FUZZ_SIZE=512x320 (>= 2048 tiles): the launcher then also orders the tiles — by geometry for the new picture, by measured
    # costs for its repeats — and its tuner alternates the tile shape: every scene is rendered four times, all the same frame
    W, H = (int(v) for v in os.environ.get("RM_FUZZ_SIZE", "56x40").split("x"))
    seed = int(os.environ.get("RM_FUZZ_SEED", "20261012"))
    rng = np.random.default_rng(seed)
    cases = int(os.environ.get("RM_FUZZ_CASES", "128"))
    stats = {"objects": 0, "max_objects": 0, "wavefront": 0}
    for i in range(cases):
        scene, s = _random_tablewalk_case(rng, W, H)
        ref, ref_b = h.oracle_render(scene, s, W, H, bright=True)
        out, br = renderer.render(tables_of(scene), s, W, H, bright=True)
        assert_bit_equal(out.cpu().numpy(), ref, f"seed {seed} wide table-walk scene {i} ({scene[2]} objects)")
        assert_bit_equal(br.cpu().numpy(), ref_b, f"seed {seed} wide table-walk scene {i} bright")
        if (W // 8) * (H // 8) >= 2048:
            for rep in range(3):
                assert _ieq(renderer.render(tables_of(scene), s, W, H), out), f"seed {seed} scene {i}: repeat {rep + 1} differs"
            # … and with the tile shape pinned the picture settles by its fifth frame: the sixth has its heaviest eighth of the tiles
            # rendered one light per workgroup where the scene is eligible (no secondary rays, two or more lights)
            try:
                lib().rm_debug_set_tile_shape(3)
                lib().rm_debug_set_light_split(8)
                for rep in range(6):
                    assert _ieq(renderer.render(tables_of(scene), s, W, H), out), f"seed {seed} scene {i}: pinned repeat {rep + 1} differs"
                stats["split"] = stats.get("split", 0) + (1 if lib().rm_debug_last_split() > 0 else 0)
            finally:
                lib().rm_debug_set_tile_shape(-1)
                lib().rm_debug_set_light_split(-1)
        stats["objects"] += scene[2]
        stats["max_objects"] = max(stats["max_objects"], scene[2])
        if not s.enableRefraction and i % 4 == 0:
            try:
                lib().rm_set_kernel_path(5)
                wf = renderer.render(tables_of(scene), s, W, H)
                assert lib().rm_debug_last_path() == 5 and _ieq(wf, out), f"seed {seed} scene {i}: wavefront differs"
                stats["wavefront"] += 1
            finally:
                lib().rm_set_kernel_path(0)
    print(f"FUZZ_SUMMARY seed={seed} cases={cases} mismatched_words=0 mean_objects={stats['objects'] / max(cases, 1):.1f} "
          f"max_objects={stats['max_objects']} wavefront_cases={stats['wavefront']} light_split_cases={stats.get('split', 0)}")
    assert stats["max_objects"] >= 20 or cases < 32


def _random_bulb_case(rng, W, H):
    """A random scene of the single-Mandelbulb class (its own kernel instantiation: bounding-ball culls of two radii,
    v_min orbit trap, per-lane shadow-ray queue): model transform incl. anisotropic scales and tiny objects, Julia seeds
    inside and outside the tight ball's bound, powers, 1–5 lights of any kind (all-directional sets take the queue), every
    option, camera anywhere around — also inside the ball."""
    f = rng.uniform
    sc = float(rng.choice([1.0, 1.0, 1.7, 0.4, 0.04, 0.008]))
    an = (1.0, 1.0, 1.0) if f() < 0.7 else tuple(f(0.6, 2.5, 3))
    M = h.translate(*(f(-0.4, 0.4, 3) * sc)) @ rot_x(f(-1.0, 1.0)) @ h.scale(sc * an[0], sc * an[1], sc * an[2])
    o = h.make_object(abi.RM_MANDELBULB, model=M, scale_factor=sc * min(an), ambient=tuple(f(0, .4, 3)), diffuse=tuple(f(.2, 1, 3)),
                      specular=tuple(f(0, 1, 3)), shininess=float(rng.choice([0, 7.5, 25, 100])),
                      reflective=tuple(f(0, .8, 3)) if f() < 0.3 else (0, 0, 0),
                      transparent=tuple(f(0, .8, 3)) if f() < 0.2 else (0, 0, 0), ior=float(f(1.05, 1.6)))
    lights = []
    all_dir = f() < 0.6
    for _ in range(int(rng.integers(1, 6))):
        kind = abi.RM_LIGHT_DIRECTIONAL if all_dir else int(rng.integers(0, 3))
        col = tuple(f(.3, 1.6, 3))
        if kind == abi.RM_LIGHT_DIRECTIONAL:
            d = f(-1, 1, 3)
            lights.append(h.make_light(kind, col, direction=tuple(d if np.abs(d).max() > 0.1 else (0, -1, 0))))
        elif kind == abi.RM_LIGHT_POINT:
            lights.append(h.make_light(kind, col, pos=tuple(f(-4, 4, 3) * max(sc, 0.2)), func=(f(.5, 1), f(0, .1), f(0, .02))))
        else:
            lights.append(h.make_light(kind, col, direction=(f(-.3, .3), -1, f(-.6, 0)), pos=(f(-2, 2) * sc, f(3, 5) * sc, f(0, 3) * sc),
                                       func=(f(.5, 1), f(0, .1), 0), angle=float(f(.4, .9)), penumbra=float(f(.05, .3))))
    feats = int(rng.choice([abi.RM_FEAT_WHITE_BACKGROUND, abi.RM_FEAT_DARK_BACKGROUND, 0])) | (abi.RM_FEAT_PERLIN_BUMP if f() < 0.6 else 0)
    if f() < 0.2:
        feats |= abi.RM_FEAT_BULB_POWER8_ALGEBRAIC
    s = abi.default_settings(features=feats, enableSoftShadow=int(f() < 0.25), enableAmbientOcclusion=int(f() < 0.3),
                             enableReflection=int(f() < 0.4), enableRefraction=int(f() < 0.3),
                             maxSteps=int(rng.choice([1, 17, 64, 256])), fractalIters=int(rng.choice([1, 4, 12, 20])),
                             numReflection=int(rng.choice([1, 2])))
    julia = (0, 0) if f() < 0.6 else (tuple(f(-.6, .6, 2)) if f() < 0.6 else tuple(f(-1.6, 1.6, 2)))
    g = h.make_globals(ka=f(.2, .8), kd=f(.3, 1), ks=f(.2, 1), kt=f(.2, 1), power=float(rng.choice([8.0, 8.0, 8.0, 6.0, 3.5])), julia=julia)
    dist = float(rng.choice([4.5, 3.0, 1.6, 0.8])) * sc * max(an)
    dirv = f(-1, 1, 3)
    dirv = dirv / (np.linalg.norm(dirv) + 1e-9)
    pos = tuple(dirv * dist)
    look = tuple(-dirv + f(-0.15, 0.15, 3))
    cam = h.make_camera(pos, look, (0.1, 1, 0.05), float(f(25, 70)), W, H, near=0.02 * dist, far=float(rng.choice([100.0, 100.0, 6.0 * dist])))
    return (cam, (abi.RmObject * 1)(o), 1, (abi.RmLight * len(lights))(*lights), len(lights), g), s


def test_random_bulb_scenes_bit_exact(renderer):
    """Seeded random scenes of the single-Mandelbulb class — the benchmark's kernel with all its bit-identical shortcuts —
    against the oracle in every bit, plus the BrightColor plane; counters of the executed-work build stay below the
    reference-work ones."""
    W, H = 48, 40
    rng = np.random.default_rng(int(os.environ.get("RM_FUZZ_SEED", "20261004")))
    hits = 0
    for i in range(int(os.environ.get("RM_FUZZ_CASES", "24"))):
        scene, s = _random_bulb_case(rng, W, H)
        ref, ref_b = h.oracle_render(scene, s, W, H, bright=True)
        out, br = renderer.render(tables_of(scene), s, W, H, bright=True)
        assert_bit_equal(out.cpu().numpy(), ref, f"random bulb scene {i}")
        assert_bit_equal(br.cpu().numpy(), ref_b, f"random bulb scene {i} bright")
        hits += int((ref[..., 3] > 0).any() and (ref[..., :3] != ref[0, 0, :3]).any())
        if i % 8 == 0:
            _, c1 = renderer.render_counted(tables_of(scene), s, W, H, abi.RM_COUNT_REFERENCE)
            _, c2 = renderer.render_counted(tables_of(scene), s, W, H, abi.RM_COUNT_EXECUTED)
            assert c2.sceneEvals <= c1.sceneEvals and c2.bulbIters <= c1.bulbIters and c2.hitPixels == c1.hitPixels
    assert hits >= 0.5 * int(os.environ.get("RM_FUZZ_CASES", "24"))  # most random views see the bulb


def test_bounding_ball_cull_edge_cases(renderer):
    """The launcher's bounding ball (scene_cull_ball / bulbCullEnd) must never change a bit: strongly non-uniform and
    rotated models, tiny and huge objects, objects far from the origin, the camera inside the ball, a far plane shorter
    than the ball, secondary rays, scaled / translated / Julia Mandelbulbs, hard and soft shadows."""
    W, H = 72, 48
    rz = np.eye(4)
    a = 0.7
    rz[0, 0], rz[0, 1], rz[1, 0], rz[1, 1] = np.cos(a), -np.sin(a), np.sin(a), np.cos(a)
    lights = (abi.RmLight * 2)(h.make_light(abi.RM_LIGHT_DIRECTIONAL, (1, 1, 1), (-0.3, -1, -0.5)),
                               h.make_light(abi.RM_LIGHT_POINT, (.9, .8, .7), pos=(3, 4, 4), func=(0.7, 0.05, 0)))
    refl = dict(ambient=(.1, .1, .1), diffuse=(.7, .6, .5), specular=(.8, .8, .8), shininess=20, reflective=(.5, .5, .5))
    cases = []
    # 1: slab + needle, strongly non-uniform, rotated
    objs = (abi.RmObject * 3)(
        h.make_object(abi.RM_CUBE, model=h.translate(0, -1, 0) @ rz @ h.scale(6, 0.15, 3), scale_factor=0.15, **refl),
        h.make_object(abi.RM_CYLINDER, model=h.translate(1, 0.5, -1) @ rot_x(0.9) @ h.scale(0.2, 4, 0.2), scale_factor=0.2, **refl),
        h.make_object(abi.RM_TORUS, model=h.translate(-1.5, 0.4, 0.5) @ rz @ h.scale(2, 2, 0.7), scale_factor=0.7, **refl))
    cases.append((h.make_camera((0, 1.5, 6), (0, -0.25, -1), (0, 1, 0), 45.0, W, H), objs, 3, h.make_globals()))
    # 2: tiny and huge objects, far from the origin; camera inside the bounding ball
    objs = (abi.RmObject * 3)(
        h.make_object(abi.RM_SPHERE, model=h.translate(50, 2, -40) @ h.scale(30, 30, 30), scale_factor=30, **refl),
        h.make_object(abi.RM_OCTAHEDRON, model=h.translate(0.3, 0, -2) @ h.scale(0.05, 0.05, 0.05), scale_factor=0.05, **refl),
        h.make_object(abi.RM_DEATHSTAR, model=h.translate(-2, 0.5, -4) @ h.scale(2, 2, 2), scale_factor=2, **refl))
    cases.append((h.make_camera((0, 0.5, 2), (0.2, 0, -1), (0, 1, 0), 60.0, W, H), objs, 3, h.make_globals()))
    # 3: far plane inside the ball
    cases.append((h.make_camera((0, 1.5, 6), (0, -0.25, -1), (0, 1, 0), 45.0, W, H, far=5.5), cases[0][1], 3, h.make_globals()))
    # 4-9: Mandelbulbs: scaled + translated, Julia (inside and outside the tight ball's seed bound), tiny (scaleFactor
    # below the cull's threshold), small (between the thresholds of the tight and the wide ball), strongly anisotropic
    # (world t and object-space distance differ by the scale ratio: the soft-shadow bound must account for it)
    for model, sf, glob in ((h.translate(0.4, -0.2, 0.3) @ rz @ h.scale(1.6, 1.6, 1.6), 1.6, h.make_globals()),
                            (np.eye(4), 1.0, h.make_globals(julia=(0.4, -0.3))),
                            (h.scale(0.005, 0.005, 0.005), 0.005, h.make_globals()),
                            (np.eye(4), 1.0, h.make_globals(julia=(1.2, -0.9))),
                            (h.scale(0.03, 0.03, 0.03), 0.03, h.make_globals()),
                            (rz @ h.scale(3.0, 1.0, 1.0), 1.0, h.make_globals())):
        objs = (abi.RmObject * 1)(h.make_object(abi.RM_MANDELBULB, model=model, scale_factor=sf, ambient=(.3, .3, .3), specular=(1, 1, 1),
                                                shininess=100, reflective=(.4, .4, .4)))
        pos = (0, 0, 4.5) if sf > 0.05 else (0, 0, 4.0 * sf)
        cases.append((h.make_camera(pos, (0, 0, -1), (0, 1, 0), 30.0, W, H, near=0.001 if sf < 0.05 else 0.1), objs, 1, glob))
    for k, (cam, objs, no, g) in enumerate(cases):
        scene = (cam, objs, no, lights, 2, g)
        for over in ({"enableReflection": 1, "numReflection": 2}, {"enableSoftShadow": 1, "enableAmbientOcclusion": 1, "fractalIters": 8}):
            s = abi.default_settings(features=abi.RM_FEAT_WHITE_BACKGROUND, **over)
            ref = h.oracle_render(scene, s, W, H)
            assert_bit_equal(renderer.render(tables_of(scene), s, W, H).cpu().numpy(), ref, f"cull case {k} {over}")
            assert (ref[..., :3] != 1.0).any(-1).mean() > 0.02, f"case {k}: the objects must be in view"



def test_bounding_box_cull_edge_cases(renderer):
    """The launcher's bounding BOX on top of the ball (flat / elongated scenes; rm_debug_cull_bounds) must never change a bit:
    rays with exactly zero direction components (axis-aligned views and lights — the slab test divides by them), the camera on
    a face plane of the box, inside it, below the floor, a far plane inside the box, secondary rays, hard and soft shadows."""
    from raymarcher_amd import lib
    W, H = 72, 48
    mat = dict(ambient=(.1, .1, .1), diffuse=(.7, .6, .5), specular=(.8, .8, .8), shininess=20, reflective=(.5, .5, .5),
               transparent=(.3, .3, .3), ior=1.3)
    objs = (abi.RmObject * 5)(
        h.make_object(abi.RM_CUBE, model=h.translate(0, -1, 0) @ h.scale(14, 0.2, 6), scale_factor=0.2, **mat),
        h.make_object(abi.RM_SPHERE, model=h.translate(-4, -0.4, 0), **mat),
        h.make_object(abi.RM_CONE, model=h.translate(-1.5, -0.4, 0.5) @ rot_x(0.3), **mat),
        h.make_object(abi.RM_CYLINDER, model=h.translate(1.5, -0.4, -0.5), **mat),
        h.make_object(abi.RM_TORUS, model=h.translate(4.5, -0.3, 0.2) @ rot_x(1.2) @ h.scale(1.5, 1.5, 1.5), scale_factor=1.5, **mat))
    g = h.make_globals()
    b = np.zeros(13, dtype=np.float32)
    assert lib().rm_debug_cull_bounds(objs, 5, C.byref(g), b.ctypes.data_as(C.POINTER(C.c_float))) == 0
    assert b[0] == 1 and b[6] == 1, "the scene must have its box"
    lo, hi = b[7:10], b[10:13]
    # lights along the axes: direction components exactly zero
    lights = (abi.RmLight * 3)(h.make_light(abi.RM_LIGHT_DIRECTIONAL, (1, 1, 1), (0, -1, 0)),
                               h.make_light(abi.RM_LIGHT_DIRECTIONAL, (.6, .6, .9), (-1, 0, 0)),
                               h.make_light(abi.RM_LIGHT_POINT, (.9, .8, .7), pos=(0, 3, 4), func=(0.7, 0.05, 0)))
    cams = [h.make_camera((0, 0.3, 9), (0, 0, -1), (0, 1, 0), 40.0, W, H),            # along −z: the centre column has rd.x = 0
            h.make_camera((0, 8, 0.0), (0, -1, 0), (0, 0, -1), 50.0, W, H),           # straight down
            h.make_camera((float(hi[0]), 0.5, 2), (-1, -0.1, -0.3), (0, 1, 0), 50.0, W, H),   # on the box's +x face plane
            h.make_camera((0.5, float(hi[1]), 3), (0, -0.2, -1), (0, 1, 0), 50.0, W, H),      # on its top plane
            h.make_camera((0.2, 0.0, 1.5), (0.3, -0.2, -1), (0, 1, 0), 60.0, W, H),   # inside the box
            h.make_camera((0, -3, 4), (0, 0.6, -1), (0, 1, 0), 50.0, W, H),           # below the floor, looking up at it
            h.make_camera((0, 0.3, 9), (0, 0, -1), (0, 1, 0), 40.0, W, H, far=8.0)]   # far plane inside the box
    for k, cam in enumerate(cams):
        scene = (cam, objs, 5, lights, 3, g)
        for over in ({"enableReflection": 1, "enableRefraction": 1, "numReflection": 2}, {"enableSoftShadow": 1, "enableAmbientOcclusion": 1}):
            s = abi.default_settings(features=abi.RM_FEAT_WHITE_BACKGROUND, **over)
            ref, ref_b = h.oracle_render(scene, s, W, H, bright=True)
            out, br = renderer.render(tables_of(scene), s, W, H, bright=True)
            assert_bit_equal(out.cpu().numpy(), ref, f"box cull camera {k} {over}")
            assert_bit_equal(br.cpu().numpy(), ref_b, f"box cull camera {k} {over} bright")
    # the same through the wavefront pipeline (its refill computes the march ends)
    try:
        lib().rm_set_kernel_path(5)
        s = abi.default_settings(features=abi.RM_FEAT_WHITE_BACKGROUND, enableReflection=1, numReflection=2)
        for k in (0, 2, 4):
            scene = (cams[k], objs, 5, lights, 3, g)
            out = renderer.render(tables_of(scene), s, W, H)
            assert lib().rm_debug_last_path() == 5
            assert_bit_equal(out.cpu().numpy(), h.oracle_render(scene, s, W, H), f"box cull, wavefront, camera {k}")
    finally:
        lib().rm_set_kernel_path(0)



def test_table_walk_skip_edge_cases(renderer):
    """The table walk passes over objects that cannot lower a lane's minimum (sdScene<…, SKIP>, seeded by a Lipschitz bound from
    the previous step) and follows a single object while the runner-up stays above that bound (march()'s fast path): neither may
    ever change a bit — ties between identical objects (the lower index wins: different materials
    show it), the camera inside an object (negative distances), a full table of 30, a scaleFactor that is NOT the smallest
    scale (the distance values are then 2-Lipschitz and the march overshoots, as the reference's would), sheared model
    matrices, every shading option, both schedules."""
    from raymarcher_amd import lib
    W, H = 72, 48
    lights = (abi.RmLight * 3)(h.make_light(abi.RM_LIGHT_DIRECTIONAL, (1, 1, 1), (-0.3, -1, -0.5)),
                               h.make_light(abi.RM_LIGHT_POINT, (.9, .8, .7), pos=(3, 4, 4), func=(0.7, 0.05, 0)),
                               h.make_light(abi.RM_LIGHT_SPOT, (.8, .8, 1), direction=(0, -1, -0.2), pos=(0, 5, 1), func=(0.8, 0.02, 0),
                                            angle=0.8, penumbra=0.2))
    red = dict(ambient=(.2, 0, 0), diffuse=(.9, .1, .1), specular=(.5, .5, .5), shininess=15, reflective=(.4, .4, .4))
    blue = dict(ambient=(0, 0, .2), diffuse=(.1, .1, .9), specular=(.5, .5, .5), shininess=15, transparent=(.4, .4, .4), ior=1.4)
    floor = h.make_object(abi.RM_CUBE, model=h.translate(0, -1.1, 0) @ h.scale(12, 0.2, 8), scale_factor=0.2, diffuse=(.6, .6, .6))
    cam = h.make_camera((0, 1.2, 6), (0, -0.2, -1), (0, 1, 0), 45.0, W, H)
    cases = []
    # ties: the same sphere twice (and the same torus twice, the second pair in the other order of materials)
    cases.append((cam, [h.make_object(abi.RM_SPHERE, model=h.translate(-1.5, 0, 0), **red),
                        h.make_object(abi.RM_SPHERE, model=h.translate(-1.5, 0, 0), **blue),
                        h.make_object(abi.RM_TORUS, model=h.translate(1.5, 0, 0) @ rot_x(0.8) @ h.scale(2, 2, 2), scale_factor=2, **blue),
                        h.make_object(abi.RM_TORUS, model=h.translate(1.5, 0, 0) @ rot_x(0.8) @ h.scale(2, 2, 2), scale_factor=2, **red), floor]))
    # the camera inside a large cube that holds everything else
    cases.append((cam, [h.make_object(abi.RM_CUBE, model=h.scale(30, 30, 30), scale_factor=30, diffuse=(.3, .5, .3)),
                        h.make_object(abi.RM_CONE, model=h.translate(0, 0, 0) @ h.scale(2, 2, 2), scale_factor=2, **red),
                        h.make_object(abi.RM_OCTAHEDRON, model=h.translate(2.5, 0.3, -1), **blue)]))
    # a full table
    grid = [h.make_object([abi.RM_SPHERE, abi.RM_CUBE, abi.RM_CYLINDER, abi.RM_CAPSULE, abi.RM_DEATHSTAR][k % 5],
                          model=h.translate(-4.5 + (k % 10), -0.4 + 0.9 * (k // 10), -1.5 * (k // 10)) @ h.scale(.7, .7, .7), scale_factor=.7,
                          **(red if k % 2 else blue)) for k in range(29)]
    cases.append((cam, grid + [floor]))
    # scaleFactor twice the smallest scale; a sheared model
    shear = np.eye(4)
    shear[0, 1], shear[2, 0] = 0.6, -0.4
    cases.append((cam, [h.make_object(abi.RM_SPHERE, model=h.translate(-2, 0, 0) @ h.scale(1.5, 0.5, 1.5), scale_factor=1.0, **red),
                        h.make_object(abi.RM_CYLINDER, model=h.translate(0.5, 0, 0) @ shear @ h.scale(1.2, 1.2, 1.2), scale_factor=1.2, **blue),
                        h.make_object(abi.RM_RECTANGLE, model=h.translate(2.5, 0.2, -0.5) @ rot_x(0.4) @ h.scale(2, 2, 2), scale_factor=2, **red), floor]))
    # a single object (the runner-up bound is +inf: the march follows that object alone from its second step on)
    cases.append((cam, [h.make_object(abi.RM_TORUS, model=rot_x(0.9) @ h.scale(3, 3, 3), scale_factor=3, **red)]))
    # objects that touch: crevices where two of them stay equally near for hundreds of steps, rays that pass one object closely
    # and hit the next (the nearest object changes along the ray)
    cases.append((cam, [h.make_object(abi.RM_SPHERE, model=h.translate(-0.5, -0.5, 0), **red),
                        h.make_object(abi.RM_SPHERE, model=h.translate(0.5, -0.5, 0), **blue),
                        h.make_object(abi.RM_CUBE, model=h.translate(0, -0.5, -1.0), **red),
                        h.make_object(abi.RM_CAPSULE, model=h.translate(1.2, -1.0, 0.8) @ h.scale(2, 2, 2), scale_factor=2, **blue), floor]))
    g = h.make_globals()
    for k, (c, objs) in enumerate(cases):
        arr = (abi.RmObject * len(objs))(*objs)
        scene = (c, arr, len(objs), lights, 3, g)
        for over in ({"enableReflection": 1, "enableRefraction": 1, "numReflection": 2},
                     {"enableSoftShadow": 1, "enableAmbientOcclusion": 1, "features": abi.RM_FEAT_WHITE_BACKGROUND | abi.RM_FEAT_PERLIN_BUMP}):
            s = abi.default_settings(**over)
            ref, ref_b = h.oracle_render(scene, s, W, H, bright=True)
            out, br = renderer.render(tables_of(scene), s, W, H, bright=True)
            assert_bit_equal(out.cpu().numpy(), ref, f"skip case {k} {over}")
            assert_bit_equal(br.cpu().numpy(), ref_b, f"skip case {k} {over} bright")
        try:  # the wavefront pipeline's instantiation of the test
            lib().rm_set_kernel_path(5)
            s = abi.default_settings(enableReflection=1, numReflection=2, enableAmbientOcclusion=1)
            out = renderer.render(tables_of(scene), s, W, H)
            assert lib().rm_debug_last_path() == 5
            assert_bit_equal(out.cpu().numpy(), h.oracle_render(scene, s, W, H), f"skip case {k}, wavefront")
        finally:
            lib().rm_set_kernel_path(0)



def test_stamped_diagnostic_builds_render_the_same_frame(renderer):
    """rm_render_clocked (production code + per-wave clock stamps; scripts/wave_timeline.py, wave_lives.py): the frame is the
    production frame, the shader clock plausible, every wave that holds pixels has a life span — for the single-Mandelbulb class
    and the plain table walk; other classes are refused."""
    from raymarcher_amd import Scene
    W, H = 200, 120
    cases = [(tables_of(h.scene_mandelbulb(W, H)), abi.default_settings(fractalIters=8)),
             (Scene(path=os.path.join(SCENES, "lighting", "directional_light_2.json")).tables(W, H),
              abi.default_settings(enableSoftShadow=1, enableAmbientOcclusion=1))]
    for t, s in cases:
        ref = renderer.render(t, s, W, H)
        out, mhz, spans = renderer.render_clocked(t, s, W, H, wave_spans=True)
        assert _ieq(out, ref)
        assert 500.0 < mhz < 4000.0
        sp = spans.cpu().numpy()
        live = sp[sp[:, 1] > 0]
        assert len(live) == ((W + 7) // 8) * ((H + 7) // 8) and (live[:, 1] >= live[:, 0]).all()
    with pytest.raises(Exception):
        renderer.render_clocked(tables_of(env_scene(W, H)), abi.default_settings(features=ENV_ALL), W, H)


CXX_HOST = r'''
// A C++ host with no Python and no torch: what a maintainer of the reference links (INTEGRATION.md §1).
#include <hip/hip_runtime_api.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "raymarcher_amd.h"
extern "C" int rmo_render(const RmCamera *, const RmObject *, int, const RmLight *, int, const RmGlobals *, const RmSettings *, int, int,
                          int, int, float *, float *, RmCounters *, int);   // the checker (oracle/), test only
int main(int argc, char **argv) {
  const int W = 96, H = 54;
  RmScene *sc = nullptr;
  if (rm_scene_load(argv[1], &sc) != RM_OK) { std::printf("load: %s\n", rm_last_error()); return 2; }
  RmCameraData cd; RmCamera cam; RmGlobals g; RmSettings s;
  rm_scene_camera_data(sc, &cd);
  if (rm_camera_build(&cd, W, H, 0.1f, 100.0f, nullptr, nullptr, &cam) != RM_OK) return 3;
  if (rm_scene_globals(sc, nullptr, &g) != RM_OK) return 4;
  rm_settings_default(&s);
  s.fractalIters = 12;
  float *dFrame = nullptr; uint8_t *d8 = nullptr;
  if (hipMalloc(reinterpret_cast<void **>(&dFrame), size_t(W) * H * 16) != hipSuccess || hipMalloc(reinterpret_cast<void **>(&d8), size_t(W) * H * 4) != hipSuccess) return 5;
  int st = rm_render(&cam, rm_scene_objects(sc), rm_scene_num_objects(sc), rm_scene_lights(sc), rm_scene_num_lights(sc), &g, &s,
                     W, H, 0, H, dFrame, nullptr, nullptr);
  if (st != RM_OK) { std::printf("render: %s\n", rm_last_error()); return 6; }
  if (rm_frame_to_rgba8(dFrame, d8, W, H, nullptr) != RM_OK) return 7;
  std::vector<float> got(size_t(W) * H * 4), ref(got.size());
  std::vector<uint8_t> px(size_t(W) * H * 4);
  hipMemcpy(got.data(), dFrame, got.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(px.data(), d8, px.size(), hipMemcpyDeviceToHost);
  if (rmo_render(&cam, rm_scene_objects(sc), rm_scene_num_objects(sc), rm_scene_lights(sc), rm_scene_num_lights(sc), &g, &s, W, H, 0, H,
                 ref.data(), nullptr, nullptr, 8) != 0) return 8;
  if (std::memcmp(got.data(), ref.data(), got.size() * 4) != 0) { std::printf("frame differs from the oracle\n"); return 9;  }
  if (rm_write_png(argv[2], px.data(), W, H) != RM_OK) return 10;
  std::vector<float> host(16);
  if (rm_render(&cam, rm_scene_objects(sc), 1, rm_scene_lights(sc), 3, &g, &s, 2, 2, 0, 2, host.data(), nullptr, nullptr) != RM_ERR_INVALID_ARGUMENT) return 11;
  // the sharded frame as a single-process multi-GPU host assembles it: every device of the list renders its interleaved
  // tiles on its own stream, rm_gather_tiles brings them to the root, rm_deinterleave restores frame order (one GPU here,
  // so one shard per "device" cannot be spread: the list holds the devices there are, shards = their number)
  int ndev = rm_device_count();
  if (ndev < 1) return 12;
  if (ndev > 8) ndev = 8;
  std::vector<int> devs(ndev);
  for (int k = 0; k < ndev; k++) devs[k] = k;
  RmGather *ga = nullptr;
  if (rm_gather_create(devs.data(), ndev, &ga) != RM_OK) { std::printf("gather: %s\n", rm_last_error()); return 13; }
  const int T = 8, slot = rm_gather_slot_rows(H, T, ndev);
  std::vector<float *> tiles(ndev);
  std::vector<void *> streams(ndev);
  float *dGathered = nullptr, *dFrame2 = nullptr;
  for (int k = 0; k < ndev; k++) {
    if (rm_set_device(devs[k]) != RM_OK) return 14;
    hipStream_t st_;
    if (hipStreamCreate(&st_) != hipSuccess) return 14;
    streams[k] = st_;
    if (hipMalloc(reinterpret_cast<void **>(&tiles[k]), size_t(slot) * W * 16) != hipSuccess) return 14;
    if (rm_render_tiles(&cam, rm_scene_objects(sc), rm_scene_num_objects(sc), rm_scene_lights(sc), rm_scene_num_lights(sc), &g, &s,
                        W, H, T, k, ndev, tiles[k], nullptr, streams[k]) != RM_OK) { std::printf("tiles: %s\n", rm_last_error()); return 15; }
  }
  rm_set_device(devs[0]);
  if (hipMalloc(reinterpret_cast<void **>(&dGathered), size_t(ndev) * slot * W * 16) != hipSuccess || hipMalloc(reinterpret_cast<void **>(&dFrame2), size_t(W) * H * 16) != hipSuccess) return 16;
  if (rm_gather_tiles(ga, tiles.data(), dGathered, W, H, T, 0, streams.data()) != RM_OK) { std::printf("gather: %s\n", rm_last_error()); return 17; }
  if (rm_deinterleave(dGathered, dFrame2, W, H, T, ndev, slot, streams[0]) != RM_OK) return 18;
  if (hipStreamSynchronize(static_cast<hipStream_t>(streams[0])) != hipSuccess) return 19;
  std::vector<float> got2(got.size());
  hipMemcpy(got2.data(), dFrame2, got2.size() * 4, hipMemcpyDeviceToHost);
  if (std::memcmp(got2.data(), ref.data(), got2.size() * 4) != 0) { std::printf("gathered frame differs from the oracle\n"); return 20; }
  if (rm_gather_tiles(ga, tiles.data(), dGathered, W, H, T, ndev, streams.data()) != RM_ERR_INVALID_ARGUMENT) return 21;
  rm_gather_destroy(ga);
  // the RCCL branch itself, also on a one-GPU box: RM_GATHER_FORCE_COMM loads librccl, builds the communicator(s) with
  // ncclCommInitAll and moves EVERY shard — the root's own too — through a grouped ncclSend / ncclRecv pair
  RmGather *gb = nullptr;
  if (rm_gather_create_ex(devs.data(), ndev, RM_GATHER_FORCE_COMM, &gb) != RM_OK) { std::printf("gather_ex: %s\n", rm_last_error()); return 22; }
  int cur = -1;
  hipGetDevice(&cur);
  if (cur != devs[0]) return 23;  // creating the communicators left the caller's device alone
  hipMemsetAsync(dGathered, 0, size_t(ndev) * slot * W * 16, static_cast<hipStream_t>(streams[0]));
  if (rm_gather_tiles(gb, tiles.data(), dGathered, W, H, T, 0, streams.data()) != RM_OK) { std::printf("rccl gather: %s\n", rm_last_error()); return 24; }
  hipGetDevice(&cur);
  if (cur != devs[0]) return 25;
  if (rm_deinterleave(dGathered, dFrame2, W, H, T, ndev, slot, streams[0]) != RM_OK) return 26;
  if (hipStreamSynchronize(static_cast<hipStream_t>(streams[0])) != hipSuccess) return 27;
  hipMemcpy(got2.data(), dFrame2, got2.size() * 4, hipMemcpyDeviceToHost);
  if (std::memcmp(got2.data(), ref.data(), got2.size() * 4) != 0) { std::printf("RCCL-gathered frame differs from the oracle\n"); return 28; }
  // the 4-bytes-per-pixel gather: every shard quantises its tiles, the bytes travel, the root writes the top-down image
  std::vector<uint8_t *> tiles8(ndev);
  uint8_t *dGathered8 = nullptr, *dFrame8 = nullptr;
  for (int k = 0; k < ndev; k++) {
    rm_set_device(devs[k]);
    if (hipMalloc(reinterpret_cast<void **>(&tiles8[k]), size_t(slot) * W * 4) != hipSuccess) return 29;
    if (rm_tiles_to_rgba8(tiles[k], tiles8[k], W, rm_shard_rows(H, T, k, ndev), streams[k]) != RM_OK) return 30;
  }
  rm_set_device(devs[0]);
  if (hipMalloc(reinterpret_cast<void **>(&dGathered8), size_t(ndev) * slot * W * 4) != hipSuccess || hipMalloc(reinterpret_cast<void **>(&dFrame8), size_t(W) * H * 4) != hipSuccess) return 31;
  if (rm_gather_tiles_rgba8(gb, tiles8.data(), dGathered8, W, H, T, 0, streams.data()) != RM_OK) { std::printf("rgba8 gather: %s\n", rm_last_error()); return 32; }
  if (rm_deinterleave_rgba8(dGathered8, dFrame8, W, H, T, ndev, slot, 1, streams[0]) != RM_OK) return 33;
  if (hipStreamSynchronize(static_cast<hipStream_t>(streams[0])) != hipSuccess) return 34;
  std::vector<uint8_t> px2(px.size());
  hipMemcpy(px2.data(), dFrame8, px2.size(), hipMemcpyDeviceToHost);
  if (std::memcmp(px2.data(), px.data(), px.size()) != 0) { std::printf("RGBA8 gather differs from rm_frame_to_rgba8 of the whole frame\n"); return 35; }
  // a null tile buffer is refused before the group opens; the object stays usable
  std::vector<float *> bad(tiles);
  bad[0] = nullptr;
  if (rm_gather_tiles(gb, bad.data(), dGathered, W, H, T, 0, streams.data()) != RM_ERR_INVALID_ARGUMENT) return 36;
  if (rm_gather_tiles(gb, tiles.data(), dGathered, W, H, T, 0, streams.data()) != RM_OK) return 37;
  if (hipStreamSynchronize(static_cast<hipStream_t>(streams[0])) != hipSuccess) return 38;
  rm_gather_destroy(gb);
  rm_scene_free(sc);
  std::printf("ok\n");
  return 0;
}
'''


def test_cxx_host_without_python(renderer, tmp_path):
    """The drop-in boundary exercised the way the reference would use it: a C++ program (hipcc) that loads a scenefile,
    renders through the C ABI into hipMalloc'ed memory, checks the frame against the oracle bit for bit, writes the PNG."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "host.cpp"
    src.write_text(CXX_HOST)
    exe = tmp_path / "host"
    libdir, odir = os.path.join(root, "raymarcher_amd", "lib"), os.path.join(root, "oracle", "_build")
    cmd = [hipcc, "-std=c++17", "-O1", "-I", os.path.join(root, "include"), str(src), "-o", str(exe), "-L", libdir, "-lraymarcher_amd",
           "-L", odir, "-lrm_oracle", f"-Wl,-rpath,{libdir}", f"-Wl,-rpath,{odir}"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    png = tmp_path / "bulb.png"
    r = subprocess.run([str(exe), os.path.join(SCENES, "simple", "unit_mandelbulb.json"), str(png)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "ok" in r.stdout, (r.returncode, r.stdout, r.stderr[-800:])
    from PIL import Image
    img = np.asarray(Image.open(png))
    assert img.shape == (54, 96, 4) and (img[..., :3] != 255).any(-1).mean() > 0.2


def test_single_process_multi_gpu_host_runs(renderer, tmp_path):
    """scripts/mgpu_host.cpp — one process, every visible device, three frames in flight per device, RCCL gather to device 0 —
    builds against the C ABI alone and runs here with the communicator forced (one GPU): float4 and RGBA8 gathers."""
    import json
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "mgpu_host"
    libdir = os.path.join(root, "raymarcher_amd", "lib")
    p = subprocess.run([hipcc, "-std=c++17", "-O1", "-I", os.path.join(root, "include"), os.path.join(root, "scripts", "mgpu_host.cpp"), "-o",
                        str(exe), "-L", libdir, "-lraymarcher_amd", f"-Wl,-rpath,{libdir}"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    scene = os.path.join(SCENES, "simple", "unit_mandelbulb.json")
    for extra in ([], ["--rgba8"]):
        r = subprocess.run([str(exe), scene, "--size", "640", "360", "--frames", "12", "--iters", "12", "--force-comm"] + extra,
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, (r.stdout, r.stderr[-800:])
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert line["n_gpus"] >= 1 and line["value"] > 0 and line["gather"] == ("rgba8" if extra else "float4")


def test_degenerate_knobs_bit_exact(renderer):
    """Loop bounds at zero, no lights, one-pixel frames, NaN / inf in the tables: no hang, no fault, same bits as the oracle."""
    W, H = 40, 24
    cases = [(h.scene_mandelbulb(W, H), dict(maxSteps=0)), (h.scene_mandelbulb(W, H), dict(fractalIters=0)),
             (h.scene_mandelbulb(W, H), dict(maxSteps=1, fractalIters=1)), (menger_scene(W, H), dict(mengerLevels=0, enableReflection=1)),
             (reflect_refract_scene(W, H), dict(numReflection=0, enableReflection=1, enableRefraction=1)),
             (reflect_refract_scene(W, H)[:3] + ((abi.RmLight * 1)(), 0) + (h.make_globals(),), dict(enableAmbientOcclusion=1))]
    for k, (scene, over) in enumerate(cases):
        s = abi.default_settings(**over)
        assert_bit_equal(renderer.render(tables_of(scene), s, W, H).cpu().numpy(), h.oracle_render(scene, s, W, H), f"degenerate {k} {over}")
    for w, hh in ((1, 1), (1, 7), (33, 1)):
        scene = h.scene_mandelbulb(w, hh)
        s = abi.default_settings(fractalIters=12)
        assert_bit_equal(renderer.render(tables_of(scene), s, w, hh).cpu().numpy(), h.oracle_render(scene, s, w, hh), f"{w}x{hh}")
    # non-finite numbers in the tables must not hang or fault (values then follow IEEE on both sides)
    scene = reflect_refract_scene(W, H)
    scene[1][0].invModel[12] = float("nan")
    scene[1][1].scaleFactor = float("inf")
    scene[3][0].dir[0] = float("nan")
    s = abi.default_settings(enableReflection=1, maxSteps=32)
    got, ref = renderer.render(tables_of(scene), s, W, H).cpu().numpy(), h.oracle_render(scene, s, W, H)
    both_nan = np.isnan(got) & np.isnan(ref)
    assert ((got.view(np.uint32) == ref.view(np.uint32)) | both_nan).all()
