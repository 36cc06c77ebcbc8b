"""The launcher's march-ending bounds (rm_debug_cull_bounds: a ball and, for flat or elongated scenes, an axis-aligned box): a
march whose miss distance nobody reads ends where its ray leaves ball ∩ box, which is only valid if NOTHING outside can be hit —
every distance value out there must exceed the hit threshold SURFACE_DIST = 0.001 (frag:32).  Checked here with the oracle's
sdScene on points outside the bounds (no GPU needed); that the pixels do not change is the GPU parity suite's business."""
import ctypes as C
import os

import numpy as np

import helpers as h
from raymarcher_amd import abi, lib
from raymarcher_amd.render import Scene

SCENES = os.path.join(os.path.dirname(__file__), "golden", "scenes")
TYPES = [abi.RM_CUBE, abi.RM_CONE, abi.RM_CYLINDER, abi.RM_SPHERE, abi.RM_OCTAHEDRON, abi.RM_TORUS, abi.RM_CAPSULE,
         abi.RM_DEATHSTAR, abi.RM_RECTANGLE, abi.RM_MENGERSPONGE]


def rot(axis, a):
    c, s = np.cos(a), np.sin(a)
    M = np.eye(4)
    i, j = [(1, 2), (0, 2), (0, 1)][axis]
    M[i, i], M[i, j], M[j, i], M[j, j] = c, -s, s, c
    return M


def bounds(objs, n, g):
    out = np.zeros(14, dtype=np.float32)
    assert lib().rm_debug_cull_bounds(objs, n, C.byref(g), out.ctypes.data_as(C.POINTER(C.c_float))) == 0
    return {"ok": bool(out[0]), "c": out[1:4].astype(np.float64), "R": float(np.sqrt(out[4])), "Rsoft": float(np.sqrt(out[5])),
            "box": bool(out[6]), "lo": out[7:10].astype(np.float64), "hi": out[10:13].astype(np.float64), "lip": float(out[13])}


def distances(objs, n, g, pts):
    pts = np.ascontiguousarray(pts, dtype=np.float32)
    out = np.empty((len(pts), 4), dtype=np.float32)
    s = abi.default_settings(mengerLevels=5)
    assert h.oracle().rmo_probe_sdscene(objs, n, C.byref(g), C.byref(s), h.fptr(pts), h.fptr(out), len(pts)) == 0
    return out[:, 0]


def random_scene(rng, flat):
    f = rng.uniform
    objs = []
    for _ in range(int(rng.integers(1, 8))):
        ty = int(rng.choice(TYPES))
        sx, sy, sz = (float(f(0.3, 2.5)) for _ in range(3))
        if flat:  # a row of objects over a floor slab: the box is far tighter than the ball
            M = h.translate(f(-6, 6), f(-0.5, 0.8), f(-1, 1))
        else:
            M = h.translate(f(-2, 2), f(-2, 2), f(-2, 2))
        M = M @ rot(0, f(-3, 3)) @ rot(1, f(-3, 3)) @ rot(2, f(-3, 3)) @ h.scale(sx, sy, sz)
        objs.append(h.make_object(ty, model=M, scale_factor=min(sx, sy, sz)))
    if flat:
        objs.append(h.make_object(abi.RM_CUBE, model=h.translate(0, -1.2, 0) @ h.scale(16, 0.2, 5), scale_factor=0.2))
    return (abi.RmObject * len(objs))(*objs), len(objs)


def outside_points(rng, b, n):
    """Points just outside the ball, far outside it, and (where there is a box) just outside each of its faces but possibly
    deep inside the ball."""
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    r = b["R"] * (1.0 + np.concatenate([rng.uniform(1e-6, 1e-2, n // 2), rng.uniform(0, 30, n - n // 2)]))
    pts = [b["c"] + d * r[:, None]]
    if b["box"]:
        ext = b["hi"] - b["lo"]
        for axis in range(3):
            for side, plane in ((-1, b["lo"][axis]), (1, b["hi"][axis])):
                p = b["lo"] + rng.uniform(-0.3, 1.3, (n, 3)) * ext  # anywhere over (and around) the face
                p[:, axis] = plane + side * np.concatenate([rng.uniform(1e-6, 1e-3, n // 2), rng.uniform(0, 5, n - n // 2)])
                pts.append(p)
    return np.concatenate(pts)


def test_nothing_outside_the_bounds_is_within_the_hit_threshold():
    rng = np.random.default_rng(20261006)
    g = h.make_globals(itime=3.7)
    boxes = 0
    for i in range(60):
        objs, n = random_scene(rng, flat=(i % 2 == 0))
        b = bounds(objs, n, g)
        assert b["ok"]
        boxes += b["box"]
        pts = outside_points(rng, b, 4000)
        d = distances(objs, n, g, pts)
        assert np.isfinite(d).all() and d.min() > 0.004, f"scene {i}: a distance value {d.min()} outside the bounds"
    assert boxes >= 20  # the flat scenes do get their box


def test_box_only_where_it_is_much_tighter_than_the_ball():
    g = h.make_globals()
    t = Scene(path=os.path.join(SCENES, "lighting", "directional_light_2.json")).tables(64, 36, load_textures=False)
    b = bounds(t.objects, t.num_objects, g)
    assert b["ok"] and b["box"]
    vol = np.prod(b["hi"] - b["lo"]) / (4.18879 * b["R"] ** 3)
    assert vol < 0.3
    t = Scene(path=os.path.join(SCENES, "simple", "unit_mengersponge.json")).tables(64, 36)
    b = bounds(t.objects, t.num_objects, g)
    assert b["ok"] and not b["box"]  # a lone cube: box / ball volume 0.39 — the ball alone (measured: the box costs more than it saves)
    # types without a bound: no culling at all
    o = (abi.RmObject * 1)(h.make_object(abi.RM_SIERPINSKI))
    assert not bounds(o, 1, g)["ok"]
    assert not bounds(o, 0, g)["ok"]


def test_distance_values_are_lipschitz_and_bounded_below_by_the_object_ball():
    """What the table walk's skip test (sdScene<…, SKIP>, nextMinBound) rests on: (i) sdScene's value changes by at most
    `lip` per unit of world length; (ii) a primitive's value is at least (|p_object| − boundR)·scaleFactor."""
    rng = np.random.default_rng(7)
    g = h.make_globals(itime=5.1)
    for i in range(40):
        objs, n = random_scene(rng, flat=(i % 2 == 0))
        b = bounds(objs, n, g)
        assert np.isfinite(b["lip"]) and 0.99 < b["lip"] < 1.01  # scaleFactor = the smallest scale undoes the stretch
        p = rng.uniform(-8, 8, (20000, 3))
        step = rng.normal(size=(20000, 3)) * (10.0 ** rng.uniform(-3, 0.5, (20000, 1)))
        a, c = distances(objs, n, g, p), distances(objs, n, g, p + step)
        dist = np.linalg.norm((p + step).astype(np.float32).astype(np.float64) - p.astype(np.float32).astype(np.float64), axis=1)
        assert (np.abs(c.astype(np.float64) - a) <= b["lip"] * dist * 1.0001 + 2e-5).all(), f"scene {i}"
    bound_r = {abi.RM_CUBE: 0.8662, abi.RM_CONE: .7073, abi.RM_CYLINDER: .7073, abi.RM_SPHERE: .5001, abi.RM_OCTAHEDRON: .5001,
               abi.RM_TORUS: .6252, abi.RM_CAPSULE: .6002, abi.RM_DEATHSTAR: .5001, abi.RM_RECTANGLE: .7073}
    for ty, r in bound_r.items():
        o = (abi.RmObject * 1)(h.make_object(ty))
        d = rng.normal(size=(200000, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        pts = (d * np.concatenate([rng.uniform(0, 3, 100000), 10 ** rng.uniform(0, 4, 100000)])[:, None]).astype(np.float32)
        lb = np.linalg.norm(pts.astype(np.float64), axis=1) - r
        assert (distances(o, 1, g, pts) >= lb * (1 - 1e-5) - 1e-6).all(), f"type {ty}"
    o = (abi.RmObject * 2)(h.make_object(abi.RM_SPHERE), h.make_object(abi.RM_MANDELBULB))
    assert bounds(o, 2, g)["lip"] == np.inf  # a fractal in the table: no seed


def test_bounds_hold_on_wide_random_tables():
    """The same two properties on tables drawn WIDE (helpers.random_tablewalk_objects: arbitrary-axis rotations, shear,
    anisotropy up to 25, scaleFactors that are not the smallest scale, nested and coincident objects, up to 30 objects) — the
    distribution of the GPU soak test test_random_tablewalk_scenes_bit_exact: nothing outside ball ∩ box comes within the hit
    threshold, and sdScene changes by at most `lip` per unit of world length (lip is then NOT 1)."""
    rng = np.random.default_rng(20261011)
    g = h.make_globals()
    lips = []
    for i in range(80):
        lst = h.random_tablewalk_objects(rng, materials=False)
        objs, n = (abi.RmObject * len(lst))(*lst), len(lst)
        b = bounds(objs, n, g)
        assert b["ok"] and np.isfinite(b["lip"]) and b["lip"] > 0
        lips.append(b["lip"])
        d = distances(objs, n, g, outside_points(rng, b, 3000))
        assert np.isfinite(d).all() and d.min() > 0.0039, f"table {i}: a distance value {d.min()} outside the bounds"
        p = rng.uniform(-7, 7, (8000, 3))
        step = rng.normal(size=(8000, 3)) * (10.0 ** rng.uniform(-3, 0.5, (8000, 1)))
        a, c = distances(objs, n, g, p), distances(objs, n, g, p + step)
        dist = np.linalg.norm((p + step).astype(np.float32).astype(np.float64) - p.astype(np.float32).astype(np.float64), axis=1)
        excess = np.abs(c.astype(np.float64) - a) - (b["lip"] * dist * 1.0001 + 2e-5 * max(1.0, b["lip"]))
        assert (excess <= 0).all(), f"table {i}: sdScene is not {b['lip']}-Lipschitz (excess {excess.max():.3e})"
    assert max(lips) > 1.5 and min(lips) < 1.01  # both regimes were drawn
