"""Host side (scenefile loader + camera) against golden uniform tables produced by the reference's own,
unmodified loader/camera compiled in the build container (oracle/ref/, oracle/tools/gen_host_goldens.py).
Inputs are the reference's scenefiles (data) under tests/golden/scenes/.  CPU only: these entry points of
the C-ABI library do no GPU work."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import helpers as h
from raymarcher_amd import RaymarcherError, abi, lib
from raymarcher_amd.render import Scene

GOLD = os.path.join(os.path.dirname(__file__), "golden")
with open(os.path.join(GOLD, "host_tables.json")) as f:
    TABLES = json.load(f)

OK_SCENES = sorted(k for k, v in TABLES.items() if v.get("ok"))
BAD_SCENES = sorted(k for k, v in TABLES.items() if not v.get("ok"))


def close(a, b, rel=2e-6, abs_=1e-6):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.all(np.abs(a - b) <= abs_ + rel * np.maximum(np.abs(a), np.abs(b)))


@pytest.mark.parametrize("rel", OK_SCENES)
def test_loader_matches_reference_tables(rel):
    g = TABLES[rel]
    sc = Scene(path=os.path.join(GOLD, "scenes", rel))
    ref_objs = g["objects"]
    # the reference appends one emissive rectangle per area light at render-scene build time
    # (raymarchscene.cpp:121-133); the golden dump lists parser shapes only
    n_area = sum(1 for l in g["lights"] if l["type"] == abi.RM_LIGHT_AREA)
    assert sc.num_objects == len(ref_objs) + n_area
    assert sc.num_lights == len(g["lights"])
    objs, lights = lib().rm_scene_objects(sc._h), lib().rm_scene_lights(sc._h)
    for i, ro in enumerate(ref_objs):
        o = objs[i]
        assert o.type == ro["type"]
        assert close(list(o.invModel), ro["invModel"], rel=2e-5, abs_=2e-6), (rel, i, list(o.invModel), ro["invModel"])
        assert close(o.scaleFactor, ro["scaleFactor"])
        for name in ("shininess", "blend", "ior"):
            assert getattr(o, name) == np.float32(ro[name])
        for name in ("cAmbient", "cDiffuse", "cSpecular", "cReflective", "cTransparent"):
            assert list(getattr(o, name)) == [float(np.float32(v)) for v in ro[name]]
        assert (o.texLoc != -1) == ro["textured"]
        assert (sc.texture_of(i) is not None) == ro["textured"]
        if ro["textured"]:
            assert (o.repeatU, o.repeatV) == (ro["repeatU"], ro["repeatV"])
        assert o.isEmissive == 0 and o.lightIdx == -1
    area = [i for i, l in enumerate(g["lights"]) if l["type"] == abi.RM_LIGHT_AREA]
    for k, li in enumerate(area):
        # raymarchscene.cpp:126-133 / raymarchobj.h:16-23: rectangle with the light's transform, colour and index
        o, rl = objs[len(ref_objs) + k], g["lights"][li]
        assert o.type == abi.RM_RECTANGLE and o.isEmissive == 1 and o.lightIdx == li and o.texLoc == -1
        assert close(list(o.invModel), rl["ctmInv"], rel=2e-5, abs_=2e-6) and o.scaleFactor == 1.0
        assert list(o.color) == [float(np.float32(v)) for v in rl["color"]]
    for i, ro in enumerate(ref_objs):  # the image every textured primitive names (resolved path, tail compared)
        tex = sc.texture_of(i)
        assert (tex or "").replace(os.sep, "/").endswith(ro["textureFile"]) and bool(tex) == bool(ro["textureFile"])
    for i, rl in enumerate(g["lights"]):
        li = lights[i]
        assert li.type == rl["type"]
        assert list(li.color) == [float(np.float32(v)) for v in rl["color"]]
        assert close(list(li.pos), rl["pos"]) and close(list(li.dir), rl["dir"])
        assert list(li.func) == [float(np.float32(v)) for v in rl["func"]]
        assert close(li.angle, rl["angle"], rel=1e-7) and close(li.penumbra, rl["penumbra"], rel=1e-7)
        if rl["type"] == abi.RM_LIGHT_AREA:
            # configureLightsUniforms, realtimerender.cpp:682-693: twoSided = true, the parsed intensity (which the reference's
            # reader never stores: 0), points[k] = ctm · corner k of the unit square (realtime.h:136-141: tl, tr, br, bl)
            assert li.twoSided == 1 and li.intensity == np.float32(rl["intensity"])
            m = np.array(rl["ctm"], dtype=np.float64).reshape(4, 4).T
            for k, (cx, cy) in enumerate(((-0.5, 0.5), (0.5, 0.5), (0.5, -0.5), (-0.5, -0.5))):
                assert close([li.points[k][j] for j in range(3)], list((m @ np.array([cx, cy, 0.0, 1.0]))[:3]), rel=2e-6, abs_=2e-6)
    gl = abi.RmGlobals()
    assert lib().rm_scene_globals(sc._h, None, C.byref(gl)) == 0
    assert (gl.ka, gl.kd, gl.ks) == tuple(float(np.float32(g[k])) for k in ("ka", "kd", "ks"))
    cd = sc.camera_data()
    assert close(list(cd.pos), g["camPos"]) and close(list(cd.look), g["camLook"]) and close(list(cd.up), g["camUp"])
    assert close(cd.heightAngle, g["heightAngle"], rel=1e-7)
    # camera matrices at both golden sizes
    for key, cam in g["camera"].items():
        W, H = map(int, key.split("x"))
        view, proj, out = (C.c_float * 16)(), (C.c_float * 16)(), abi.RmCamera()
        assert lib().rm_camera_build(C.byref(cd), W, H, 0.1, 100.0, view, proj, C.byref(out)) == 0
        assert close(list(view), cam["view"], rel=2e-6, abs_=1e-6)
        assert close(list(proj), cam["proj"], rel=2e-6, abs_=1e-9)
        assert close(list(out.invProjView), cam["invProjView"], rel=2e-5, abs_=2e-4), (rel, key)
        assert out.initialFar == 100.0


@pytest.mark.parametrize("rel", BAD_SCENES)
def test_scenes_the_reference_rejects_are_rejected(rel):
    with pytest.raises(RaymarcherError) as e:
        Scene(path=os.path.join(GOLD, "scenes", rel))
    assert e.value.status == abi.RM_ERR_PARSE


def test_bitwise_agreement_rate_with_reference_math():
    """The loader evaluates the same binary32 expression order as the reference's math library, so almost
    every matrix word should be IDENTICAL, not just close.  Guard against silent drift."""
    same = total = 0
    for rel in OK_SCENES:
        g = TABLES[rel]
        sc = Scene(path=os.path.join(GOLD, "scenes", rel))
        objs = lib().rm_scene_objects(sc._h)
        for i, ro in enumerate(g["objects"]):
            a = np.array(list(objs[i].invModel), dtype=np.float32)
            b = np.array(ro["invModel"], dtype=np.float32)
            same += int((a == b).sum())
            total += 16
    assert same / total > 0.97, f"only {same}/{total} invModel words identical"


SCHEMA_ERRORS = {
    "not json": "{",
    "root not object": "[1,2]",
    "missing globalData": '{"cameraData": {"position":[0,0,1],"up":[0,1,0],"heightAngle":30,"look":[0,0,-1]}}',
    "unknown root key": '{"globalData":{"ambientCoeff":1,"diffuseCoeff":1,"specularCoeff":1},"cameraData":{"position":[0,0,1],"up":[0,1,0],"heightAngle":30,"look":[0,0,-1]},"bogus":1}',
    "look and focus": '{"globalData":{"ambientCoeff":1,"diffuseCoeff":1,"specularCoeff":1},"cameraData":{"position":[0,0,1],"up":[0,1,0],"heightAngle":30,"look":[0,0,-1],"focus":[0,0,0]}}',
    "bad primitive type": '{"globalData":{"ambientCoeff":1,"diffuseCoeff":1,"specularCoeff":1},"cameraData":{"position":[0,0,1],"up":[0,1,0],"heightAngle":30,"look":[0,0,-1]},"groups":[{"primitives":[{"type":"terrain"}]}]}',
    "spot without angle": '{"globalData":{"ambientCoeff":1,"diffuseCoeff":1,"specularCoeff":1},"cameraData":{"position":[0,0,1],"up":[0,1,0],"heightAngle":30,"look":[0,0,-1]},"groups":[{"lights":[{"type":"spot","color":[1,1,1],"direction":[0,-1,0],"penumbra":10,"attenuationCoeff":[1,0,0]}]}]}',
    "translate wrong arity": '{"globalData":{"ambientCoeff":1,"diffuseCoeff":1,"specularCoeff":1},"cameraData":{"position":[0,0,1],"up":[0,1,0],"heightAngle":30,"look":[0,0,-1]},"groups":[{"translate":[1,2]}]}',
}


@pytest.mark.parametrize("name", list(SCHEMA_ERRORS))
def test_schema_errors(name):
    with pytest.raises(RaymarcherError) as e:
        Scene(text=SCHEMA_ERRORS[name])
    assert e.value.status == abi.RM_ERR_PARSE
    assert str(e.value)


def test_missing_file_is_io_error():
    with pytest.raises(RaymarcherError) as e:
        Scene(path="/nonexistent/dir/scene.json")
    assert e.value.status == abi.RM_ERR_IO


def test_camera_against_independent_numpy_restatement():
    for (pos, look, up, ang, W, H) in [((0, 0, 4.5), (0, 0, -4.5), (0, 1, 0), 30.0, 3840, 2160),
                                       ((3, 2, 5), (-3, -1.5, -5), (0.1, 1, 0), 45.0, 1024, 768),
                                       ((-2, 7, 1), (0.5, -1, 0.2), (0, 0, 1), 60.0, 640, 480)]:
        from raymarcher_amd.render import build_camera
        cam, view, proj = build_camera(pos, look, up, np.deg2rad(ang), W, H)
        v, p, inv = h.camera_numpy(pos, look, up, np.deg2rad(ang), W, H)
        assert close(view, v.T.reshape(-1), rel=1e-5, abs_=1e-6)
        assert close(proj, p.T.reshape(-1), rel=1e-5, abs_=1e-8)
        assert close(list(cam.invProjView), inv.T.reshape(-1), rel=1e-4, abs_=1e-3)


REF_SCENES = "/root/reference/scenefiles"


@pytest.mark.skipif(not os.path.isdir(REF_SCENES), reason="the reference checkout (with its texture_store) is only in the build container")
def test_every_textured_scenefile_loads_with_its_images():
    """Container-only integration check: every scenefile of the reference that names a texture loads through the
    product's loader AND its image files decode (PNG, JPEG, GIF) into the texture slots the renderer will index; one
    of them is rendered by the oracle.  Also the three sky-box sets of getCubeMapWithType."""
    import glob
    from raymarcher_amd import Scene, abi, lib
    from raymarcher_amd.render import load_image
    import helpers as h
    seen, kinds = 0, set()
    for path in sorted(glob.glob(os.path.join(REF_SCENES, "*", "*.json"))):
        if "textureFile" not in open(path).read():
            continue
        try:
            sc = Scene(path=path)
        except Exception:
            continue  # scenefiles the reference's own loader rejects as well (tests above)
        t = sc.tables(64, 48)
        used = {t.objects[i].texLoc for i in range(t.num_objects) if t.objects[i].texLoc >= 0}
        if not used:
            continue
        assert t.textures is not None and len(t.textures) == max(used) + 1, path
        for a in t.textures:
            assert a.dtype == np.uint8 and a.ndim == 3 and a.shape[2] == 4 and a.size > 0
        kinds |= {os.path.splitext(sc.texture_of(i))[1].lower() for i in range(t.num_objects) if sc.texture_of(i)}
        seen += 1
        if path.endswith("unit_sphere.json"):
            img = h.oracle_render((t.camera, t.objects, t.num_objects, t.lights, t.num_lights, t.globals_),
                                  abi.default_settings(maxSteps=64), 64, 48, textures=t.textures)
            assert np.isfinite(img).all() and img[..., :3].std() > 0.05
    assert seen >= 20 and {".png", ".gif"} <= kinds, (seen, kinds)
    for which in (1, 2, 3):
        faces = [load_image(os.path.join(REF_SCENES, lib().rm_skybox_face_path(which, f).decode()), flip_vertical=True) for f in range(6)]
        assert all(a.shape == faces[0].shape and a.shape[0] == a.shape[1] for a in faces), which


@pytest.mark.parametrize("name", ["unit_mandelbulb", "unit_mengersponge"])
def test_bench_scene_constants_equal_the_scenefile(name):
    """bench.py / the full-size GPU tests take the north-star scenes from raymarcher_amd.scenes (scenefiles restated as
    constants, no file IO in the product); they must be the tables the loader gives for the reference's own scenefile,
    byte for byte — camera, object, lights and globals."""
    from raymarcher_amd import scenes
    fn = {"unit_mandelbulb": scenes.mandelbulb, "unit_mengersponge": scenes.mengersponge}[name]
    for W, H in ((3840, 2160), (7680, 4320), (96, 54)):
        a = fn(W, H)
        b = Scene(path=os.path.join(GOLD, "scenes", "simple", name + ".json")).tables(W, H)
        raw = lambda x: bytes(memoryview(x).cast("B"))
        assert (a.num_objects, a.num_lights) == (b.num_objects, b.num_lights) == (1, 3)
        assert raw(a.camera) == raw(b.camera) and raw(a.objects) == raw(b.objects)
        assert raw(a.lights) == raw(b.lights) and raw(a.globals_) == raw(b.globals_)


def test_self_referencing_template_groups_terminate():
    """A template group that contains itself (here twice, which a depth cap alone would expand 2^depth times) is cut where
    it re-enters: the load returns at once with the primitives outside the cycle."""
    import time
    text = json.dumps({
        "name": "root", "globalData": {"ambientCoeff": 0.5, "diffuseCoeff": 0.5, "specularCoeff": 0.5},
        "cameraData": {"position": [0, 0, 4], "up": [0, 1, 0], "heightAngle": 30.0, "look": [0, 0, -1]},
        "templateGroups": [{"name": "loop", "primitives": [{"type": "sphere", "diffuse": [1, 1, 1]}],
                            "groups": [{"name": "loop"}, {"name": "loop"}]}],
        "groups": [{"name": "loop"}, {"primitives": [{"type": "cube", "diffuse": [1, 0, 0]}]}]})
    t0 = time.perf_counter()
    sc = Scene(text=text)
    assert time.perf_counter() - t0 < 2.0
    assert 2 <= sc.num_objects <= 30


def test_ray_planes_of_launcher_and_oracle_are_the_same_bits():
    """The corner values of nearClip / farClip from which every primary ray is interpolated: the launcher
    computes them on the host (rm_debug_ray_planes = what it stages for the kernels), the oracle in C — the same fused
    sequence, so the same 48 words for every camera; and in float64 they are what raymarch.vert:23-24 says."""
    L = lib()
    rng = np.random.default_rng(5)
    cams = [Scene(path=os.path.join(GOLD, "scenes", rel)).tables(W, H).camera
            for rel, W, H in (("simple/unit_mandelbulb.json", 3840, 2160), ("lighting/reflections_complex.json", 64, 36),
                              ("simple/volumetric.json", 96, 54))]
    for _ in range(20):
        eye = rng.uniform(-30, 30, 3)
        cams.append(h.make_camera(tuple(eye), tuple(rng.normal(size=3)), (0, 1, 0), float(rng.uniform(20, 90)), 640, 360,
                                  far=float(rng.choice([100.0, 2000.0]))))
    for cam in cams:
        got = np.zeros(48, dtype=np.float32)
        ref = np.zeros(48, dtype=np.float32)
        assert L.rm_debug_ray_planes(C.byref(cam), got.ctypes.data_as(C.POINTER(C.c_float))) == 0
        h.oracle().rmo_ray_planes(C.byref(cam), h.fptr(ref))
        assert (got.view(np.uint32) == ref.view(np.uint32)).all()
        M = np.array(list(cam.invProjView), dtype=np.float64).reshape(4, 4).T  # column-major
        P = got.reshape(2, 2, 3, 4).astype(np.float64)
        for tri, sg in enumerate((-1.0, 1.0)):
            for k, z in enumerate((-1.0, 1.0)):
                p0, p1, p2 = (M @ np.array([x, y, z, 1.0]) for x, y in ((sg, sg), (-sg, sg), (sg, -sg)))
                scale = np.abs(M).max() * 4
                assert np.abs(P[tri, k, 0] - p0).max() <= 1e-6 * scale
                assert np.abs(P[tri, k, 1] - (p1 - p0)).max() <= 2e-6 * scale
                assert np.abs(P[tri, k, 2] - (p2 - p0)).max() <= 2e-6 * scale
