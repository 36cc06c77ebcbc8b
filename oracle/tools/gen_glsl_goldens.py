#!/usr/bin/env python3
"""Generate tests/golden/glsl/*.npz: outputs of the REFERENCE SHADER ITSELF (resources/raymarch.frag, adapted
mechanically to GLSL ES 3.00 in memory — oracle/tools/glsl_ref/essl_adapt.py) executed on the SwiftShader
software rasteriser of this container.  TEST INFRASTRUCTURE, container-only; the fixtures (inputs = ABI table
bytes, outputs = float32 arrays) are what travels.  Run:  python oracle/tools/gen_glsl_goldens.py
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle", "tools", "glsl_ref"))

import helpers as h  # noqa: E402
import run_ref  # noqa: E402
import test_gpu_parity as tg  # noqa: E402
from raymarcher_amd import abi  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "glsl")


def subset(sc, idx):
    cam, objs, no, lights, nl, g = sc
    o = (abi.RmObject * len(idx))(*[objs[i] for i in idx])
    return cam, o, len(idx), lights, nl, g


def pack(scene, settings):
    cam, objs, no, lights, nl, g = scene
    raw = lambda x: np.frombuffer(bytes(x), dtype=np.uint8)
    return dict(cam=raw(cam), objs=raw(objs), num_objects=no, lights=raw(lights), num_lights=nl, globals=raw(g),
                settings=raw(settings))


def frame_case(name, scene, settings, W, H, texture=None, ub10=False, ub1=False, **res):
    """res: noise= / skybox= (six faces) / ltc1=, ltc2= — the sampler inputs of tg.resource_case, stored in the fixture."""
    ltc = (res["ltc1"], res["ltc2"]) if "ltc1" in res else None
    rgba, bright = run_ref.render(scene, settings, W, H, texture, noise=res.get("noise"), skybox=res.get("skybox"), ltc=ltc, ub10=ub10, ub1=ub1)
    extra = {} if texture is None else {"texture": texture}
    for k, v in res.items():
        if k == "noise":  # 256 KB of random bytes: stored once, shared by every fixture that samples it
            np.save(os.path.join(OUT, "noise_synthetic.npy"), v)
            extra["uses_noise"] = 1
        else:
            extra["res_" + k] = np.stack(v) if k == "skybox" else v
    np.savez_compressed(os.path.join(OUT, f"frame_{name}.npz"), W=W, H=H, rgba=rgba, bright=bright, **extra, **pack(scene, settings))
    print("frame", name, rgba.shape, float(np.nanmax(rgba)))


def probe_case(name, kind, scene, settings, pts):
    out = run_ref.probe(kind, scene, settings, pts)
    np.savez_compressed(os.path.join(OUT, f"probe_{name}.npz"), kind=kind, pts=np.asarray(pts, dtype=np.float32), out=out,
                        **pack(scene, settings))
    print("probe", name, out.shape)


def env_cases():
    W, H = 64, 36
    sky_terr = abi.RM_FEAT_SKY_BACKGROUND | abi.RM_FEAT_TERRAIN
    frame_case("env_terrain_sky", tg.env_scene(W, H), abi.default_settings(features=sky_terr), W, H)
    frame_case("env_terrain_cloud_sky", tg.env_scene(W, H), abi.default_settings(features=sky_terr | abi.RM_FEAT_CLOUD), W, H)
    frame_case("env_all_reflect", tg.env_scene(W, H, (0, 560, 0), (0.2, 0.3, -1)),
               abi.default_settings(features=tg.ENV_ALL, enableReflection=1), W, H)
    # the same two cloud frames with the shader's unset `nnd` given the oracle's UB10 value (essl_adapt.define_ub10)
    frame_case("env_terrain_cloud_sky_ub10", tg.env_scene(W, H), abi.default_settings(features=sky_terr | abi.RM_FEAT_CLOUD), W, H, ub10=True)
    frame_case("env_all_reflect_ub10", tg.env_scene(W, H, (0, 560, 0), (0.2, 0.3, -1)),
               abi.default_settings(features=tg.ENV_ALL, enableReflection=1), W, H, ub10=True)
    if len(sys.argv) > 2 and sys.argv[2] == "frames":
        return
    # function-level probes of the procedural layers (the defines must be on for these functions to exist)
    orig = run_ref.build_program

    def with_env(defines, consts, **kw):
        d = dict(defines)
        d.update({"CLOUD": True, "TERRAIN": True, "SKY_BACKGROUND": True, "WHITE_BACKGROUND": False})
        return orig(d, consts, **kw)

    run_ref.build_program = with_env
    try:
        rng = np.random.default_rng(7)
        pts = np.stack([rng.uniform(-2000, 2000, 4096), rng.uniform(600, 1200, 4096), rng.uniform(-2000, 2000, 4096)], 1)
        sc, s = tg.env_scene(8, 8), abi.default_settings(features=tg.ENV_ALL)
        probe_case("env_cloudsfbm", "cloudsfbm", sc, s, pts)
        probe_case("env_cloudsmap", "cloudsmap", sc, s, pts)
        probe_case("env_terrain", "terrain", sc, s, pts)
    finally:
        run_ref.build_program = orig


def texture_cases():
    """unit_sphere.json-like: the four textured primitive types sharing ONE texture (the ESSL adaptation reads
    objTextures[0] for every texLoc), LINEAR filtering, REPEAT wrap."""
    W, H = 64, 48
    scene = tg.textured_scene(W, H)
    for o in scene[1]:
        if o.texLoc != -1:
            o.texLoc = 0
    frame_case("textured_prims", scene, abi.default_settings(features=abi.RM_FEAT_WHITE_BACKGROUND), W, H,
               texture=tg.synthetic_textures()[1])


def post_cases():
    """Post passes (blur.frag ×10 / hdr.frag / fxaa.frag through fullscreen.vert) on SwiftShader, driven as
    Realtime::applyBloom / applyLightEffects / applyFXAA do, on an over-bright reflective frame from the oracle."""
    import run_post
    W, H = 72, 48
    scene = tg.reflect_refract_scene(W, H)
    for li in scene[3]:
        for k in range(3):
            li.color[k] *= 2.5
    frag, bright = h.oracle_render(scene, abi.default_settings(enableReflection=1), W, H, bright=True)
    assert (bright[..., :3].max(-1) > 0).mean() > 0.02, "the bloom source must not be empty"
    cases = {
        "gamma": dict(enableGamma=1),
        "hdr": dict(enableHDR=1, exposure=1.7),
        "bloom": dict(enableBloom=1),
        "bloom_hdr": dict(enableBloom=1, enableHDR=1, exposure=0.6),
        "fxaa": dict(enableFXAA=1),
        "all": dict(enableBloom=1, enableHDR=1, enableFXAA=1, exposure=0.8),
    }
    for name, kw in cases.items():
        out = run_post.post_process(frag, bright, **kw)
        extra = {}
        if kw.get("enableFXAA"):
            stage = run_post.post_process(frag, bright, **{**kw, "enableFXAA": 0}) if len(kw) > 1 else frag
            extra["tie"] = run_post.fxaa_tie_mask(stage)
            extra["rgba_u8src"] = run_post.post_process(frag, bright, fxaa_source="u8", **kw)
        ps = np.array([kw.get("enableFXAA", 0), kw.get("enableGamma", 0), kw.get("enableHDR", 0), kw.get("enableBloom", 0)], np.int32)
        np.savez_compressed(os.path.join(OUT, f"post_{name}.npz"), frag=frag, bright=bright, flags=ps,
                            exposure=np.float32(kw.get("exposure", 1.0)), rgba=out, **extra)
        print("post", name, out.shape, {k: float(v.mean()) for k, v in extra.items() if k == "tie"})


def softshadow_cases():
    """Soft shadows with the shader's unset `r.d` given the UB1 value in the harness (essl_adapt.define_ub1)."""
    W, H = 64, 48
    prims = tg.all_primitives_scene(W, H)
    WB = abi.RM_FEAT_WHITE_BACKGROUND
    frame_case("prims_a_softshadow_ub1", subset(prims, range(0, 6)), abi.default_settings(features=WB, enableSoftShadow=1), W, H, ub1=True)
    frame_case("prims_softshadow_ao_bump_ub1", subset(prims, [0, 2, 4, 6, 8, 9]),
               abi.default_settings(enableSoftShadow=1, enableAmbientOcclusion=1), W, H, ub1=True)
    scene, s, res = tg.resource_case("area_light_soft_bump", 64, 40)
    frame_case("res_area_light_soft_bump_ub1", scene, s, 64, 40, ub1=True, **res)


def resource_cases(only=None):
    """Night sky, sea, sky box and area lights: the reference shader with its noise / cube-map / LTC samplers bound to
    the synthetic inputs of tests/test_gpu_parity.py."""
    W, H = 64, 40
    for name in tg.RESOURCE_CASES:
        if only and name not in only:
            continue
        scene, s, res = tg.resource_case(name, W, H)
        if s.enableSoftShadow or (s.features & abi.RM_FEAT_CLOUD):
            continue  # UB1 (soft shadow reads an unset variable) / UB10 (cloud density sample): covered by the env cases
        frame_case("res_" + name, scene, s, W, H, **res)
    if only:
        return
    # function-level probes of the sea and the night sky
    rng = np.random.default_rng(3)
    scene, s = tg.sea_scene(8, 8), abi.default_settings()
    pts = rng.uniform(-30, 30, (4096, 3)).astype(np.float32)
    noise = tg.synthetic_noise()
    for kind in ("sea", "moon", "sinhash"):
        out = run_ref.probe(kind, scene, s, pts, noise=noise)
        np.savez_compressed(os.path.join(OUT, f"probe_{kind}.npz"), kind=kind, pts=pts, out=out, uses_noise=1, **pack(scene, s))
        print("probe", kind, out.shape)


# ---- scenefile → pixels: the BASELINE.json scenefiles through the REFERENCE'S OWN loader (oracle/_ref/dump_tables, the
# unmodified sceneparser / scenefilereader / camera sources) and then through the reference shader on SwiftShader.
REF_SCENES = "/root/reference/scenefiles"
SCENEFILE_CASES = {
    # name: (scenefile, W, H, settings overrides, harness flags)
    "c1_unit_sphere_64steps": ("simple/unit_sphere.json", 64, 64, dict(maxSteps=64), {}),
    "unit_sphere_defaults": ("simple/unit_sphere.json", 96, 54, {}, {}),
    "c2_directional_light_2_soft_ao_ub1": ("lighting/directional_light_2.json", 96, 54, dict(enableSoftShadow=1, enableAmbientOcclusion=1), dict(ub1=True)),
    "directional_light_2_defaults": ("lighting/directional_light_2.json", 96, 54, {}, {}),
    "c3_unit_mandelbulb_12iters": ("simple/unit_mandelbulb.json", 96, 54, dict(fractalIters=12), {}),
    "unit_mandelbulb_defaults": ("simple/unit_mandelbulb.json", 96, 54, {}, {}),
    "c5_unit_mengersponge_l5_refl2": ("simple/unit_mengersponge.json", 96, 54, dict(mengerLevels=5, numReflection=2, enableReflection=1), {}),
    "unit_mengersponge_defaults": ("simple/unit_mengersponge.json", 96, 54, {}, {}),
    # BASELINE configuration 4: volumetric.json as the file is, with the TERRAIN / CLOUD / SKY_BACKGROUND layers compiled in
    # (`nnd` given the UB10 value) — and the terrain + sky layers alone, which need no such edit
    "c4_volumetric_terrain_cloud_sky_ub10": ("simple/volumetric.json", 96, 54, dict(features=tg.ENV_ALL), dict(ub10=True)),
    "c4_volumetric_terrain_sky": ("simple/volumetric.json", 96, 54,
                                  dict(features=abi.RM_FEAT_SKY_BACKGROUND | abi.RM_FEAT_TERRAIN), {}),
}
# The sweep: every other scenefile of the reference that needs no image asset beyond blackmarble.png, no LTC table (area
# lights: lighting/bloom.json, arealight.json, simple/unit_plane.json) and no sky-box, with the reference's default settings
# (reflection scenes: reflection switched on, which is what they are for).
_SWEEP = {
    "lighting": ["directional_light_1", "point_light_1", "point_light_2", "simple_shadow", "spot_light_1", "spot_light_2"],
    "lighting+reflect": ["reflections_basic", "reflections_complex", "test_reflectiveness"],
    "simple": ["blank", "parse_matrix", "phong_total", "unit_capsule", "unit_cone", "unit_cube", "unit_cylinder", "unit_deathstar",
               "unit_mandelbrot", "unit_octa", "unit_sierpinski", "unit_torus"],
}
# ... and the single-texture scenes whose image is small enough to keep as a fixture (board.png 8-bit grey, topleft.png /
# mandril.png / marsTexture.png RGB, colorGrid.png RGBA: the product's PNG reader against QImage's conversions)
_SWEEP["textures_tests"] = ["texture_cone", "texture_cone2", "texture_cube", "texture_cube2", "texture_cube_sample", "texture_cyl",
                            "texture_cyl2", "texture_cyl3", "texture_sphere", "texture_sphere2"]
_SWEEP["simple"] += ["recursive_sphere_2"]  # recursive_sphere_3: more uniforms than SwiftShader links
# ... and a second pass over the scenes whose materials are transparent / reflective or whose lights cast visible shadows,
# with EVERY shading option on: soft shadows (r.d given the UB1 value), ambient occlusion, reflection, refraction
_FULL = dict(enableSoftShadow=1, enableAmbientOcclusion=1, enableReflection=1, enableRefraction=1)
for _rel in ("simple/unit_capsule", "simple/unit_cone", "simple/unit_cylinder", "simple/unit_deathstar", "simple/unit_octa",
             "simple/unit_torus", "simple/unit_cube", "simple/phong_total", "lighting/point_light_2", "lighting/spot_light_2",
             "lighting/simple_shadow", "lighting/reflections_basic"):
    SCENEFILE_CASES[f"sweepfull_{_rel.split('/')[1]}_ub1"] = (_rel + ".json", 64, 36, _FULL, dict(ub1=True))
# ... and the area-light scenes (the LTC tables travel in the fixture as the 8-bit textures the reference uploads);
# lighting/arealight.json also has the blackmarble floor
_SWEEP["lighting"] += ["bloom", "arealight", "depth_of_field", "shadow_test"]  # shadow_test: two images
# (simple/unit_plane.json, the third area-light scene: SwiftShader did not finish it in 30 minutes)
for _grp, _names in _SWEEP.items():
    for _n in _names:
        SCENEFILE_CASES[f"sweep_{_n}"] = (f"{_grp.split('+')[0]}/{_n}.json", 64, 36,
                                          dict(enableReflection=1) if _grp.endswith("+reflect") else {}, {})
# Round 3: the reference's refraction scene that is defined (lighting/refract2.json: four glass spheres over a checker floor,
# reflection + refraction on — refract1.json textures an octahedron, for which the shader has no uv map: its `uv` is read
# unset, frag:1746-1781), its HDR scene (three textured and four plain cubes under three point lights), the five-texture scene
# of textures_tests (one of the images a GIF) and the sky-box scene (cubemap/beach.json with the reference's own JPEG faces,
# texture_store/cube_map/beach, loaded as initCubeMap does: RGBA8888, mirrored; realtimerender.cpp:557-590)
SCENEFILE_CASES["sweep_unit_plane"] = ("simple/unit_plane.json", 48, 27, {}, {})  # the third area-light scene (SwiftShader needs over an hour)
SCENEFILE_CASES["sweep_refract2"] = ("lighting/refract2.json", 64, 36, dict(enableReflection=1, enableRefraction=1), {})
SCENEFILE_CASES["sweep_hdr"] = ("lighting/hdr.json", 64, 36, {}, {})
SCENEFILE_CASES["sweep_directional_light_textured"] = ("textures_tests/directional_light_textured.json", 64, 36, {}, {})
SCENEFILE_CASES["sweep_beach"] = ("cubemap/beach.json", 64, 36, dict(enableReflection=1, enableRefraction=1, enableSkyBox=1), dict(skybox="beach"))


def reference_tables(rel, W, H):
    """ABI tables built from what the reference's own loader + camera produce for scenefile `rel` (realtimerender.cpp:596-811
    uploads exactly these): nothing of the product's loader is involved."""
    import json
    import subprocess
    binp = os.path.join(ROOT, "oracle", "_ref", "dump_tables")
    env = dict(os.environ, LD_LIBRARY_PATH="/usr/lib/x86_64-linux-gnu:/opt/conda/lib")
    out = subprocess.run([binp, os.path.join(REF_SCENES, rel), str(W), str(H)], env=env, capture_output=True, text=True, check=True).stdout
    t = json.loads([l for l in out.splitlines() if l.startswith("@@JSON ")][-1][7:])
    assert t["ok"]
    cam = abi.RmCamera()
    for i in range(16):
        cam.invProjView[i] = t["invProjView"][i]
    cam.initialFar = 100.0  # settings.farPlane, mainwindow.cpp:130
    for i in range(4):
        cam.eyePosition[i] = t["camPos"][i]
    objs = (abi.RmObject * max(len(t["objects"]), 1))()
    tex_files = {}
    for i, o in enumerate(t["objects"]):
        d = objs[i]
        d.type = o["type"]
        for k in range(16):
            d.invModel[k] = o["invModel"][k]
        d.scaleFactor, d.shininess, d.blend, d.ior = o["scaleFactor"], o["shininess"], o["blend"], o["ior"]
        for k in range(3):
            d.cAmbient[k], d.cDiffuse[k], d.cSpecular[k] = o["cAmbient"][k], o["cDiffuse"][k], o["cSpecular"][k]
            d.cReflective[k], d.cTransparent[k] = o["cReflective"][k], o["cTransparent"][k]
        d.texLoc, d.lightIdx = -1, -1
        if o["textured"]:  # configureShapesUniforms: one texture unit per FILE NAME, in first-use order (realtimerender.cpp:735-806)
            d.texLoc, d.repeatU, d.repeatV = tex_files.setdefault(o["textureFile"], len(tex_files)), o["repeatU"], o["repeatV"]
    lights = (abi.RmLight * max(len(t["lights"]), 1))()
    n_objs = len(t["objects"])
    area = [i for i, l in enumerate(t["lights"]) if l["type"] == abi.RM_LIGHT_AREA]
    if area:  # RayMarchScene::initScene appends one emissive RECTANGLE per area light (raymarchscene.cpp:126-133, raymarchobj.h:16-23)
        grown = (abi.RmObject * (n_objs + len(area)))()
        for i in range(n_objs):
            grown[i] = objs[i]
        objs = grown
    for i, l in enumerate(t["lights"]):
        d = lights[i]
        d.type, d.angle, d.penumbra = l["type"], l["angle"], l["penumbra"]
        for k in range(3):
            d.color[k], d.dir[k], d.pos[k], d.func[k] = l["color"][k], l["dir"][k], l["pos"][k], l["func"][k]
        if l["type"] == abi.RM_LIGHT_AREA:
            # configureLightsUniforms, realtimerender.cpp:682-693: intensity, twoSided = true, points[k] = ctm · corner k
            # (realtime.h:136-141: tl, tr, br, bl of the unit square), glm's mat4·vec4 in binary32: (m0·x + m1·y) + (m2·z + m3·w)
            d.intensity, d.twoSided = l["intensity"], 1
            m = np.array(l["ctm"], dtype=np.float32).reshape(4, 4)  # m[c] = column c
            for k, (cx, cy) in enumerate(((-0.5, 0.5), (0.5, 0.5), (0.5, -0.5), (-0.5, -0.5))):
                v = (m[0] * np.float32(cx) + m[1] * np.float32(cy)) + (m[2] * np.float32(0.0) + m[3] * np.float32(1.0))
                for j in range(3):
                    d.points[k][j] = float(v[j])
            o = objs[n_objs + area.index(i)]
            o.type, o.scaleFactor, o.texLoc, o.isEmissive, o.lightIdx = abi.RM_RECTANGLE, 1.0, -1, 1, i
            for k in range(16):
                o.invModel[k] = l["ctmInv"][k]
            for k in range(3):
                o.color[k] = l["color"][k]
    n_objs += len(area)
    g = abi.RmGlobals(t["ka"], t["kd"], t["ks"], t["kt"], 8.0)  # power 8, juliaSeed 0, iTime 0: settings.h defaults
    return (cam, objs, n_objs, lights, len(t["lights"]), g), list(tex_files)


def scenefile_cases(only=None):
    from PIL import Image
    for name, (rel, W, H, over, flags) in SCENEFILE_CASES.items():
        if only and name not in only:
            continue
        scene, tex_files = reference_tables(rel, W, H)
        s = abi.default_settings(**over)
        tex = None
        if tex_files:  # QImage::load + convertToFormat(RGBA8888) + mirrored() (raymarchscene.cpp:198-209), decoded here with PIL
            tex = [np.ascontiguousarray(np.asarray(Image.open(f).convert("RGBA"))[::-1]) for f in tex_files]
            tex = tex[0] if len(tex) == 1 else tex  # several: a selection chain over the units (essl_adapt E4)
        extra = {}
        if any(scene[3][i].type == abi.RM_LIGHT_AREA for i in range(scene[4])):
            # the LTC tables as the reference uploads them: 64×64 RGBA floats into a GL_RGBA (8-bit unorm) texture
            # (loadMTexture / loadLTUTexture, realtimerender.cpp:896-930): dumped from the reference's header by
            # oracle/_ref/dump_tables --ltc, quantised as the GL does (rmo_ltc_quantise)
            import subprocess
            import tempfile
            with tempfile.NamedTemporaryFile(suffix=".bin") as f:
                subprocess.run([os.path.join(ROOT, "oracle", "_ref", "dump_tables"), "--ltc", f.name], check=True,
                               env=dict(os.environ, LD_LIBRARY_PATH="/usr/lib/x86_64-linux-gnu:/opt/conda/lib"))
                raw = np.fromfile(f.name, dtype=np.float32).reshape(2, 64, 64, 4)
            extra = {"ltc1": h.oracle_ltc_quantise(raw[0]), "ltc2": h.oracle_ltc_quantise(raw[1])}
            flags = dict(flags, ltc=(extra["ltc1"], extra["ltc2"]))
        if "skybox" in flags:  # six faces +x, -x, +y, -y, +z, -z (raymarchscene.cpp:50-86), decoded here with PIL (libjpeg, as Qt does)
            base = os.path.join(REF_SCENES, "texture_store", "cube_map", flags["skybox"])
            ext = ".jpg" if os.path.exists(os.path.join(base, "+x.jpg")) else ".png"
            faces = [np.ascontiguousarray(np.asarray(Image.open(os.path.join(base, f + ext)).convert("RGBA"))[::-1]) for f in ("+x", "-x", "+y", "-y", "+z", "-z")]
            extra["cubemap"] = {"beach": 1, "night": 2, "island": 3}[flags["skybox"]]  # the product loads the faces itself (rm_skybox_face_path)
            flags = dict(flags, skybox=faces)
        rgba, bright = run_ref.render(scene, s, W, H, tex, **flags)
        flags = {k: v for k, v in flags.items() if k not in ("ltc", "skybox")}
        # what saveViewportImage writes (realtime.cpp:284-350): clamp, ×255, round, rows top-down
        png8 = (np.clip(rgba[::-1], 0, 1) * 255.0 + 0.5).astype(np.uint8)
        np.savez_compressed(os.path.join(OUT, f"scenefile_{name}.npz"), W=W, H=H, rgba=rgba, bright=bright, png8=png8,
                            scenefile=rel, **{k: int(v) for k, v in flags.items()}, **extra, **pack(scene, s))
        print("scenefile", name, rgba.shape, float(np.nanmean(rgba[..., :3])))


def main():
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "scenefile":
        return scenefile_cases(sys.argv[2:])
    if len(sys.argv) > 1 and sys.argv[1] == "soft":
        return softshadow_cases()
    if len(sys.argv) > 1 and sys.argv[1] == "res":
        return resource_cases(sys.argv[2:])
    if len(sys.argv) > 1 and sys.argv[1] == "post":
        return post_cases()
    if len(sys.argv) > 1 and sys.argv[1] == "env":
        return env_cases()
    if len(sys.argv) > 1 and sys.argv[1] == "tex":
        return texture_cases()
    scenefile_cases()
    texture_cases()
    env_cases()
    post_cases()
    resource_cases()
    softshadow_cases()
    W, H = 64, 48
    WB, DB = abi.RM_FEAT_WHITE_BACKGROUND, abi.RM_FEAT_DARK_BACKGROUND
    prims = tg.all_primitives_scene(W, H)
    frame_case("prims_a_phong", subset(prims, range(0, 6)), abi.default_settings(features=WB), W, H)
    frame_case("prims_b_phong", subset(prims, [6, 7, 8, 9]), abi.default_settings(features=WB, maxSteps=128), W, H)
    frame_case("prims_a_bump", subset(prims, range(0, 6)), abi.default_settings(), W, H)
    frame_case("prims_ao", subset(prims, [0, 2, 4, 6, 8, 9]), abi.default_settings(features=DB, enableAmbientOcclusion=1), W, H)
    frame_case("reflect_refract", tg.reflect_refract_scene(W, H),
               abi.default_settings(features=WB, enableReflection=1, enableRefraction=1), W, H)
    frame_case("menger_reflect", tg.menger_scene(W, H), abi.default_settings(features=WB, enableReflection=1), W, H)
    frame_case("bulb_reference_consts", h.scene_mandelbulb(96, 54), abi.default_settings(), 96, 54)
    frame_case("bulb_12iters_nobump", h.scene_mandelbulb(96, 54), abi.default_settings(features=WB, fractalIters=12), 96, 54)
    two_d = h.scene_mandelbulb(W, H)[:5] + (h.make_globals(two_d=1, itime=3.0),)
    frame_case("mandelbrot_2d", two_d, abi.default_settings(), W, H)
    # function-level probes
    rng = np.random.default_rng(42)
    s = abi.default_settings()
    bulb = h.scene_mandelbulb(8, 8)
    probe_case("sd_bulb_p8", "sdscene", bulb, s, rng.normal(0, 0.8, (4096, 3)))
    probe_case("sd_bulb_p8_12iters", "sdscene", bulb, abi.default_settings(fractalIters=12), rng.normal(0, 0.8, (2048, 3)))
    julia = bulb[:5] + (h.make_globals(julia=(0.35, -0.2), power=6.0),)
    probe_case("sd_julia_p6", "sdscene", julia, s, rng.normal(0, 0.8, (2048, 3)))
    probe_case("sd_menger", "sdscene", tg.menger_scene(8, 8), s, rng.uniform(-1.5, 1.5, (4096, 3)))
    pts = rng.uniform(-3.5, 3.5, (4096, 3))
    pts[:, 2] *= 0.4
    probe_case("sd_prims_a", "sdscene", subset(prims, range(0, 6)), s, pts)
    probe_case("sd_prims_b", "sdscene", subset(prims, [6, 7, 8, 9]), s, pts)
    probe_case("pnoise", "pnoise", bulb, s, rng.uniform(-40, 40, (8192, 3)))
    probe_case("normal_prims_a", "normal", subset(prims, range(0, 6)), s, pts[:2048])


if __name__ == "__main__":
    main()
