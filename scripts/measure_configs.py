#!/usr/bin/env python3
"""Time every BASELINE.json configuration on one GPU (whole frame, one launch, HIP-event kernel time) — informative
numbers for DESIGN.md / profiles/, not bench lines.  Usage: python scripts/measure_configs.py [out.md]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import test_gpu_parity as tg
    from raymarcher_amd import Renderer, Scene, abi, lib, scenes
    r = Renderer(0)
    L = lib()
    S = tg.SCENES
    cases = []
    t = Scene(path=os.path.join(S, "simple", "unit_sphere.json")).tables(256, 256, load_textures=False)
    t.textures = [tg.synthetic_textures()[1]]
    cases.append(("C1 unit_sphere 256x256, 64 steps, Phong", t, abi.default_settings(maxSteps=64), 256, 256))
    t = Scene(path=os.path.join(S, "simple", "unit_sphere.json")).tables(3840, 2160, load_textures=False)
    t.textures = [tg.synthetic_textures()[1]]
    cases.append(("C1@4K unit_sphere.json (textured floor) 3840x2160, 256 steps", t, abi.default_settings(), 3840, 2160))
    sk = tg.resource_case("skybox_reflect", 3840, 2160)
    tk = tg.tables_of(sk[0])
    tk.skybox = sk[2]["skybox"]
    cases.append(("SKY sky box behind reflection + refraction 3840x2160", tk, sk[1], 3840, 2160))
    t = Scene(path=os.path.join(S, "lighting", "directional_light_2.json")).tables(1920, 1080)
    cases.append(("C2 directional_light_2 1920x1080, soft shadow + AO", t, abi.default_settings(enableSoftShadow=1, enableAmbientOcclusion=1), 1920, 1080))
    t = Scene(path=os.path.join(S, "lighting", "directional_light_2.json")).tables(3840, 2160)
    cases.append(("C2@4K the same scene and options at 3840x2160", t, abi.default_settings(enableSoftShadow=1, enableAmbientOcclusion=1), 3840, 2160))
    t = Scene(path=os.path.join(S, "lighting", "reflections_complex.json")).tables(3840, 2160)
    cases.append(("RC reflections_complex.json 3840x2160, reflection (1 bounce) + Perlin bump", t, abi.default_settings(enableReflection=1), 3840, 2160))
    cases.append(("C3 Mandelbulb p8 12 iters 3840x2160 (headline)", scenes.mandelbulb(3840, 2160), abi.default_settings(fractalIters=12), 3840, 2160))
    cases.append(("C3' same, RM_FEAT_BULB_POWER8_ALGEBRAIC", scenes.mandelbulb(3840, 2160),
                  abi.default_settings(fractalIters=12, features=abi.RM_FEAT_REFERENCE_DEFAULT | abi.RM_FEAT_BULB_POWER8_ALGEBRAIC), 3840, 2160))
    cases.append(("C3'' Mandelbulb p8 20 iters (reference constant) 3840x2160", scenes.mandelbulb(3840, 2160), abi.default_settings(), 3840, 2160))
    t = Scene(path=os.path.join(S, "simple", "volumetric.json")).tables(3840, 2160, far=2000.0)
    cases.append(("C4 volumetric.json as is (camera below the terrain, looking down) + terrain + cloud + sky 3840x2160", t,
                  abi.default_settings(features=tg.ENV_ALL), 3840, 2160))
    t = Scene(path=os.path.join(S, "simple", "volumetric.json")).tables(3840, 2160, far=2000.0)
    t.camera = tg.env_scene(3840, 2160)[0]
    cases.append(("C4 same scene, camera turned to the horizon: terrain + cloud + sky 3840x2160 (1 GPU)", t, abi.default_settings(features=tg.ENV_ALL), 3840, 2160))
    cases.append(("C5 unit_mengersponge.json, 5 levels, reflection 2 bounces 7680x4320 (1 GPU)",
                  Scene(path=os.path.join(S, "simple", "unit_mengersponge.json")).tables(7680, 4320),
                  abi.default_settings(mengerLevels=5, numReflection=2, enableReflection=1), 7680, 4320))
    for tag, (w, hh) in (("C5@4K", (3840, 2160)), ("C5@1080p", (1920, 1080))):
        cases.append((f"{tag} unit_mengersponge.json, 5 levels, reflection 2 bounces {w}x{hh}",
                      Scene(path=os.path.join(S, "simple", "unit_mengersponge.json")).tables(w, hh),
                      abi.default_settings(mengerLevels=5, numReflection=2, enableReflection=1), w, hh))
    cases.append(("RC@1080p reflections_complex.json 1920x1080, reflection + Perlin bump",
                  Scene(path=os.path.join(S, "lighting", "reflections_complex.json")).tables(1920, 1080),
                  abi.default_settings(enableReflection=1), 1920, 1080))
    cases.append(("RC2 reflections_complex.json 3840x2160, reflection 2 bounces + Perlin bump",
                  Scene(path=os.path.join(S, "lighting", "reflections_complex.json")).tables(3840, 2160),
                  abi.default_settings(enableReflection=1, numReflection=2), 3840, 2160))
    sea = tg.resource_case("sea_sky", 3840, 2160)
    ts = tg.tables_of(sea[0])
    ts.noise = sea[2]["noise"]
    cases.append(("sea + sky + reflective sphere 3840x2160", ts, sea[1], 3840, 2160))
    al = tg.resource_case("area_light", 1920, 1080)
    ta = tg.tables_of(al[0])
    ta.ltc1, ta.ltc2 = al[2]["ltc1"], al[2]["ltc2"]
    cases.append(("area light (LTC) + point light, reflection 1920x1080", ta, al[1], 1920, 1080))
    only = os.environ.get("RM_ONLY")  # e.g. RM_ONLY=C5: a single configuration (PMC passes profile one kernel at a time)
    if only:
        cases = [c for c in cases if any(c[0].startswith(o + " ") for o in only.split(","))]
    rows = ["| configuration | kernel ms | Mpixels/s | sceneEvals (reference / executed) |", "|---|---|---|---|"]
    for name, t, s, W, H in cases:
        out = torch.empty((H, W, 4), dtype=torch.float32, device=r.device)
        for _ in range(9):  # the tile-shape tuner's eight frames …
            r.render(t, s, W, H, out=out)
        torch.cuda.synchronize()
        for _ in range(3):  # … its decision (the timings are in now) and the first frames of its choice
            r.render(t, s, W, H, out=out)
        torch.cuda.synchronize()
        L.rm_set_timing(1)
        n = 10
        for _ in range(n):
            r.render(t, s, W, H, out=out)
        torch.cuda.synchronize()
        ms, k = C.c_double(), C.c_int()
        L.rm_get_timing(C.byref(ms), C.byref(k))
        L.rm_set_timing(0)
        ev = ""
        if not os.environ.get("RM_NO_COUNT"):
            try:
                _, c1 = r.render_counted(t, s, W, H, abi.RM_COUNT_REFERENCE)
                _, c2 = r.render_counted(t, s, W, H, abi.RM_COUNT_EXECUTED)
                if c1.sceneEvals:
                    ev = f"{c1.sceneEvals / 1e6:.1f} M / {c2.sceneEvals / 1e6:.1f} M ({c1.sceneEvals / (W * H):.1f} / {c2.sceneEvals / (W * H):.1f} per pixel)"
            except Exception as e:  # counted renders do not take sampler resources
                ev = ""
        rows.append(f"| {name} | {ms.value:.3f} | {W * H / ms.value / 1e3:.1f} | {ev} |")
        print(rows[-1], flush=True)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            f.write("\n".join(rows) + "\n")


if __name__ == "__main__":
    main()
