#!/usr/bin/env python3
"""A/B of the light split (rm_kernels.hip: the heaviest tiles of a settled picture one light per workgroup) on the reference's
lighting scenefiles with several lights: kernel ms of the settled frame with the split off / forced at 1/256 of the tiles / left to
the launcher's own measurement (the default), hard and soft shadows, 1080p and 4K.  GPU box only.   python scripts/light_split_probe.py [out.md]"""
import ctypes as C
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from raymarcher_amd import Renderer, Scene, abi, lib  # noqa: E402


def main():
    r = Renderer(0)
    L = lib()
    S = os.path.join(ROOT, "tests", "golden", "scenes", "lighting")
    rows = ["| scene | size | shadows | split off, ms | forced, 1/256 of the tiles | measured by the launcher (default) | tiles split by default |", "|---|---|---|---|---|---|---|"]
    for name in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("directional_light_2", "point_light_2", "spot_light_2", "hdr", "depth_of_field", "shadow_test")):
        for W, H in ((1920, 1080), (3840, 2160)):
            for soft in (0, 1):
                t = Scene(path=os.path.join(S, name + ".json")).tables(W, H, load_textures=False)
                for k in range(t.num_objects):
                    t.objects[k].texLoc = -1  # no textures here: the plain table-walk class
                s = abi.default_settings(enableSoftShadow=soft, enableAmbientOcclusion=soft)
                out = torch.empty((H, W, 4), dtype=torch.float32, device=r.device)
                ms, ref, split = [], None, 0
                for div in (0, 256, -1):
                    L.rm_debug_set_light_split(div)
                    for _ in range(28):  # tile-shape tuner (8 frames) + settling (4) + the split's own measurement (4) + margin
                        r.render(t, s, W, H, out=out)
                    torch.cuda.synchronize()
                    if ref is None:
                        ref = out.clone()
                    assert bool((out.view(torch.int32) == ref.view(torch.int32)).all()), (name, W, soft, div)
                    L.rm_set_timing(1)
                    for _ in range(20):
                        r.render(t, s, W, H, out=out)
                    torch.cuda.synchronize()
                    k, n = C.c_double(), C.c_int()
                    L.rm_get_timing(C.byref(k), C.byref(n))
                    L.rm_set_timing(0)
                    ms.append(k.value)
                    if div == -1:
                        split = L.rm_debug_last_split()
                rows.append(f"| {name} | {W}x{H} | {'soft + AO' if soft else 'hard'} | {ms[0]:.3f} | {ms[1]:.3f} ({(ms[1] / ms[0] - 1) * 100:+.1f} %) | "
                            f"{ms[2]:.3f} ({(ms[2] / ms[0] - 1) * 100:+.1f} %) | {split} |")
                print(rows[-1], flush=True)
    L.rm_debug_set_light_split(-1)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write("\n".join(rows) + "\n")


if __name__ == "__main__":
    main()
