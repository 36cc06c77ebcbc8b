#!/usr/bin/env python3
"""A frame with no history (first frame, or the first after a change of size): which static launch order of the 8×8 tiles
is best?  Raster (what the launcher does today), centre-out (tiles sorted by distance from the image centre), and a few
others, timed with rm_debug_set_tile_order on several scenes.  GPU box only."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    from raymarcher_amd import Renderer, Scene, abi, lib, scenes
    S = os.path.join(ROOT, "tests", "golden", "scenes")
    r = Renderer(0)
    L = lib()
    W, H = 3840, 2160
    cases = [("bulb 4K", scenes.mandelbulb(W, H), abi.default_settings(fractalIters=12)),
             ("directional_light_2 4K soft+AO", Scene(path=os.path.join(S, "lighting", "directional_light_2.json")).tables(W, H),
              abi.default_settings(enableSoftShadow=1, enableAmbientOcclusion=1)),
             ("reflections_complex 4K", Scene(path=os.path.join(S, "lighting", "reflections_complex.json")).tables(W, H), abi.default_settings(enableReflection=1)),
             ("unit_mengersponge 4K refl 2", Scene(path=os.path.join(S, "simple", "unit_mengersponge.json")).tables(W, H),
              abi.default_settings(mengerLevels=5, numReflection=2, enableReflection=1))]
    tx, ty = (W + 7) // 8, (H + 7) // 8
    n = tx * ty
    ix = torch.arange(n, device=r.device)
    cx, cy = (ix % tx).float() - (tx - 1) / 2, (ix // tx).float() - (ty - 1) / 2
    orders = {"raster (today)": None,
              "centre-out": torch.argsort(cx * cx + cy * cy, stable=True),
              "centre-out, aspect-normalised": torch.argsort((cx / tx) ** 2 + (cy / ty) ** 2, stable=True),
              "outside-in": torch.argsort(-(cx * cx + cy * cy), stable=True),
              "rows from the middle": torch.argsort(cy.abs(), stable=True)}
    L.rm_set_tile_order(0)
    for name, t, s in cases:
        out = torch.empty((H, W, 4), dtype=torch.float32, device=r.device)
        row = [name]
        # round 3: orders from a classification of the tiles a launcher could get from the scene's projected bounds — here the
        # ideal version of it, from the frame itself (which tiles hold object pixels): objects first / background last, and
        # silhouette tiles (object and background pixels) before full-object tiles
        r.render(t, s, W, H, out=out)
        bgc = out[H - 1, 0].clone()
        isbg = (out == bgc).all(dim=-1)  # (H, W)
        pad = torch.ones((ty * 8, tx * 8), dtype=torch.bool, device=r.device)
        pad[:H, :W] = isbg
        tiles = pad.view(ty, 8, tx, 8).permute(0, 2, 1, 3).reshape(n, 64)
        nbg = tiles.sum(dim=1)
        orders["objects first, background last"] = torch.argsort((nbg == 64).to(torch.int32), stable=True)
        orders["silhouette, objects, background"] = torch.argsort(torch.where(nbg == 64, 2, torch.where(nbg > 0, 0, 1)).to(torch.int32), stable=True)
        # round 4: the same classification from GEOMETRY alone (what a launcher could compute before the frame exists): the tile
        # centre's ray against every object's world-space bounding ball — ring (closest approach within [a·R, R + tile footprint]:
        # silhouette candidates) first, ball interiors next, the rest last
        import numpy as np
        inv = np.array(list(t.camera.invProjView), np.float64).reshape(4, 4).T
        ti = np.arange(n)
        ndc = np.stack([((ti % tx) * 8 + 4) / W * 2 - 1, ((ti // tx) * 8 + 4) / H * 2 - 1], -1)
        nr = np.concatenate([ndc, -np.ones((n, 1)), np.ones((n, 1))], -1) @ inv.T
        fr = np.concatenate([ndc, np.ones((n, 1)), np.ones((n, 1))], -1) @ inv.T
        ro = nr[:, :3] / nr[:, 3:]
        rd = fr[:, :3] / fr[:, 3:] - ro
        rd /= np.linalg.norm(rd, axis=1, keepdims=True)
        foot = np.linalg.norm(fr[1, :3] / fr[1, 3] - fr[0, :3] / fr[0, 3]) / np.linalg.norm(fr[0, :3] / fr[0, 3] - ro[0])  # one tile's angle
        for a in (0.5, 0.7):
            cls = np.zeros(n, np.int32)
            for i in range(t.num_objects):
                o = t.objects[i]
                Mi = np.linalg.inv(np.array(list(o.invModel), np.float64).reshape(4, 4).T)
                c = Mi[:3, 3]
                smax = np.linalg.svd(Mi[:3, :3], compute_uv=False)[0]
                R = {10: 1.15, 11: 1.7322}.get(o.type, 0.8662) * smax
                v = c - ro
                tca = (v * rd).sum(1)
                q = np.sqrt(np.maximum((v * v).sum(1) - tca * tca, 0.0))
                m = foot * np.maximum(tca, 0.0)
                ring = (tca > 0) & (q >= a * R) & (q <= R + m)
                inside = (tca > 0) & (q < a * R)
                cls = np.maximum(cls, np.where(ring, 2, np.where(inside, 1, 0)))
            orders[f"geometric rings a={a}"] = torch.argsort(torch.from_numpy(-cls).to(r.device), stable=True)
        for oname, o in orders.items():
            oo = None if o is None else o.to(torch.int32).contiguous()
            L.rm_debug_set_tile_order(C.c_void_p(oo.data_ptr()) if oo is not None else None, None, n if oo is not None else 0)
            for _ in range(2):
                r.render(t, s, W, H, out=out)
            torch.cuda.synchronize()
            L.rm_set_timing(1)
            for _ in range(8):
                r.render(t, s, W, H, out=out)
            torch.cuda.synchronize()
            ms, k = C.c_double(), C.c_int()
            L.rm_get_timing(C.byref(ms), C.byref(k))
            L.rm_set_timing(0)
            row.append(f"{oname} {ms.value:.3f}")
        L.rm_debug_set_tile_order(None, None, 0)
        # and with the feedback
        L.rm_set_tile_order(1)
        for _ in range(3):
            r.render(t, s, W, H, out=out)
        torch.cuda.synchronize()
        L.rm_set_timing(1)
        for _ in range(8):
            r.render(t, s, W, H, out=out)
        torch.cuda.synchronize()
        st = (C.c_double * 4)()
        ms, k = C.c_double(), C.c_int()
        L.rm_get_stage_timing(C.byref(ms), st, C.byref(k))
        L.rm_set_timing(0)
        L.rm_set_tile_order(0)
        row.append(f"feedback {st[1]:.3f} (+ sort {st[0]:.3f})")
        print(" | ".join(row))
    L.rm_set_tile_order(-1)


if __name__ == "__main__":
    main()
