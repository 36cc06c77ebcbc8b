import sys, os, ctypes as C, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from raymarcher_amd import Renderer, Scene, abi, lib
r = Renderer(0); L = lib()
S = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests", "golden", "scenes")
W, H = 7680, 4320
t = Scene(path=os.path.join(S, "simple", "unit_mengersponge.json")).tables(W, H)
s = abi.default_settings(mengerLevels=5, numReflection=2, enableReflection=1)
for N in (8, 4, 2):
    for path in (1, 5):
        L.rm_set_kernel_path(path)
        outs = [r.render_tiles(t, s, W, H, 8, 3 % N, N) for _ in range(2)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): r.render_tiles(t, s, W, H, 8, 3 % N, N, out=outs[0])
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        print(f"C5 shard 1/{N} ({outs[0].shape[0] * W / 1e6:.1f} Mpx) path {path} (ran {L.rm_debug_last_path()}): {ms:.2f} ms")
L.rm_set_kernel_path(0)
