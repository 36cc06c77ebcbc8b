import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import torch, bench
from raymarcher_amd import Renderer, lib
r = Renderer(0)
for cfg in ("c4", "c3", "c1"):
    t, s, W, H, _ = bench.build_config(cfg)
    outs = [torch.empty((H, W, 4), dtype=torch.float32, device=r.device) for _ in range(3)]
    for _ in range(20):
        r.render(t, s, W, H, out=outs[0])
    torch.cuda.synchronize()
    n = 60
    t0 = time.perf_counter()
    for _ in range(n):
        r.render(t, s, W, H, out=outs[0])
    torch.cuda.synchronize()
    one = (time.perf_counter() - t0) / n * 1e3
    streams = [torch.cuda.Stream(device=r.device) for _ in range(3)]
    for k in range(30):
        with torch.cuda.stream(streams[k % 3]):
            r.render(t, s, W, H, out=outs[k % 3])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        with torch.cuda.stream(streams[k % 3]):
            r.render(t, s, W, H, out=outs[k % 3])
    torch.cuda.synchronize()
    three = (time.perf_counter() - t0) / n * 1e3
    print(f"{cfg}: one stream {one:.3f} ms per frame, three in flight {three:.3f} ms ({one / three:.2f} x)", flush=True)
