#!/usr/bin/env python3
"""How many frames per second can ONE rank's host path submit?  At N = 8 a 4K bulb frame leaves 0.33 ms per frame for 6x scaling,
i.e. ≈3000 submits per second per rank through bench.py's path: FramePipeline.submit → rm_render_tiles (ctypes) → gather → (rank 0)
rm_deinterleave.  One GPU box: the shard is the real 1/8 shard of the frame (rank 0's tiles, the most rows), the gather is replaced
by a device copy into slot 0 of the gather buffer (the RCCL call itself is exercised by tests/test_gpu_bench_dist.py), the
de-interleave is the real whole-frame kernel.  Prints wall ms per frame for the pipeline and for its parts.
Usage: python scripts/submit_rate_probe.py [c3|c5]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class _Work:
    def wait(self):
        return True


def main():
    import torch
    import torch.distributed as dist
    import bench
    from raymarcher_amd import Renderer
    from raymarcher_amd.dist import FramePipeline, ShardPlan
    from raymarcher_amd import lib
    cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
    relief = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    N = 8
    r = Renderer(0)
    tables, settings, W, H, _ = bench.build_config(cfg)
    assert lib().rm_set_root_relief(relief) == 0
    plan = ShardPlan(H, bench.TILE_ROWS, N)
    my_rows, slot_rows = plan.rows(0), plan.slot_rows

    def fake_gather(local, outs, dst=0, async_op=False):  # rank 0's own slot travels by a device copy; peers' slots keep old data
        if outs is not None:
            outs[0].copy_(local, non_blocking=True)
        return _Work()
    dist.gather = fake_gather
    if relief:  # what a PEER has to sustain under the relief partition (rank 1 owns the most rows; no de-interleave, no receives)
        rows1 = plan.rows(1)
        pipe1 = FramePipeline(plan, 1, (W, 4), torch.float32, r.device, depth=3, multi_stream=True)
        sub1 = lambda: pipe1.submit(lambda slot: r.render_tiles(tables, settings, W, H, bench.TILE_ROWS, 1, N, out=slot[:rows1]))
        for _ in range(30):
            sub1()
        pipe1.drain(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):
            sub1()
        pipe1.drain(); torch.cuda.synchronize()
        print(f"{cfg}, root relief {relief}: rank 1's shard ({rows1} of {H} rows) through FramePipeline: wall {(time.perf_counter() - t0) / 300 * 1e3:.3f} ms per frame")
    frame = {}
    pipe = FramePipeline(plan, 0, (W, 4), torch.float32, r.device, depth=3, multi_stream=True,
                         finish=lambda g: frame.__setitem__("f", r.deinterleave(g, W, H, bench.TILE_ROWS, N, slot_rows)))

    def run(n, what):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            what()
        t_host = time.perf_counter() - t0
        pipe.drain()
        torch.cuda.synchronize()
        return t_host / n * 1e3, (time.perf_counter() - t0) / n * 1e3

    submit = lambda: pipe.submit(lambda slot: r.render_tiles(tables, settings, W, H, bench.TILE_ROWS, 0, N, out=slot[:my_rows]))
    run(30, submit)
    host, wall = run(300, submit)
    print(f"{cfg}, root relief {relief}: rank 0's shard ({my_rows} of {H} rows) through FramePipeline, three frames in flight: host {host:.3f} ms per submit "
          f"({1e3 / host:.0f} submits/s), wall {wall:.3f} ms per frame; budget for 6x at N = {N}: "
          f"{bench_ms(cfg) / 6:.3f} ms")
    # the pure host cost of a submit: the same pipeline over a frame so small that the GPU never holds the host back
    w2, h2 = 256, 128
    plan2 = ShardPlan(h2, bench.TILE_ROWS, N)
    pipe2 = FramePipeline(plan2, 0, (w2, 4), torch.float32, r.device, depth=3, multi_stream=True,
                          finish=lambda g: frame.__setitem__("f", r.deinterleave(g, w2, h2, bench.TILE_ROWS, N, plan2.slot_rows)))
    rows2 = plan2.rows(0)
    sub2 = lambda: pipe2.submit(lambda slot: r.render_tiles(tables, settings, w2, h2, bench.TILE_ROWS, 0, N, out=slot[:rows2]))
    for _ in range(30):
        sub2()
    pipe2.drain(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(1000):
        sub2()
    th = (time.perf_counter() - t0) / 1000 * 1e3
    pipe2.drain(); torch.cuda.synchronize()
    print(f"  host path alone (a {w2}x{h2} frame, GPU idle most of the time): {th:.3f} ms per submit = {1e3 / th:.0f} submits/s")
    buf = torch.empty((slot_rows, W, 4), dtype=torch.float32, device=r.device)
    host, wall = run(300, lambda: r.render_tiles(tables, settings, W, H, bench.TILE_ROWS, 0, N, out=buf[:my_rows]))
    print(f"  render_tiles alone (one stream): host {host:.3f} ms per call, wall {wall:.3f} ms per frame")
    g = pipe.gathered[0]
    host, wall = run(300, lambda: r.deinterleave(g, W, H, bench.TILE_ROWS, N, slot_rows))
    print(f"  rm_deinterleave of the whole frame alone: host {host:.3f} ms per call, wall {wall:.3f} ms")


def bench_ms(cfg):
    return {"c3": 2.02, "c5": 21.2, "c2": 0.90, "c4": 2.62, "c1": 0.14}[cfg]


if __name__ == "__main__":
    main()
