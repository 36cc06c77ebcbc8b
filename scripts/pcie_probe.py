#!/usr/bin/env python3
"""What handing a finished frame to the HOST costs (never part of bench.py's `value`: the boundary leaves frames in HBM): device →
pinned host copies of a 4K float4 frame (133 MB) and of its RGBA8 conversion (33 MB), HIP-event timed.  GPU box only."""
import torch

def main():
    dev = torch.device("cuda", 0)
    for name, shape, dt in (("float4 3840x2160", (2160, 3840, 4), torch.float32), ("rgba8 3840x2160", (2160, 3840, 4), torch.uint8),
                            ("float4 7680x4320", (4320, 7680, 4), torch.float32), ("rgba8 7680x4320", (4320, 7680, 4), torch.uint8)):
        src = torch.rand(shape, device=dev).to(dt) if dt == torch.float32 else torch.randint(0, 255, shape, device=dev, dtype=dt)
        dst = torch.empty(shape, dtype=dt).pin_memory()
        for _ in range(3):
            dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            dst.copy_(src, non_blocking=True)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        mb = src.numel() * src.element_size() / 1e6
        print(f"| {name} | {mb:.1f} MB | {ms:.3f} ms | {mb / ms:.1f} GB/s |")

if __name__ == "__main__":
    main()
