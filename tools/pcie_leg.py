#!/usr/bin/env python3
"""What a caller who keeps its batches in HOST memory would add per cfg2 step: pinned H2D of (mixed, lips) and D2H of
(masks, separated).  Never part of bench.py's `value` (the boundary takes device pointers)."""
import time, torch
dev = torch.device("cuda:0")
B = 32
mixed_h = torch.rand(B, 257, 63).pin_memory(); lips_h = torch.rand(B, 50, 32, 32).pin_memory()
out_h = torch.empty(2, B, 63, 2, 257).pin_memory()
mixed_d = torch.empty_like(mixed_h, device=dev); lips_d = torch.empty_like(lips_h, device=dev); out_d = torch.empty_like(out_h, device=dev)
def leg():
    mixed_d.copy_(mixed_h, non_blocking=True); lips_d.copy_(lips_h, non_blocking=True); out_h.copy_(out_d, non_blocking=True)
for _ in range(5): leg()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): leg()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
inb = (mixed_h.numel() + lips_h.numel()) * 4; outb = out_h.numel() * 4
print(f"H2D {inb/1e6:.1f} MB + D2H {outb/1e6:.1f} MB per step: {dt*1e3:.3f} ms ({(inb+outb)/dt/1e9:.1f} GB/s) -> serial with the 0.45 ms step: {32/(dt+0.00045):.0f} clips/s")
