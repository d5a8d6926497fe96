#!/usr/bin/env python3
"""Developer experiment (GPU box): what period would cross-step pipelining give at cfg2?  Replays the visual encoder
graph on one stream and the audio-encoder + fusion + decoder graphs on another, with no dependency between them,
and reports the steady-state period (the pipelined step would add the K/V hand-off only)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "av-separation-transformer_amd"))
import torch
import av_separation as av
dev = torch.device("cuda:0")
B, F, T, N, H, W, d = 32, 257, 63, 50, 32, 32, 256
torch.manual_seed(0)
m = av.AVSeparationTransformer(dropout=0.0).to(dev).eval()
mixed = torch.rand(B, F, T, device=dev); lips = torch.rand(B, N, H, W, device=dev)
a = torch.randn(B, T, d, device=dev); v = torch.randn(B, T, d, device=dev)

def capture(fn, s):
    with torch.cuda.stream(s), torch.no_grad():
        fn(); fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            fn()
        g.replay(); torch.cuda.synchronize()
    return g

s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
g_vis = capture(lambda: m.visual_encoder(lips, T), s1)
def main_chain():
    x = m.audio_encoder(mixed)
    f = m.fusion(x, v)
    m.decoder(f)
g_main = capture(main_chain, s2)
reps = 200
for name, graphs in (("visual only", [(g_vis, s1)]), ("audio+fusion+decoder only", [(g_main, s2)]),
                     ("both concurrently", [(g_vis, s1), (g_main, s2)])):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for g, s in graphs:
            with torch.cuda.stream(s):
                g.replay()
    torch.cuda.synchronize()
    print(f"{name:28s} {(time.perf_counter() - t0) / reps * 1e6:8.1f} us per step", flush=True)

# eager launches (no graph): do two streams overlap then?
def vis():
    m.visual_encoder(lips, T)
for name, work in (("eager visual only", [(vis, s1)]), ("eager main only", [(main_chain, s2)]),
                   ("eager both", [(vis, s1), (main_chain, s2)])):
    with torch.no_grad():
        for fn, s in work:
            with torch.cuda.stream(s):
                fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            for fn, s in work:
                with torch.cuda.stream(s):
                    fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
    print(f"{name:28s} {(time.perf_counter() - t0) / reps * 1e6:8.1f} us per step (host enqueue {(t1 - t0) / reps * 1e6:.1f})", flush=True)
