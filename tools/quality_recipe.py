#!/usr/bin/env python3
"""The quality half of BASELINE.json's metric ("SI-SNRi dB vs ref") on TRAINED weights: the reference's end-to-end recipe
(/root/reference/demo.py:116-198 -- 500-item SyntheticAVDataset 1 s @ 8 kHz, d_model 128 / 4 heads / 2+2 layers / dropout
0.1; SNR of the untrained model on the first 20 items; one pass of the shuffled batch-8 loader = 63 Adam steps at lr 3e-4
with SeparationLoss(0.5) and clip 1.0, demo.py:83-113; SNR again; "SNR improvement" = output SNR after - input SNR before,
demo.py:177; README.md:61-65 prints +37.23 dB) run twice from the SAME initial weights and the SAME batch order:

  run_gpu       on the HIP path (av_separation: train-mode forward / backward of _train.py, eval forward of the fused path)
  run_cpu_port  on the CPU port of the reference (oracle/torch_cpu.py forward_train / forward -- TEST INFRASTRUCTURE, pinned
                against the reference's own gradients and outputs in tests/test_oracle.py)

Dropout masks differ (torch's CPU generator vs the HIP path's counter hash), so the two trained models are two samples of
the same recipe; tests/test_train_gpu.py gates |output SNR (HIP) - output SNR (CPU port)| < 1 dB and improvement >= 35 dB,
bench.py prints the pair in `cpu_baseline.quality`.  Importable (bench.py, tests) and runnable:
    python3 tools/quality_recipe.py            -> profiles/rNN_train_eval_recipe.txt"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (ROOT, os.path.join(ROOT, "av-separation-transformer_amd")):
    if p_ not in sys.path:
        sys.path.insert(0, p_)
import numpy as np  # noqa: E402
import torch  # noqa: E402

MODEL = dict(freq_bins=257, d_model=128, nhead=4, num_encoder_layers=2, num_fusion_layers=2, num_speakers=2)   # demo.py:148-156
DATA = dict(num_samples=500, sample_rate=8000, duration=1.0, n_fft=512, hop_length=128, num_frames=25, frame_h=32, frame_w=32,
            speaker_freqs=(220.0, 440.0))                                                                       # demo.py:126-137


def batch_order(n_items, seed, batch=8, steps=100):
    """DataLoader(batch_size=8, shuffle=True) semantics with a seeded permutation: one pass, last batch ragged, at most
    `steps` batches (demo.py:87,94-111: 500 items give 63 batches, so the reference's "100 steps" are 63)."""
    perm = torch.randperm(n_items, generator=torch.Generator().manual_seed(seed)).tolist()
    out = [perm[i:i + batch] for i in range(0, n_items, batch)]
    return out[:steps]


def _stack(items, idx, key):
    return torch.stack([items[i][key] for i in idx])


def evaluate_with(forward, items, num_eval=20):
    """demo.py:31-64 with any `forward(mixed[1,F,T], lips[1,N,H,W]) -> separated[1,S,F,T]` (CPU tensors in and out)."""
    from av_separation.evaluate import snr_db, permutation_snr
    snr_in, snr_out = [], []
    for i in range(num_eval):
        it = items[i]
        clean = it["clean_specs"].numpy()
        sep = forward(it["mixed_spec"].unsqueeze(0), it["lip_frames"].unsqueeze(0)).squeeze(0).numpy()
        mix = it["mixed_spec"].numpy()
        snr_in.extend(snr_db(clean[s], mix - clean[s]) for s in range(clean.shape[0]))
        snr_out.append(permutation_snr(sep, clean))
    return float(np.mean(snr_in)), float(np.mean(snr_out))


def make_items(av, order):
    ds = av.SyntheticAVDataset(**DATA)
    need = sorted(set(range(20)) | {i for b in order for i in b})
    return {i: ds[i] for i in need}


def run_gpu(av, dev, items, order, init_seed=0, lr=3e-4):
    from av_separation.losses import SeparationLoss
    torch.manual_seed(init_seed)
    model = av.AVSeparationTransformer(dropout=0.1, **MODEL).to(dev)
    state0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}

    def fwd(mx, lp):
        with torch.no_grad():
            return model(mx.to(dev), lp.to(dev))[0].cpu()
    model.eval()
    in0, out0 = evaluate_with(fwd, items)
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    crit = SeparationLoss(l1_weight=0.5)
    model.train()
    losses, t0 = [], time.perf_counter()
    for idx in order:
        opt.zero_grad()
        sep, _ = model(_stack(items, idx, "mixed_spec").to(dev), _stack(items, idx, "lip_frames").to(dev))
        loss = crit(sep, _stack(items, idx, "clean_specs").to(dev))
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        losses.append(float(loss.detach()))
    torch.cuda.synchronize()
    secs = time.perf_counter() - t0
    model.eval()
    in1, out1 = evaluate_with(fwd, items)
    return dict(in_snr=in0, out_snr_untrained=out0, out_snr=out1, improvement=out1 - in0, losses=losses, train_seconds=secs,
                steps=len(order)), state0


def run_cpu_port(state0, items, order, lr=3e-4, torch_seed=0):
    from oracle import torch_cpu
    from av_separation.losses import SeparationLoss
    state = {k: v.clone() for k, v in state0.items()}
    params = []
    for k, t in state.items():
        if t.is_floating_point() and "running_" not in k and not k.endswith(".pe"):
            t.requires_grad_()
            params.append(t)
    h, S = MODEL["nhead"], MODEL["num_speakers"]

    def fwd(mx, lp):
        return torch_cpu.forward({k: v.detach() for k, v in state.items()}, mx, lp, h, S)[0].contiguous()
    avail = torch.get_num_threads()
    torch.set_num_threads(min(avail, 16))               # small ops: a 128-core host is slowest with all its threads
    in0, out0 = evaluate_with(fwd, items)
    torch.manual_seed(torch_seed)                       # the port's dropout masks
    opt = torch.optim.Adam(params, lr=lr)
    crit = SeparationLoss(l1_weight=0.5)
    losses, t0 = [], time.perf_counter()
    for idx in order:
        opt.zero_grad()
        sep, _ = torch_cpu.forward_train(state, _stack(items, idx, "mixed_spec"), _stack(items, idx, "lip_frames"), h, S, 0.1)
        loss = crit(sep, _stack(items, idx, "clean_specs"))
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        losses.append(float(loss.detach()))
    secs = time.perf_counter() - t0
    in1, out1 = evaluate_with(fwd, items)
    torch.set_num_threads(avail)
    return dict(in_snr=in0, out_snr_untrained=out0, out_snr=out1, improvement=out1 - in0, losses=losses, train_seconds=secs,
                steps=len(order))


def main():
    import av_separation as av
    from av_separation import _native
    dev = torch.device("cuda:0")
    order = batch_order(DATA["num_samples"], seed=7)
    items = make_items(av, order)
    g, state0 = run_gpu(av, dev, items, order)
    c = run_cpu_port(state0, items, order)
    print(f"library build {_native.load().avsep_build_id().decode()}; model {MODEL}; {len(order)} steps (one pass of the "
          f"500-item batch-8 loader, demo.py:94-111), seeded batch order, same initial weights for both runs")
    for name, r in (("HIP path", g), ("CPU port of the reference", c)):
        print(f"{name:26s}: input SNR {r['in_snr']:.2f} dB | untrained output {r['out_snr_untrained']:.2f} dB | trained output "
              f"{r['out_snr']:.2f} dB | improvement {r['improvement']:+.2f} dB | loss {r['losses'][0]:.2f} -> "
              f"{np.mean(r['losses'][-5:]):.2f} | training {r['train_seconds']:.2f} s")
    print(f"difference of the trained output SNR: {g['out_snr'] - c['out_snr']:+.2f} dB (gate 1 dB, tests/test_train_gpu.py); "
          f"reference README.md:61-65: input 0.01, untrained 3.20, trained 37.24, improvement +37.23 dB (unseeded)")


if __name__ == "__main__":
    main()
