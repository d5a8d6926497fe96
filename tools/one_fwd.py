import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "av-separation-transformer_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, av_separation as av, bench
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]; B = wl["batch"]
torch.manual_seed(0)
m = av.AVSeparationTransformer(dropout=0.0, **wl["model"]).cuda().eval()
ds = av.SyntheticAVDataset(num_samples=B, **wl["data"]); it = [ds[i] for i in range(B)]
mx = torch.stack([x["mixed_spec"] for x in it]).cuda(); lp = torch.stack([x["lip_frames"] for x in it]).cuda()
with torch.no_grad():
    for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 2): m(mx, lp)
torch.cuda.synchronize()
