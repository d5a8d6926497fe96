import torch
dev = torch.device("cuda:0")
for M, N, K in ((16064, 2048, 512), (16064, 512, 2048), (16064, 512, 512), (2016, 768, 256)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev)
    for _ in range(3):
        y = torch.mm(x, w.t())
torch.cuda.synchronize()
