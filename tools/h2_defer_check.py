import os, sys, ctypes as C
os.environ["AVSEP_LIB"] = "dev"
sys.path.insert(0, "av-separation-transformer_amd")
import torch
from av_separation import _native
lib = _native.load(); dev = torch.device("cuda:0")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
torch.manual_seed(0)
for (M, N, K, act, planes) in ((16064, 1536, 512, 0, False), (16064, 2048, 512, 1, True), (16064, 512, 512, 0, False), (8032, 4096, 512, 0, False)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
    xp = torch.zeros(K // 32 * 2 * M * 32, dtype=torch.int16, device=dev)
    assert lib.avsep_op_split_h2(x.data_ptr(), K, xp.data_ptr(), M, M, K, None, 10, st) == 0
    ew = torch.zeros(N, dtype=torch.int32, device=dev); l2 = torch.zeros(N, device=dev)
    assert lib.avsep_op_h2_row_stats(w.data_ptr(), N, K, ew.data_ptr(), l2.data_ptr(), st) == 0
    wp = torch.zeros(K // 32 * 2 * N * 32, dtype=torch.int16, device=dev)
    assert lib.avsep_op_split_h2(w.data_ptr(), K, wp.data_ptr(), N, N, K, ew.data_ptr(), 0, st) == 0
    cs = torch.ldexp(torch.ones(N, device=dev), -(ew + 10))
    outs = []
    for env in ({}, {"AVSEP_H2_NO_DEFER": "1"}):
        os.environ.pop("AVSEP_H2_NO_DEFER", None); os.environ.update(env)
        if planes:
            y = torch.full((N // 32 * 2 * M * 32,), -1, dtype=torch.int16, device=dev)
            rc = lib.avsep_op_linear_h2(xp.data_ptr(), M, wp.data_ptr(), N, cs.data_ptr(), None, b.data_ptr(), None, None, y.data_ptr(), M, 3, M, N, K, act, st)
        else:
            y = torch.full((M, N), float("nan"), device=dev)
            rc = lib.avsep_op_linear_h2(xp.data_ptr(), M, wp.data_ptr(), N, cs.data_ptr(), None, b.data_ptr(), None, y.data_ptr(), None, 0, 0, M, N, K, act, st)
        assert rc == 0, lib.avsep_last_error()
        torch.cuda.synchronize(); outs.append(y.clone())
    a, c = outs
    if planes:
        bad = (a != c).nonzero().flatten()
        print((M, N, K, act, planes), "mismatching int16:", bad.numel(), "of", a.numel(), bad[:8].tolist())
    else:
        neq = (a != c) | (a.isnan() != c.isnan())
        bad = neq.nonzero()
        print((M, N, K, act, planes), "mismatching:", bad.shape[0], "of", a.numel(), "nan in deferred:", int(a.isnan().sum()))
        if bad.shape[0]:
            rows = bad[:, 0].unique(); cols = bad[:, 1].unique()
            print("   rows", rows[:10].tolist(), "... n", rows.numel(), "cols", cols[:10].tolist(), "... n", cols.numel(), "max diff", float((a - c)[neq].abs().max()))
