import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "av-separation-transformer_amd"))
import torch
from av_separation import _native
lib = _native.load(); dev = torch.device("cuda:0"); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (M, Cc) in [(4016, 512), (4016, 2048), (1200, 512), (307200, 32), (76800, 64), (19200, 128)]:
    a = torch.randn(M, Cc, device=dev); b = torch.randn(M, Cc, device=dev)
    ns = lib.avsep_op_colreduce_scratch_floats(M, Cc); scr = torch.empty(ns, device=dev)
    o0 = torch.empty(Cc, device=dev); o1 = torch.empty(Cc, device=dev)
    f = lambda: lib.avsep_op_colreduce(a.data_ptr(), b.data_ptr(), scr.data_ptr(), o0.data_ptr(), o1.data_ptr(), M, Cc, st)
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 100 * 1e3
    ref0 = a.double().sum(0); ref1 = (a.double() * b.double()).sum(0)
    print(f"M={M:6d} C={Cc:4d}: {us:7.2f} us (both launches)  {2.0*M*Cc*4/us/1e6:5.2f} TB/s   err {float((o0.double()-ref0).abs().max()):.2e} {float((o1.double()-ref1).abs().max()):.2e}")
