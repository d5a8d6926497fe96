import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import numpy_forward as onp, seeded

def bf16_round(x):
    # round-to-nearest-even fp32 -> bf16, returned as fp32
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return u.astype(np.uint32).view(np.float32)

def split(x, terms):
    parts=[]; r=x.astype(np.float32)
    for _ in range(terms):
        p=bf16_round(r); parts.append(p); r=(r-p).astype(np.float32)
    return parts

def mm_split(x, w, terms, nprod):
    # x [M,K], w [N,K]; products chosen by order of magnitude: (i,j) with i+j < ... ; fp32 accumulation emulated by float64 sum then cast (optimistic) -> use float32 matmul per product
    xs=split(x,terms); ws=split(w,terms)
    combos=sorted([(i,j) for i in range(terms) for j in range(terms)], key=lambda t:(t[0]+t[1],t[0]))[:nprod]
    acc=np.zeros((x.shape[0],w.shape[0]),np.float32)
    for i,j in reversed(combos):   # small terms first
        acc=acc+(xs[i].astype(np.float32)@ws[j].astype(np.float32).T)
    return acc

rng=np.random.default_rng(0)
M,N,K=512,768,256
x=(rng.standard_normal((M,K))*2+0.7).astype(np.float32); w=(rng.standard_normal((N,K))*0.06).astype(np.float32)
ref=x.astype(np.float64)@w.astype(np.float64).T
scale=np.abs(ref).max()
print('fp32 matmul      rel err', np.abs(x@w.T-ref).max()/scale)
for terms,nprod in ((1,1),(2,3),(2,4),(3,6),(3,9)):
    y=mm_split(x,w,terms,nprod)
    print(f'bf16 terms {terms} products {nprod}: rel err', np.abs(y-ref).max()/scale)

# model level: patch oracle linear / conv to split versions
F,d,h,Le,Lf,S=257,256,4,2,2,2
B,T,Nf,H,W=2,63,50,32,32
st=seeded.fill_state(seeded.model_shapes(F,d,h,Le,Lf,S),1234)
mixed,lips=seeded.inputs(1234,B,F,T,Nf,H,W)
_,m64=onp.forward(st,mixed,lips,h,S,dtype=np.float64)
_,m32=onp.forward(st,mixed,lips,h,S,dtype=np.float32)
print('model fp32 numpy vs fp64: masks', np.abs(m32-m64).max())
orig=onp.linear
for terms,nprod in ((2,3),(3,6)):
    def lin(x,w,b=None,terms=terms,nprod=nprod):
        if x.dtype!=np.float32: return orig(x,w,b)
        sh=x.shape; y=mm_split(x.reshape(-1,sh[-1]),w,terms,nprod).reshape(*sh[:-1],w.shape[0])
        return y if b is None else y+b
    onp.linear=lin
    _,ms=onp.forward(st,mixed,lips,h,S,dtype=np.float32)
    print(f'model with split-bf16 linears (terms {terms}, products {nprod}; convs and attention matmuls still fp32): masks vs fp64', np.abs(ms-m64).max())
onp.linear=orig
