import sys, numpy as np, os
sys.path.insert(0, '/root/repo')
from oracle import numpy_forward as onp, seeded

def trunc_bf16(x):
    return (x.astype(np.float32).view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
def split_bf16_3(x):
    parts=[]; r=x.astype(np.float32)
    for _ in range(3):
        p=trunc_bf16(r); parts.append(p); r=(r-p).astype(np.float32)
    return parts
def split_f16_2(x):
    hi=x.astype(np.float16).astype(np.float32); lo=(x-hi).astype(np.float16).astype(np.float32)
    return [hi,lo]
def mm32(a,b):  # fp32 accumulate (numpy sgemm)
    return a.astype(np.float32)@b.astype(np.float32).T
def mm_bf16_6(x,w):
    xs=split_bf16_3(x); ws=split_bf16_3(w)
    acc=np.zeros((x.shape[0],w.shape[0]),np.float32)
    for i,j in ((0,2),(2,0),(1,1),(1,0),(0,1),(0,0)): acc=acc+mm32(xs[i],ws[j])
    return acc
def pow2_scale(absmax, target=2.0**14):
    # power of two s with absmax * s in (target/2, target]
    e=np.floor(np.log2(np.maximum(absmax,1e-38)))
    return np.exp2(np.log2(target)-1-e).astype(np.float32)
def mm_f16_3(x,w,group=64):
    M,K=x.shape; N=w.shape[0]
    sw=pow2_scale(np.abs(w).max(axis=1,keepdims=True))
    ws=split_f16_2(w*sw)
    acc=np.zeros((M,N),np.float32)
    for g0 in range(0,K,group):
        xg=x[:,g0:g0+group]
        sx=pow2_scale(np.abs(xg).max(axis=1,keepdims=True))
        xs=split_f16_2(xg*sx)
        tmp=np.zeros((M,N),np.float32)
        for i,j in ((1,0),(0,1),(0,0)):
            tmp=tmp+mm32(xs[i],ws[j][:,g0:g0+group])
        acc=acc+tmp*(1.0/sx)
    return acc*(1.0/sw.T)

rng=np.random.default_rng(0)
for (M,N,K,kind) in ((512,768,256,'n'),(512,512,2048,'n'),(512,2048,512,'relu'),(256,512,512,'wide')):
    x=(rng.standard_normal((M,K))*2+0.7).astype(np.float32); w=(rng.standard_normal((N,K))*0.06).astype(np.float32)
    if kind=='relu': x=np.maximum(x,0)
    if kind=='wide': x=x*np.exp(rng.standard_normal((M,K))*4).astype(np.float32); w=w*np.exp(rng.standard_normal((N,K))*3).astype(np.float32)
    ref=x.astype(np.float64)@w.astype(np.float64).T
    scale=np.abs(ref).max()
    rms=np.sqrt((ref**2).mean())
    def rep(name,y): print(f'  {name:28s} max err/max|y| {np.abs(y-ref).max()/scale:.3e}   rms err/rms|y| {np.sqrt(((y-ref)**2).mean())/rms:.3e}')
    print(M,N,K,kind)
    rep('fp32 sgemm', mm32(x,w)); rep('bf16 x3, 6 products', mm_bf16_6(x,w))
    for g in (32,64,128,K): rep(f'fp16 x2, 3 products, group {g}', mm_f16_3(x,w,g))

F,d,h,Le,Lf,S=257,256,4,2,2,2
B,T,Nf,H,W=2,63,50,32,32
st=seeded.fill_state(seeded.model_shapes(F,d,h,Le,Lf,S),1234)
mixed,lips=seeded.inputs(1234,B,F,T,Nf,H,W)
_,m64=onp.forward(st,mixed,lips,h,S,dtype=np.float64)
_,m32=onp.forward(st,mixed,lips,h,S,dtype=np.float32)
print('model fp32 numpy vs fp64: masks', np.abs(m32-m64).max())
orig=onp.linear
for name,f in (('bf16x3/6',mm_bf16_6),('fp16x2/3 g64',lambda a,b: mm_f16_3(a,b,64)),('fp16x2/3 g=K',lambda a,b: mm_f16_3(a,b,a.shape[1]))):
    def lin(x,w,b=None,f=f):
        if x.dtype!=np.float32: return orig(x,w,b)
        sh=x.shape; y=f(x.reshape(-1,sh[-1]),w).reshape(*sh[:-1],w.shape[0])
        return y if b is None else y+b
    onp.linear=lin
    _,ms=onp.forward(st,mixed,lips,h,S,dtype=np.float32)
    print(f'model with {name} linears: masks vs fp64', np.abs(ms-m64).max())
onp.linear=orig
