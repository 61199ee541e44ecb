import os, sys
sys.path.insert(0, "/root/repo")
import importlib, numpy as np
rt = importlib.import_module("raytracing-1w_amd")
ctx = rt.Context(rt.Scene.reference(5, build_seed=1), 0)
W=H=600
a,sa=ctx.render(W,H,256); b,sb=ctx.render(W,H,256,f32=True)
print("segments ratio", sb["segments"]/sa["segments"])
qa,qb=rt.quantize(a).astype(float),rt.quantize(b).astype(float)
print("global", qa.mean(axis=(0,1))-qb.mean(axis=(0,1)))
bl=(qa-qb).reshape(6,100,6,100,3).mean(axis=(1,3)); print("block max", np.abs(bl).max())
print("black f32 not f64", int(((b==0).all(axis=2)&~(a==0).all(axis=2)).sum()), "black f64 not f32", int(((a==0).all(axis=2)&~(b==0).all(axis=2)).sum()), "zero-channel px f32", int((b==0).any(axis=2).sum()), "f64", int((a==0).any(axis=2).sum()))
for sl in (np.s_[:,5:25],np.s_[:,-25:-5],np.s_[5:25,:],np.s_[-25:-5,:]): print("wall", b[sl].mean()/a[sl].mean())
for arm,aspect,(w,h,spp) in ((0,1.5,(240,160,32)),(6,None,(128,128,32)),(7,None,(128,128,32)),(2,None,(128,72,16))):
    c=rt.Context(rt.Scene.reference(arm,build_seed=1,aspect_ratio=aspect),0); x,_=c.render(w,h,spp); y,sy=c.render(w,h,spp,f32=True); print(arm, y.mean()/x.mean(), np.isfinite(y).all(), sy["sorted"])
