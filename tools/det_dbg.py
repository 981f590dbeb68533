import sys; sys.path.insert(0,'.')
import numpy as np, torch
from tests.scenes import scene, settings_args
from oracle import gsr_oracle as O
from gaussianavatars_amd import rasterizer as R
from gaussianavatars_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
dev=torch.device('cuda:0')
cam,sp,bg,deg,mod=scene(sys.argv[1] if len(sys.argv)>1 else "sh3_small"); a=settings_args(cam,bg,deg,mod)
s=O.make_settings(**a); st=O.forward(s,sp["means3D"],sp["shs"],None,sp["opacities"],sp["scales"],sp["rotations"],None)
gpix=np.random.default_rng(5).normal(0,1,(3,a["H"],a["W"])).astype(np.float32); ref=O.backward(s,st,gpix)
t=lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
rs=GaussianRasterizationSettings(a["H"],a["W"],a["tanfovx"],a["tanfovy"],t(a["bg"]),mod,t(a["viewmatrix"]),t(a["projmatrix"]),deg,t(a["campos"]),False,False)
for det in (False,True):
    R.set_deterministic(det)
    tt=lambda x: t(x).requires_grad_(True)
    m3,sh,op,sc,ro=tt(sp["means3D"]),tt(sp["shs"]),tt(sp["opacities"]),tt(sp["scales"]),tt(sp["rotations"]); m2=torch.zeros_like(m3,requires_grad=True)
    color,radii=GaussianRasterizer(rs)(means3D=m3,means2D=m2,shs=sh,opacities=op,scales=sc,rotations=ro)
    (color*t(gpix)).sum().backward()
    for k,g in (("means2D",m2.grad),("opacities",op.grad),("shs",sh.grad),("means3D",m3.grad),("scales",sc.grad),("rotations",ro.grad)):
        r=ref[k]; gg=g.cpu().numpy().reshape(r.shape); d=np.abs(gg-r)
        i=np.unravel_index(d.argmax(),d.shape)
        print(det,k,"rel",d.max()/np.abs(r).max(),"at",i,"got",gg[i],"ref",r[i])
