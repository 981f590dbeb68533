import sys, numpy as np, torch, math
sys.path.insert(0, '.')
from tests.scenes import scene, settings_args
from gaussianavatars_amd.debug import forward_state
from gaussianavatars_amd.rasterizer import GaussianRasterizationSettings
dev = torch.device('cuda:0')
cam, sp, bg, deg, mod = scene("cfg1")
a = settings_args(cam, bg, deg, mod)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
rs = GaussianRasterizationSettings(a["H"], a["W"], a["tanfovx"], a["tanfovy"], t(a["bg"]), mod, t(a["viewmatrix"]), t(a["projmatrix"]), deg, t(a["campos"]), False, False)
args = (t(sp["means3D"]), t(sp["shs"]), None, t(sp["opacities"]), t(sp["scales"]), t(sp["rotations"]), None)
for c in (False, True):
    h = forward_state(rs, *args, tile_culling=c)
    print(c, h["num_rendered"], h["rect_instances"], int(h["tiles_touched"].sum()))
