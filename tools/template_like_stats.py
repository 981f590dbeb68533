#!/usr/bin/env python3
"""Depth statistics (CPU oracle) of the three 100 000-splat avatars bench.py and tools/ref_on_gpu.py know: the ellipsoid stand-in, the avatar staged on the
reference's real head template (needs /root/reference: THIS container only; nothing of the OBJ is kept) and synthetic.head_mesh(kind="template_like").
The numbers quoted at synthetic.TEMPLATE_LIKE_LOG_SCALE_OFFSET come from here:   python tools/template_like_stats.py"""
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussianavatars_amd import synthetic as S, io as gio
from gaussianavatars_amd.gaussian_model import FlameGaussianModel
import oracle.gsr_oracle as O

def world_splats(rig, n_splats=100_000, t=0, log_scale_offset=0.0, translation_fix=None, n_faces=None):
    g = FlameGaussianModel(3, rig, binding_impl="unfused", device="cpu")
    F = rig["faces"].shape[0]
    arrs = S.bound_splats(n_splats, F, 3, seed=2)
    arrs["_scaling"] = arrs["_scaling"] + np.float32(log_scale_offset)
    arrs = gio.spatial_sort(arrs, rig["v_template"][rig["faces"]].mean(1))
    g.load_arrays(arrs, device="cpu", requires_grad=False)
    seq = S.flame_sequence(4, seed=4)
    V = rig["v_template"].shape[0]
    seq["static_offset"] = seq["static_offset"][:, :V]; seq["dynamic_offset"] = seq["dynamic_offset"][:, :V]
    if translation_fix is not None:
        seq["translation"] = translation_fix(seq["translation"])
    g.load_flame_param(seq, device="cpu", requires_grad=False)
    with torch.no_grad():
        g.select_mesh_by_timestep(t)
        ins = dict(means3D=g.get_xyz, shs=g.get_features, opacities=g.get_opacity, scales=g.get_scaling, rotations=g.get_rotation)
        return {k: v.detach().numpy().copy() for k, v in ins.items()}, g

def stats(ins, W=550, H=802, name=""):
    cam = S.orbit_camera(W, H, r=1.0, fovy_deg=20.0)
    tfx, tfy = math.tan(cam.FoVx*0.5), math.tan(cam.FoVy*0.5)
    s = O.make_settings(H, W, tfx, tfy, [1,1,1], 1.0, cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center)
    st = O.forward(s, ins["means3D"], ins["shs"], None, ins["opacities"], ins["scales"], ins["rotations"], None)
    rng = st.ranges.astype(np.int64); cnt = rng[:,1]-rng[:,0]
    gx=(W+15)//16; gy=(H+15)//16
    nc = st.n_contrib.reshape(H,W).astype(np.int64)
    # per-quadrant max n_contrib (position in the tile list)
    qh, qw = (H+7)//8, (W+7)//8
    pad = np.zeros((qh*8, qw*8), np.int64); pad[:H,:W]=nc
    qmax = pad.reshape(qh,8,qw,8).max(axis=(1,3)).ravel()
    q = lambda a: np.round(np.quantile(a,[0.5,0.75,0.9,0.95,0.99,0.999,1.0]),0).astype(int).tolist()
    out = dict(name=name, rect_instances=int(st.num_rendered), visible=int((st.radii>0).sum()), tile_cnt_q=q(cnt[cnt>0]), tiles_nonempty=int((cnt>0).sum()),
               quad_walk_q=q(qmax[qmax>0]), quads_nonempty=int((qmax>0).sum()), sum_quad_walk=int(qmax.sum()), quads_gt_300=int((qmax>300).sum()), quads_gt_600=int((qmax>600).sum()),
               mean_contrib_px=float(nc.mean()), radius_q=q(st.radii[st.radii>0]))
    return out

if __name__ == "__main__":
    import json
    O.set_threads(8)
    # ellipsoid (bench.py's scene)
    rig = S.flame_rig(seed=4)
    ins,_ = world_splats(rig)
    print(json.dumps(stats(ins, name="ellipsoid")))
    # staged avatar on the real template (no teeth), as tools/ref_on_gpu.py stages it
    v,f = S.read_obj_topology('/root/reference/flame_model/assets/flame/head_template_mesh.obj')
    d = S.flame_pickle_dict(v, f, 4)
    rigt = dict(v_template=d["v_template"].astype(np.float32), shapedirs=d["shapedirs"], posedirs=np.ascontiguousarray(d["posedirs"].reshape(-1,36).T),
                J_regressor=d["J_regressor"].astype(np.float32), lbs_weights=d["weights"].astype(np.float32), parents=np.asarray(S.FLAME_PARENTS,np.int64), faces=f)
    ext = float((v.max(0)-v.min(0)).max())
    fix = lambda tr: (tr*(ext/0.24) - (v.min(0)+v.max(0))/2).astype(np.float32)
    ins,_ = world_splats(rigt, log_scale_offset=S.BENCHMARK_LOG_SCALE_OFFSET, translation_fix=fix)
    print(json.dumps(stats(ins, name="template_staged")))
    ins,_ = world_splats(S.flame_rig(4, kind="template_like"), log_scale_offset=S.TEMPLATE_LIKE_LOG_SCALE_OFFSET)
    print(json.dumps(stats(ins, name="template_like")))
