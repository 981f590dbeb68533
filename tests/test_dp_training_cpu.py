"""SURVEY.md 8(f) N4 in a LOOP: frame-parallel (data-parallel) training of the reference's own FlameGaussianModel on two ranks (gloo).

Twenty iterations of what train.py:118-206 does per frame -- select_mesh_by_timestep, render, L1, backward, the two statistics lines,
every fifth iteration densify_and_prune, then the Adam step -- with the two additions a replica needs
(gaussianavatars_amd/frame_parallel.py): `allreduce_gradients(method="reduce_scatter")` before the optimiser, and
`sync_densification_stats` + `sync_mesh_for_densification` + `seed_all_ranks` before the reference's densify_and_prune (scene/gaussian_model.py:426-515), whose
clone / split / prune decisions and torch.normal draws must come out the same on every rank although every rank saw different frames.
Afterwards every parameter, the binding, the densification state and the Adam moments are bit-identical on both ranks, and the model has
actually been densified and pruned on the way.

The classes are the reference's (imported from a directory of symlinks to /root/reference that also holds the generated FLAME pickles,
see tests/test_reference_entry_cpu.py); this box has no GPU, so "cuda" is mapped to the host (tests/ref_cpu_env.py), the model runs its
composed-torch methods (GAA_BINDING_IMPL=unfused) and the rasterizer is a differentiable torch stand-in at the Function boundary."""
import hashlib
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "scene")), reason="reference checkout not present on this box")


def _stub_rasterize(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, rs, sh_rest=None):
    """Stands in for _RasterizeGaussians.apply: an image that depends smoothly and per splat on every input, radii that change from
    frame to frame, a screen-space gradient whose norm differs from splat to splat (so that the densification thresholds cut through
    the model)."""
    P = means3D.shape[0]
    i = torch.arange(P, dtype=torch.float32)
    w = 0.5 + 0.5 * torch.sin(0.37 * i)
    dc = sh[:, 0, :].sum(1) if sh.numel() else colors_precomp.sum(1)
    per = torch.tanh(4.0 * means3D.sum(1)) * opacities.squeeze(1) + 0.1 * scales.sum(1) + 0.01 * (rotations ** 2).sum(1) + 0.05 * dc
    v = (w * per).mean() + (means2D[:, :2] * (1e-3 * (1.0 + (torch.arange(P) % 7).float()))[:, None]).sum()   # |d/d means2D| = sqrt(2) (1..7) e-3 per splat
    img = v + torch.zeros(3, int(rs.image_height), int(rs.image_width))
    phase = int(abs(float(means3D[0, 0].detach())) * 1e6)
    radii = ((torch.arange(P) * 7 + phase) % 5).to(torch.int32)
    return img, radii, radii > 0


def _digest(t):
    return hashlib.sha1(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()


def _worker(rank, world, port, farm, ply, q):
    try:
        _train(rank, world, port, farm, ply, q)
    except BaseException:
        import traceback

        q.put((rank, "ERROR", traceback.format_exc()))
        raise


def _train(rank, world, port, farm, ply, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      GAA_BINDING_IMPL="unfused")
    os.chdir(farm)
    sys.path.insert(0, ROOT)
    import types
    from pathlib import Path

    import torch.distributed as dist

    from tests import ref_cpu_env

    ref_cpu_env._no_cuda()
    from gaussianavatars_amd import frame_parallel as fp
    from gaussianavatars_amd import patch
    from gaussianavatars_amd import rasterizer as R

    patch.patch_reference(reference_root=farm)
    R._RasterizeGaussians.apply = staticmethod(_stub_rasterize)
    from gaussian_renderer import render                     # the mirror with the reference's signature (patch_reference installed it)
    from scene.flame_gaussian_model import FlameGaussianModel
    from utils.loss_utils import l1_loss

    fp.init_process_group("gloo")
    torch.manual_seed(0)
    g = FlameGaussianModel(3)
    # patch_reference keeps a model in Morton order by default (patch._hook_spatial_order: after load_ply and after every
    # densify_and_prune); the loop below runs with the hook OFF so that the explicit re-sort at its end has something to move, the hook
    # itself is exercised after that
    os.environ["GAA_SPATIAL_SORT"] = "0"
    g.load_ply(Path(ply), has_target=False)
    binding_in_file = g.binding.clone()

    def order_holds(what):
        # `_gaa_order`: row i of the model is row _gaa_order[i] of the loaded file, -1 for splats born since -- followed through the reference's
        # prune_points / densification_postfix and every re-sort (a splat never changes its face, so the binding tells)
        o = g._gaa_order
        assert o is not None and o.shape[0] == g._xyz.shape[0], what
        old = o >= 0
        assert torch.equal(g.binding[old], binding_in_file[o[old]]), what + ": the recorded order names other rows of the file"
        assert o[old].unique().numel() == int(old.sum()), what
        return int(old.sum())

    assert order_holds("as loaded") == binding_in_file.shape[0]
    g.spatial_lr_scale = 1.0
    g.max_radii2D = torch.zeros(g.get_xyz.shape[0])     # (create_from_pcd / restore set it in train.py's own start-up, load_ply does not)
    opt = types.SimpleNamespace(percent_dense=0.01, position_lr_init=0.005, position_lr_final=0.00005, position_lr_delay_mult=0.01,
                                position_lr_max_steps=600_000, feature_lr=0.0025, opacity_lr=0.05, scaling_lr=0.017, rotation_lr=0.001,
                                flame_pose_lr=1e-5, flame_trans_lr=1e-6, flame_expr_lr=1e-3)
    g.training_setup(opt)
    from gaussianavatars_amd import synthetic as S

    cam = S.orbit_camera(48, 40)
    for k in ("world_view_transform", "full_proj_transform", "camera_center"):
        setattr(cam, k, torch.as_tensor(getattr(cam, k)))
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
    bg = torch.ones(3)
    gt = torch.full((3, 40, 48), 0.4)
    T = g.num_timesteps
    sizes = [int(g._xyz.shape[0])]
    for it in range(1, 21):
        t = (world * it + rank) % T                                  # every rank its own frame
        g.update_learning_rate(it)
        g.select_mesh_by_timestep(t)
        pkg = render(cam, g, pipe, bg)
        loss = l1_loss(pkg["render"], gt)
        loss.backward()
        with torch.no_grad():
            vis, radii = pkg["visibility_filter"], pkg["radii"]
            g.max_radii2D[vis] = torch.max(g.max_radii2D[vis], radii[vis])                      # train.py:197
            g.add_densification_stats(pkg["viewspace_points"], vis)                              # train.py:198
            params = [p for grp in g.optimizer.param_groups for p in grp["params"]]
            fp.allreduce_gradients(params, average=True, method="reduce_scatter")               # N4: before anything reads the gradients
            if it % 5 == 0:
                fp.sync_densification_stats(g)                                                  # the statistics of ALL ranks' frames
                fp.sync_mesh_for_densification(g, t)                                            # ... decided on ONE mesh (rank 0's frame): get_scaling reads face_scaling
                fp.seed_all_ranks(it)                                                           # the same torch.normal draws in densify_and_split
                g.densify_and_prune(6e-3, 0.3, 1.0, 20)                                      # the reference's, scene/gaussian_model.py:498-515
                sizes.append(int(g._xyz.shape[0]))
            g.optimizer.step()
            g.optimizer.zero_grad(set_to_none=True)
    # ---- after all that appending and pruning: the splats back into Morton order (gaussian_model.spatial_resort), parameters, Adam moments,
    # statistics and binding moving together; then one more iteration on the re-sorted model
    from gaussianavatars_amd.gaussian_model import spatial_resort

    def rows():
        grp = {gr["name"]: gr["params"][0] for gr in g.optimizer.param_groups if len(gr["params"]) == 1}
        st = g.optimizer.state
        cols = [g._xyz, g._opacity, g._scaling, g._rotation, g._features_dc.flatten(1), g._features_rest.flatten(1), st[grp["xyz"]]["exp_avg"],
                st[grp["opacity"]]["exp_avg_sq"], st[grp["f_rest"]]["exp_avg"].flatten(1), g.binding[:, None], g.max_radii2D[:, None], g.denom, g.xyz_gradient_accum]
        return torch.cat([c.detach().double().reshape(c.shape[0], -1) for c in cols], 1)

    survivors = order_holds("after four rounds of densify_and_prune")
    assert 0 < survivors and survivors < int(g._xyz.shape[0]) and (sizes[-1] != sizes[0] or survivors < sizes[0])   # (splats were born, some of the file's may be gone)
    before = rows()
    perm = spatial_resort(g)
    assert order_holds("after the re-sort") == survivors
    assert sorted(perm.tolist()) == list(range(before.shape[0])) and not torch.equal(perm, torch.arange(before.shape[0]))
    assert torch.equal(before[perm], rows()), "a per-splat quantity did not move with its splat"
    assert all(g.optimizer.state.get(gr["params"][0]) is not None for gr in g.optimizer.param_groups if gr["name"] in ("xyz", "opacity", "f_rest"))
    assert g._xyz is [gr["params"][0] for gr in g.optimizer.param_groups if gr["name"] == "xyz"][0]     # the model holds the optimiser's parameter objects
    from gaussianavatars_amd.gaussian_model import template_face_centers
    where = torch.as_tensor(template_face_centers(g))[g.binding]
    spread = lambda w: float((w[1:] - w[:-1]).norm(dim=1).mean())
    assert spread(where) < 0.5 * spread(torch.as_tensor(template_face_centers(g))[g.binding[torch.argsort(perm)]])   # neighbours in memory sit on neighbouring faces now
    g.select_mesh_by_timestep(rank)
    pkg = render(cam, g, pipe, bg)
    l1_loss(pkg["render"], gt).backward()
    with torch.no_grad():
        fp.allreduce_gradients([p for grp in g.optimizer.param_groups for p in grp["params"]], average=True, method="reduce_scatter")
        g.optimizer.step()
        g.optimizer.zero_grad(set_to_none=True)
    # ---- the default: the reference's densify_and_prune followed by the re-sort hook -- nothing left for an explicit re-sort to move
    os.environ["GAA_SPATIAL_SORT"] = "1"
    with torch.no_grad():
        fp.sync_densification_stats(g)
        fp.sync_mesh_for_densification(g, 0)
        fp.seed_all_ranks(99)
        g.densify_and_prune(6e-3, 0.3, 1.0, 20)
        again = spatial_resort(g)
    assert torch.equal(again, torch.arange(again.shape[0])), "densify_and_prune did not leave the model in Morton order"
    order_holds("after densify_and_prune with the re-sort hook on")
    assert g._xyz is [gr["params"][0] for gr in g.optimizer.param_groups if gr["name"] == "xyz"][0]
    state = {k: _digest(getattr(g, k)) for k in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "binding",
                                                   "binding_counter", "max_radii2D", "xyz_gradient_accum", "denom")}
    state.update({"flame_" + k: _digest(v) for k, v in g.flame_param.items()})
    for gi, grp in enumerate(g.optimizer.param_groups):
        for pi, p in enumerate(grp["params"]):
            st = g.optimizer.state.get(p, {})
            for name in ("exp_avg", "exp_avg_sq"):
                if name in st:
                    state[f"adam_{grp['name']}_{pi}_{name}"] = _digest(st[name])
    q.put((rank, state, sizes))
    dist.barrier()
    dist.destroy_process_group()


@needs_ref
def test_two_rank_training_loop_with_the_references_densification(tmp_path):
    from gaussianavatars_amd import synthetic as S
    from tests.test_reference_entry_cpu import symlink_farm

    farm = str(tmp_path / "checkout")
    os.makedirs(farm)
    out = S.write_reference_assets(symlink_farm(farm), str(tmp_path / "avatar"),
                                   os.path.join(REF, "flame_model", "assets", "flame", "head_template_mesh.obj"), n_frames=12)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, farm, out["point_cloud"], q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600)]
    if res[0][1] != "ERROR":
        res.append(q.get(timeout=600))
    for p in procs:
        p.join(60 if res[0][1] != "ERROR" else 5)
        if p.is_alive():
            p.kill()
    for r in res:
        assert r[1] != "ERROR", r[2]
    assert all(p.exitcode == 0 for p in procs)
    res.sort()
    (_, s0, n0), (_, s1, n1) = res
    assert n0 == n1 and len(n0) == 5
    assert len(set(n0)) > 2, f"the model was never densified / pruned: sizes {n0}"     # the loop really changed the model's size
    assert s0.keys() == s1.keys() and len(s0) > 25
    diff = [k for k in s0 if s0[k] != s1[k]]
    assert not diff, f"replicas diverged in {diff}"
