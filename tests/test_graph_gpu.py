"""The frame step recorded once and replayed as ONE hipGraph launch (gaussianavatars_amd/graphs.py) against the same step run
eagerly: same image bits, and -- with the deterministic backward -- the same splat-gradient bits, for every replayed timestep.  The
recorded forward does not wait for its instance count (include/gsr.h: GsrSettings.deferred_count): a frame that overflows the
recorded binning capacity must be reported, must not fault, and must be recoverable by recording again."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class _Pipe:
    debug = False
    compute_cov3D_python = False
    convert_SHs_python = False


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _scene(dev, n=20000, frames=12):
    from gaussianavatars_amd import synthetic as S
    from gaussianavatars_amd.gaussian_model import FlameGaussianModel

    g = FlameGaussianModel(3, S.flame_rig(seed=4), device=dev)
    g.load_arrays(S.bound_splats(n, S.FLAME_F, 3, seed=2), device=dev, requires_grad=True)
    g.load_flame_param(S.flame_sequence(frames, seed=4), device=dev, requires_grad=True)
    cam = S.orbit_camera(208, 176, r=1.0, fovy_deg=20.0)
    for k in ("world_view_transform", "full_proj_transform", "camera_center"):
        setattr(cam, k, torch.as_tensor(getattr(cam, k), device=dev))
    return g, cam


_LEAVES = ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity")


def _zero(g):
    from gaussianavatars_amd.graphs import release_mesh

    release_mesh(g)
    for n in _LEAVES:
        getattr(g, n).grad = None
    for v in g.flame_param.values():
        if v.requires_grad:
            v.grad = None


def _step(g, cam, bg, target, t):
    from gaussianavatars_amd.gaussian_renderer import l1_loss, render

    g.select_mesh_by_timestep(t)
    pkg = render(cam, g, _Pipe, bg)
    loss = l1_loss(pkg["render"], target)
    loss.backward()
    return loss.detach(), pkg["render"].detach(), pkg["radii"], pkg["viewspace_points"]


def test_graphed_step_equals_the_eager_step_frame_by_frame():
    from gaussianavatars_amd import rasterizer as R
    from gaussianavatars_amd.graphs import FlameRowFeeder, GraphedStep

    dev = _dev()
    g, cam = _scene(dev)
    bg = torch.ones(3, device=dev)
    target = torch.full((3, 176, 208), 0.5, device=dev)
    prev = R.set_deterministic(True)
    try:
        want = {}
        full = g.flame_param
        for t in (3, 7, 0):
            _zero(g)
            loss, img, radii, vsp = _step(g, cam, bg, target, t)
            want[t] = dict(loss=loss.clone(), img=img.clone(), radii=radii.clone(), vsp=vsp.grad.clone(),
                           leaves=[getattr(g, n).grad.clone() for n in _LEAVES],
                           expr=full["expr"].grad[t].clone(), jaw=full["jaw_pose"].grad[t].clone(), n=R.last_forward_info()["num_rendered"])
        feeder = FlameRowFeeder(full, requires_grad=True)
        g.flame_param = feeder.static_param
        step = GraphedStep(lambda: _step(g, cam, bg, target, 0), before_capture=lambda: _zero(g))
        for t in (3, 7, 0, 7):
            feeder.feed(t)
            loss, img, radii, vsp = step.replay()
            torch.cuda.synchronize()
            step.check()
            w = want[t]
            assert step.instances() == [w["n"]]
            assert torch.equal(img, w["img"]) and torch.equal(radii, w["radii"]) and torch.equal(loss, w["loss"]), t
            assert torch.equal(vsp.grad, w["vsp"]), t
            for n, a in zip(_LEAVES, w["leaves"]):
                assert torch.equal(getattr(g, n).grad, a), (t, n)
            # (the FLAME backward sums with float atomics: equal up to summation order)
            for k, ref in (("expr", w["expr"]), ("jaw_pose", w["jaw"])):
                got = feeder.static_param[k].grad[0]
                assert float((got - ref).abs().max()) <= 1e-5 * float(ref.abs().max()) + 1e-12, (t, k)
        assert step.replays == 4 and step.capacity >= 4 * max(v["n"] for v in want.values()) - (1 << 16)
        step.close()
    finally:
        R.set_deterministic(prev)


def test_three_frames_per_recording_fed_from_a_device_side_schedule():
    """K frames in ONE recording, each preceded by the feeder's kernel (include/gab.h: gab_feed_row) that takes the next timestep of a DEVICE
    schedule: two replays walk six schedule entries on their own (no host call between frames), wrap around the schedule, and every frame has
    the eager step's image bits and loss; seek() repositions the walk."""
    from gaussianavatars_amd import rasterizer as R
    from gaussianavatars_amd.graphs import FlameRowFeeder, GraphedStep

    dev = _dev()
    g, cam = _scene(dev)
    bg = torch.ones(3, device=dev)
    target = torch.full((3, 176, 208), 0.5, device=dev)
    prev = R.set_deterministic(True)
    try:
        full = g.flame_param
        sched = [3, 7, 0, 11, 5]
        want = {}
        for t in sched:
            _zero(g)
            loss, img, _, _ = _step(g, cam, bg, target, t)
            want[t] = (loss.clone(), img.clone(), [getattr(g, n).grad.clone() for n in _LEAVES])
        feeder = FlameRowFeeder(full, requires_grad=True)
        feeder.set_schedule(sched)
        g.flame_param = feeder.static_param

        def three():
            out = []
            for j in range(3):
                if j:
                    _zero(g)
                feeder.feed_next()
                loss, img, _, _ = _step(g, cam, bg, target, 0)
                out += [loss, img]
            return tuple(out)

        step = GraphedStep(three, before_capture=lambda: _zero(g))
        assert len(step.instances()) == 3          # one count slot per recorded forward
        feeder.seek(0)
        seen = []
        for _ in range(2):
            out = step.replay()
            torch.cuda.synchronize()
            step.check()
            seen += [(out[2 * j].clone(), out[2 * j + 1].clone()) for j in range(3)]
        for (loss, img), t in zip(seen, [3, 7, 0, 11, 5, 3]):
            assert torch.equal(img, want[t][1]) and torch.equal(loss, want[t][0]), t
        for n, a in zip(_LEAVES, want[3][2]):      # the gradients left behind are the last frame's
            assert torch.equal(getattr(g, n).grad, a), n
        assert int(feeder.cursor.item()) == 6 % len(sched)   # (the device cursor lives in [0, len): it cannot overflow however long the run)
        feeder.seek(3)
        out = step.replay()
        torch.cuda.synchronize()
        assert torch.equal(out[1], want[11][1]) and torch.equal(out[5], want[3][1])
        step.close()
    finally:
        R.set_deterministic(prev)


def test_overflowing_frame_is_reported_and_recapture_recovers():
    from gaussianavatars_amd import rasterizer as R
    from gaussianavatars_amd.graphs import CapacityOverflow, FlameRowFeeder, GraphedStep

    dev = _dev()
    g, cam = _scene(dev, n=60000)
    bg = torch.ones(3, device=dev)
    target = torch.full((3, 176, 208), 0.5, device=dev)
    feeder = FlameRowFeeder(g.flame_param, requires_grad=True)
    g.flame_param = feeder.static_param
    _zero(g)
    _step(g, cam, bg, target, 0)
    need = R.last_forward_info()["num_rendered"]
    assert need > 2 * (1 << 16), "the scene must need more than the smallest capacity the library hands out"
    step = GraphedStep(lambda: _step(g, cam, bg, target, 0), before_capture=lambda: _zero(g), headroom=0.2)
    assert step.capacity < need
    step.replay()
    torch.cuda.synchronize()          # an overflowing frame skips its kernels: nothing faults
    assert step.instances() == [need]
    with pytest.raises(CapacityOverflow):
        step.check()
    step.recapture(headroom=2.0)
    feeder.feed(5)
    loss, img, _, _ = step.replay()
    torch.cuda.synchronize()
    step.check()
    assert step.capacity >= need and float(loss) > 0 and bool(torch.isfinite(img).all())
    g.flame_param = feeder.static_param
    _zero(g)
    feeder.feed(5)
    eager = _step(g, cam, bg, target, 0)
    assert torch.equal(eager[1], img)
    step.close()


def test_an_overflow_in_an_earlier_replay_stays_reported():
    """The count slot of a recording only shows its NEWEST replay; a frame that overflowed three replays ago skipped its binning and blend
    (stale image, zero gradients) all the same.  The device leaves a sticky mark (include/gsr.h: gsr_count_slot_overflow): check() after a
    run of replays reports it even though the last frame fitted, and clears it."""
    from gaussianavatars_amd.graphs import CapacityOverflow, FlameRowFeeder, GraphedStep

    dev = _dev()
    g, cam = _scene(dev, n=60000)
    bg = torch.ones(3, device=dev)
    target = torch.full((3, 176, 208), 0.5, device=dev)
    feeder = FlameRowFeeder(g.flame_param, requires_grad=True)
    g.flame_param = feeder.static_param
    step = GraphedStep(lambda: _step(g, cam, bg, target, 0), before_capture=lambda: _zero(g), headroom=1.5)
    need = step.warm_instances
    step.replay()
    torch.cuda.synchronize()
    step.check()                                    # fits
    with torch.no_grad():
        g._scaling.add_(1.0)                        # e-times larger splats: several times the tile instances (the recording reads the leaf in place)
    step.replay()
    with torch.no_grad():
        g._scaling.sub_(1.0)
    for _ in range(3):
        step.replay()                               # three fitting frames on top
    torch.cuda.synchronize()
    assert max(step.instances()) <= step.capacity   # the newest count is innocent ...
    assert abs(max(step.instances()) - need) <= need // 10
    with pytest.raises(CapacityOverflow):
        step.check()                                # ... the mark is not
    step.check()                                    # reported once, then cleared
    step.close()


def test_deferred_count_outside_a_graph():
    """The deferred forward by itself (run-ahead without a recording): same image, the count arrives in the slot."""
    from gaussianavatars_amd import rasterizer as R
    from gaussianavatars_amd.gaussian_renderer import render

    dev = _dev()
    g, cam = _scene(dev)
    bg = torch.ones(3, device=dev)
    free0 = len(R._free_slots)
    with torch.no_grad():
        g.select_mesh_by_timestep(2)
        ref = render(cam, g, _Pipe, bg)["render"].clone()
        need = R.last_forward_info()["num_rendered"]
        with R.deferred_count(2 * need) as d:
            img = render(cam, g, _Pipe, bg)["render"]
        torch.cuda.synchronize()
    assert d.counts() == [need] and torch.equal(img, ref)
    assert R.last_forward_info()["num_rendered"] == -1
    d.release()
    assert len(R._free_slots) == free0


def test_recorded_lanes_on_separate_streams_do_not_disturb_each_other():
    """Frame parallelism inside one GPU (bench.py --graph --streams S): several recorded steps, each with its own model replica,
    replayed concurrently on separate streams.  Every lane's last frame must equal the same frame run alone (image bits; with
    the deterministic backward the splat-gradient bits): the native libraries keep no state shared between calls."""
    from gaussianavatars_amd import rasterizer as R
    from gaussianavatars_amd.graphs import FlameRowFeeder, GraphedStep

    dev = _dev()
    bg = torch.ones(3, device=dev)
    target = torch.full((3, 176, 208), 0.5, device=dev)
    prev = R.set_deterministic(True)
    try:
        g, cam = _scene(dev)
        want = {}
        for t in (1, 4, 6, 9):
            _zero(g)
            loss, img, radii, vsp = _step(g, cam, bg, target, t)
            want[t] = dict(img=img.clone(), loss=loss.clone(), leaves=[getattr(g, n).grad.clone() for n in _LEAVES])
        lanes = []
        for k in range(3):
            gk, _ = _scene(dev)     # same seeds: identical replicas
            feeder = FlameRowFeeder(gk.flame_param, requires_grad=True)
            gk.flame_param = feeder.static_param
            step = GraphedStep(lambda gk=gk: _step(gk, cam, bg, target, 0), before_capture=lambda gk=gk: _zero(gk))
            lanes.append((gk, feeder, step, torch.cuda.Stream(dev)))
        order = [(0, 1), (1, 4), (2, 6), (0, 9), (1, 1), (2, 4), (0, 6), (1, 9), (2, 1)]     # (lane, timestep)
        cur = torch.cuda.current_stream()
        for _ in range(5):          # several rounds back to back: the lanes really run side by side
            for k, t in order:
                gk, feeder, step, st = lanes[k]
                with torch.cuda.stream(st):
                    feeder.feed(t)
                    step.replay()
        for _, _, _, st in lanes:
            cur.wait_stream(st)
        torch.cuda.synchronize()
        last = {0: 6, 1: 9, 2: 1}
        for k, (gk, feeder, step, st) in enumerate(lanes):
            step.check()
            loss, img, radii, vsp = step.out
            w = want[last[k]]
            assert torch.equal(img, w["img"]) and torch.equal(loss, w["loss"]), k
            for n, a in zip(_LEAVES, w["leaves"]):
                assert torch.equal(getattr(gk, n).grad, a), (k, n)
            step.close()
    finally:
        R.set_deterministic(prev)


def test_recorded_step_in_a_training_loop_with_an_optimiser():
    """INTEGRATION.md 3c as written: feed, replay, optimizer.step() on the static .grad tensors.  Three Adam iterations through the
    recorded step leave the parameters where three eager iterations leave them (deterministic backward: bit for bit for the
    splat leaves)."""
    from gaussianavatars_amd import rasterizer as R
    from gaussianavatars_amd.graphs import FlameRowFeeder, GraphedStep

    dev = _dev()
    bg = torch.ones(3, device=dev)
    target = torch.full((3, 176, 208), 0.5, device=dev)
    frames = (2, 5, 9)
    prev = R.set_deterministic(True)
    try:
        def params(g):
            return [getattr(g, n) for n in _LEAVES]

        # eager
        g, cam = _scene(dev)
        opt = torch.optim.Adam(params(g), lr=1e-3)
        for t in frames:
            _zero(g)
            _step(g, cam, bg, target, t)
            opt.step()
        want = [p.detach().clone() for p in params(g)]
        # recorded
        g, cam = _scene(dev)
        feeder = FlameRowFeeder(g.flame_param, requires_grad=True)
        g.flame_param = feeder.static_param
        opt = torch.optim.Adam(params(g), lr=1e-3)
        step = GraphedStep(lambda: _step(g, cam, bg, target, 0), before_capture=lambda: _zero(g))
        for t in frames:
            feeder.feed(t)
            step.replay()
            opt.step()              # reads the static .grad tensors the replay just rewrote
        torch.cuda.synchronize()
        step.check()
        for n, p, w in zip(_LEAVES, params(g), want):
            assert torch.equal(p.detach(), w), n
        step.close()
    finally:
        R.set_deterministic(prev)


def test_lanes_on_shared_leaves_accumulate_like_sequential_backwards():
    """graphs.shared_lane_model + accumulate_lane_grads (round 4): two recorded lanes on two streams read ONE set of splat parameters and
    write their own gradients; added up in lane order they are, with the deterministic backward, the BITS of loss_a.backward();
    loss_b.backward() on the one model (gradient accumulation over the lanes' frames), FLAME rows included; an optimiser step on the
    shared storage is seen by every lane's next replay."""
    from gaussianavatars_amd import rasterizer as R
    from gaussianavatars_amd.graphs import FlameRowFeeder, GraphedStep, accumulate_lane_grads, shared_lane_model

    dev = _dev()
    g, cam = _scene(dev)
    bg = torch.ones(3, device=dev)
    target = torch.full((3, 176, 208), 0.5, device=dev)
    prev = R.set_deterministic(True)
    try:
        frames = (2, 9)
        # ---- the reference result: two frames accumulated on ONE model, eagerly
        _zero(g)
        want_loss = []
        for t in frames:
            from gaussianavatars_amd.graphs import release_mesh

            release_mesh(g)
            want_loss.append(_step(g, cam, bg, target, t)[0].clone())
        want = [getattr(g, n).grad.clone() for n in _LEAVES]
        want_flame = {k: v.grad.clone() for k, v in g.flame_param.items() if v.requires_grad}
        _zero(g)
        # ---- two lanes over the same storage, each with its own recording, feeder and stream
        lanes = []
        for t in frames:
            m = shared_lane_model(g)
            feeder = FlameRowFeeder(m.flame_param, requires_grad=True)
            full = m.flame_param
            m.flame_param = feeder.static_param
            step = GraphedStep(lambda m=m: _step(m, cam, bg, target, 0)[0], before_capture=lambda m=m: _zero(m))
            lanes.append(dict(m=m, feeder=feeder, step=step, stream=torch.cuda.Stream(dev), t=t, full=full))
        cur = torch.cuda.current_stream(dev)
        outs = []
        for ln in lanes:
            ln["stream"].wait_stream(cur)
            with torch.cuda.stream(ln["stream"]):
                ln["feeder"].feed(ln["t"])
                outs.append(ln["step"].replay())
        for ln in lanes:
            cur.wait_stream(ln["stream"])
        # the lanes' FLAME gradients are one-row tables: scatter them to the rows of the frames they rendered before adding up
        accumulate_lane_grads(g, [ln["m"] for ln in lanes], flame=False)
        torch.cuda.synchronize()
        for ln, wl, out in zip(lanes, want_loss, outs):
            ln["step"].check()
            assert torch.equal(out, wl)
        for n, w in zip(_LEAVES, want):
            assert torch.equal(getattr(g, n).grad, w), n
        for k, w in want_flame.items():
            if k not in FlameRowFeeder.ROWS:
                continue
            got = torch.zeros_like(w)
            for ln in lanes:
                got[ln["t"]] += ln["m"].flame_param[k].grad.reshape(-1)
            # (the binding kernels' reductions run over a (T, k) table in the eager frames and over the lane's one-row table here: same sums, not the same order)
            assert torch.allclose(got, w, rtol=1e-4, atol=1e-7 * float(w.abs().max())), k
        # ---- one optimiser step on the shared storage: the lanes' next replays render the moved splats
        with torch.no_grad():
            g._xyz.add_(0.01 * torch.sign(g._xyz.grad))
        _zero(g)
        from gaussianavatars_amd.graphs import release_mesh

        release_mesh(g)
        ref_loss = _step(g, cam, bg, target, frames[0])[0].clone()
        with torch.cuda.stream(lanes[0]["stream"]):
            lanes[0]["stream"].wait_stream(cur)
            lanes[0]["feeder"].feed(frames[0])
            out = lanes[0]["step"].replay()
        torch.cuda.synchronize()
        assert torch.equal(out, ref_loss) and not torch.equal(out, want_loss[0])
        for ln in lanes:
            ln["step"].close()
    finally:
        R.set_deterministic(prev)
