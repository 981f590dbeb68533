"""30 000 eager fwd+bwd steps on each bench scene (four timesteps in turn): gradients finite, memory flat, and the loss of a timestep the SAME BITS
every time it comes round (the forward is deterministic; profiles/r06_d_soak.txt)."""
import hashlib, sys, time, torch
sys.path.insert(0, '.')
import bench
dev = torch.device('cuda:0')
for scene in ("ellipsoid", "template_like"):
    bench.SCENE = scene
    g, cam = bench.build_scene(dev, 100_000, 3, 550, 802, 4, "fused", True)
    bg = torch.ones(3, device=dev); target = torch.ones(3, 802, 550, device=dev)
    losses = {}; t0 = time.time()
    for it in range(30000):
        bench.zero_grads(g)
        out = bench.one_step(g, cam, bg, target, it % 4, True)
        if it % 3000 < 4:
            losses.setdefault(it % 4, set()).add(float(out))   # the same frame, the same parameters: the same loss bits every time
        if it % 3000 == 0:
            torch.cuda.synchronize()
            gr = [p.grad for p in (g._xyz, g._features_dc, g._features_rest, g._opacity, g._scaling, g._rotation) if p.grad is not None]
            fin = all(bool(torch.isfinite(x).all()) for x in gr)
            print(scene, it, "finite" if fin else "NOT FINITE", "grad |xyz| %.6e" % float(gr[0].abs().sum()), "mem MB", torch.cuda.memory_allocated() >> 20, flush=True)
            assert fin
    torch.cuda.synchronize()
    print(scene, "distinct loss values per timestep:", {k: len(v) for k, v in losses.items()})
    print(scene, "30000 steps in %.1f s = %.0f steps/s" % (time.time() - t0, 30000 / (time.time() - t0)))
