"""Soak run of the eager frame loop WITH an optimiser in it (the leaves change in place every step, so every plan / cache keyed on version counters has to keep up):
   python tools/soak.py [seconds]   -- prints steps, loss, allocated / reserved memory every 300 steps; SOAK_OK when the loss stayed finite and the allocation did not grow.
   Round 5: 361 200 steps in 150 s on one MI355X, 336.2 MB allocated throughout, no replays."""
import sys, time, torch
sys.path.insert(0, '.')
import bench
from gaussianavatars_amd.frame_parallel import pin_to_gpu_numa_node
from gaussianavatars_amd.loss import install_backward_seed
from gaussianavatars_amd import rasterizer as R
pin_to_gpu_numa_node(0); install_backward_seed()
dev = torch.device('cuda:0')
g, cam = bench.build_scene(dev, 100_000, 3, 550, 802, 300, "fused", True)
bg = torch.ones(3, device=dev); target = torch.ones(3, 802, 550, device=dev)
opt = torch.optim.Adam([g._xyz, g._features_dc, g._features_rest, g._scaling, g._rotation, g._opacity], lr=1e-5)
t_end = time.time() + float(sys.argv[1]) if len(sys.argv) > 1 else time.time() + 60
n = 0; mem0 = None; bad = 0
while time.time() < t_end:
    for i in range(300):
        opt.zero_grad(set_to_none=True)
        loss = bench.one_step(g, cam, bg, target, i, True)
        opt.step()                      # the leaves change in place every step: plans / caches keyed on versions must keep up
        n += 1
    torch.cuda.synchronize()
    if not bool(torch.isfinite(loss)): bad += 1
    m = torch.cuda.memory_allocated()
    if mem0 is None: mem0 = m
    print("steps", n, "loss", float(loss), "allocated MB", round(m / 1e6, 1), "reserved MB", round(torch.cuda.memory_reserved() / 1e6, 1), "replays", R.last_forward_info().get("replays"), flush=True)
print("SOAK_OK" if bad == 0 and abs(m - mem0) < 64e6 else "SOAK_FAIL", n, mem0, m)
