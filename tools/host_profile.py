"""Host time of the frame loop (main thread + autograd's device thread):  python tools/host_profile.py [steps] [--small] [--profile]
   default scene: BASELINE configs[2] (the loop is then paced by max(host, GPU): the count wait couples them once per frame);
   --small: 10 144 small splats at 401x275 -- the GPU side is a few launch floors, so the loop time IS the host time per step;
   --profile: cProfile of the main thread on top.
Both host sides are timed: the compiled one (gaa_host.so, the default) and the Python twins (GAA_NATIVE_HOST=0's path)."""
import cProfile, pstats, sys, time, io
import torch
sys.path.insert(0, '.')
import bench
from gaussianavatars_amd import _host
from gaussianavatars_amd.frame_parallel import pin_to_gpu_numa_node
from gaussianavatars_amd.loss import install_backward_seed
pin_to_gpu_numa_node(0)
install_backward_seed()
dev = torch.device('cuda:0')
small = "--small" in sys.argv
args = [a for a in sys.argv[1:] if not a.startswith("--")]
N = int(args[0]) if args else 300
n_splats, W, H = (10_144, 275, 401) if small else (100_000, 550, 802)
g, cam = bench.build_scene(dev, n_splats, 3, W, H, 4, "fused", True)
if small:
    with torch.no_grad():
        g._scaling -= 1.2      # small footprints: a few instances per splat, so that the kernels are launch floors and the host paces the loop
bg = torch.ones(3, device=dev); target = torch.ones(3, H, W, device=dev)
def run(n):
    for i in range(n):
        bench.one_step(g, cam, bg, target, i % 4, True); bench.zero_grads(g)
for native in (True, False):
    _host.set_enabled(native)
    run(30); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); run(N); torch.cuda.synchronize(); t1 = time.perf_counter()
        best = min(best, (t1 - t0) / N * 1e6)
    print("%s host, %d splats %dx%d: %.1f us/step (best of 5 runs of %d steps, device synchronised at the end)" % ("compiled" if native else "python  ", n_splats, W, H, best, N))
    if "--profile" in sys.argv:
        pr = cProfile.Profile(); pr.enable(); run(N); pr.disable(); torch.cuda.synchronize()
        s = io.StringIO(); ps = pstats.Stats(pr, stream=s).sort_stats("tottime"); ps.print_stats(30)
        for line in s.getvalue().splitlines():
            print(line[:170])
