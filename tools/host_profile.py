"""cProfile of the frame loop's host side (main thread + the autograd thread's Function.backward bodies):  python tools/host_profile.py [steps]"""
import cProfile, pstats, sys, time, io
import torch
sys.path.insert(0, '.')
import bench
from gaussianavatars_amd.frame_parallel import pin_to_gpu_numa_node
pin_to_gpu_numa_node(0)
dev = torch.device('cuda:0')
g, cam = bench.build_scene(dev, 100_000, 3, 550, 802, 4, "fused", True)
bg = torch.ones(3, device=dev); target = torch.ones(3, 802, 550, device=dev)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
def run(n):
    for i in range(n):
        bench.one_step(g, cam, bg, target, i % 4, True); bench.zero_grads(g)
run(30); torch.cuda.synchronize()
t0 = time.perf_counter(); run(N); t1 = time.perf_counter(); torch.cuda.synchronize()
print("unprofiled loop %.1f us/step host" % ((t1 - t0) / N * 1e6))
pr = cProfile.Profile(); pr.enable(); run(N); pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); ps = pstats.Stats(pr, stream=s).sort_stats("tottime"); ps.print_stats(45)
for line in s.getvalue().splitlines():
    print(line[:170])
