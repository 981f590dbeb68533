"""Where does the HOST spend its time per frame?  cProfile over the benchmark step (GPU work is asynchronous)."""
import cProfile, pstats, sys, torch
sys.path.insert(0, '.')
import bench
dev = torch.device('cuda:0')
g, cam = bench.build_scene(dev, 100_000, 3, 550, 802, 4, "fused", True)
bg = torch.ones(3, device=dev); target = torch.ones(3, 802, 550, device=dev)
def run(n):
    for i in range(n):
        bench.one_step(g, cam, bg, target, i % 4, True); bench.zero_grads(g)
run(30); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable(); run(300); pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(28)
st.sort_stats('cumtime').print_stats(22)
