"""Host time per frame, split by autograd node (forward and backward bodies timed with perf_counter on whichever thread runs them)
and by C-ABI call.  python tools/host_breakdown.py [steps]"""
import collections, ctypes, sys, time
import torch
sys.path.insert(0, '.')
import bench
from gaussianavatars_amd import _lib, binding, loss, rasterizer
from gaussianavatars_amd.frame_parallel import pin_to_gpu_numa_node

pin_to_gpu_numa_node(0)
acc = collections.defaultdict(float)
cnt = collections.defaultdict(int)

def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[name] += time.perf_counter() - t
            cnt[name] += 1
    return w

for mod in (binding, loss, rasterizer):
    for cname in dir(mod):
        c = getattr(mod, cname)
        if isinstance(c, type) and issubclass(c, torch.autograd.Function) and c is not torch.autograd.Function:
            c.forward = staticmethod(timed(cname + ".forward", c.forward))
            c.backward = staticmethod(timed(cname + ".backward", c.backward))

class Wrap:   # times every foreign function of a ctypes library
    def __init__(self, lib, tag):
        self._lib, self._tag, self._c = lib, tag, {}
    def __getattr__(self, n):
        f = self._c.get(n)
        if f is None:
            f = self._c[n] = timed("C " + n, getattr(self._lib, n))
        return f

for getter, tag in (("gsr", "gsr"), ("gab", "gab"), ("gls", "gls")):
    lib = getattr(_lib, getter)()
    w = Wrap(lib, tag)
    setattr(_lib, getter, (lambda w=w: w))
torch_empty = torch.empty
torch.empty = timed("torch.empty", torch_empty)

dev = torch.device('cuda:0')
g, cam = bench.build_scene(dev, 100_000, 3, 550, 802, 4, "fused", True)
bg = torch.ones(3, device=dev); target = torch.ones(3, 802, 550, device=dev)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
def run(n):
    for i in range(n):
        t = time.perf_counter(); bench.one_step(g, cam, bg, target, i % 4, True); acc["one_step"] += time.perf_counter() - t
        t = time.perf_counter(); bench.zero_grads(g); acc["zero_grads"] += time.perf_counter() - t
run(30); torch.cuda.synchronize(); acc.clear(); cnt.clear()
t0 = time.perf_counter(); run(N); t1 = time.perf_counter(); torch.cuda.synchronize()
print("loop %.1f us/step (host side, GPU not awaited)" % ((t1 - t0) / N * 1e6))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("%-38s %8.1f us/step  (%d calls/step)" % (k, v / N * 1e6, round(cnt[k] / N) if cnt[k] else 1))
