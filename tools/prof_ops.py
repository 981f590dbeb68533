import sys, torch
sys.path.insert(0, '.')
import bench
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda:0')
g, cam = bench.build_scene(dev, 100_000, 3, 550, 802, 4, "fused", True)
bg = torch.ones(3, device=dev); target = torch.ones(3, 802, 550, device=dev)
for i in range(5):
    bench.zero_grads(g); bench.one_step(g, cam, bg, target, i % 4, True)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for i in range(4):
        bench.zero_grads(g); bench.one_step(g, cam, bg, target, i % 4, True)
    torch.cuda.synchronize()
evs = [e for e in prof.events()]
import collections
cnt = collections.Counter()
for e in evs:
    n = e.name
    if 'emcpy' in n or 'emset' in n or 'fill' in n.lower() or 'copy' in n.lower():
        st = [s for s in (e.stack or []) if 'repo' in s][:3]
        cnt[(n[:50], tuple(st))] += 1
def chain(e):
    out = []
    while e is not None:
        out.append(e.name[:40]); e = e.cpu_parent
    return " <- ".join(out[:5])
cc = collections.Counter()
for e in evs:
    if e.name in ('aten::clone', 'aten::zeros', 'aten::zero_', 'aten::fill_', 'hipMemsetAsync', 'aten::add', 'aten::gt', 'aten::add_'):
        cc[chain(e)] += 1
for k, v in sorted(cc.items(), key=lambda kv: -kv[1]):
    print(v, k)
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print(v, k)
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=5, max_name_column_width=60))
