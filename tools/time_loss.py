"""Fused L1+SSIM (include/gls.h) vs the reference's composed-torch formulation (utils/loss_utils.py) on a 3x802x550 pair."""
import sys, time, torch, torch.nn.functional as F
sys.path.insert(0, '.')
from gaussianavatars_amd import loss
from oracle import loss_oracle as LO
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
img = torch.rand(3, 802, 550, generator=g).to(dev).requires_grad_(True)
gt = torch.rand(3, 802, 550, generator=g).to(dev)
w1 = torch.from_numpy(LO.window_1d()).to(dev)
win = (w1[:, None] * w1[None, :]).expand(3, 1, 11, 11).contiguous()
def torch_loss(a, b, lam=0.2):
    conv = lambda x: F.conv2d(x[None], win, padding=5, groups=3)[0]
    mu1, mu2 = conv(a), conv(b)
    s1, s2, s12 = conv(a * a) - mu1 * mu1, conv(b * b) - mu2 * mu2, conv(a * b) - mu1 * mu2
    ssim = (((2 * mu1 * mu2 + LO.C1) * (2 * s12 + LO.C2)) / ((mu1 * mu1 + mu2 * mu2 + LO.C1) * (s1 + s2 + LO.C2))).mean()
    return (1 - lam) * (a - b).abs().mean() + lam * (1 - ssim)
def fused_loss(a, b, lam=0.2):
    l1, ss = loss.l1_ssim(a, b)
    return (1 - lam) * l1 + lam * (1 - ss)
for name, fn in (("torch composed", torch_loss), ("fused HIP", fused_loss)):
    for _ in range(10):
        img.grad = None; fn(img, gt).backward()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 200
    for _ in range(n):
        img.grad = None; fn(img, gt).backward()
    torch.cuda.synchronize()
    print(f"{name:16s} {1e6 * (time.perf_counter() - t0) / n:8.1f} us per fwd+bwd")
