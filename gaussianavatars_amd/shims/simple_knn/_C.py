"""distCUDA2(points (N,3) f32) -> (N,) f32: mean SQUARED distance to the 3 nearest other points.
Init-only and off the per-frame path, so this is chunked torch on whatever device the points live on."""
import torch


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    p = points.detach().float()
    n = p.shape[0]
    out = torch.empty(n, dtype=torch.float32, device=p.device)
    if n == 0:
        return out
    k = min(4, n)                      # self + 3 neighbours
    sq = (p * p).sum(1)
    chunk = max(1, min(n, (1 << 26) // max(n, 1)))   # ~256 MB of fp32 distances per chunk
    for s in range(0, n, chunk):
        q = p[s: s + chunk]
        d2 = (sq[s: s + chunk, None] + sq[None, :] - 2.0 * (q @ p.t())).clamp_min_(0.0)
        d2[torch.arange(q.shape[0], device=p.device), torch.arange(s, s + q.shape[0], device=p.device)] = 0.0
        near = torch.topk(d2, k, dim=1, largest=False).values[:, 1:]   # drop the point itself
        out[s: s + chunk] = near.sum(1) / 3.0 if near.shape[1] == 3 else near.sum(1) / max(near.shape[1], 1)
    return out
