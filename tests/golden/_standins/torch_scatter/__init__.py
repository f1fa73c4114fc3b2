"""Stand-in for torch_scatter.scatter (sum / mean via scatter_add_)."""
import torch


def _broadcast(index, src, dim):
    if dim < 0:
        dim = src.dim() + dim
    if index.dim() == 1:
        for _ in range(0, dim):
            index = index.unsqueeze(0)
    for _ in range(index.dim(), src.dim()):
        index = index.unsqueeze(-1)
    return index.expand(src.size())


def scatter(src, index, dim=-1, out=None, dim_size=None, reduce="sum"):
    idx = _broadcast(index, src, dim)
    size = list(src.size())
    if dim_size is not None:
        size[dim] = dim_size
    elif idx.numel() == 0:
        size[dim] = 0
    else:
        size[dim] = int(idx.max()) + 1
    res = torch.zeros(size, dtype=src.dtype, device=src.device).scatter_add_(dim, idx, src)
    if reduce in ("sum", "add"):
        return res
    if reduce == "mean":
        ones = torch.ones(index.size(), dtype=src.dtype, device=src.device)
        cnt = torch.zeros(size[dim], dtype=src.dtype, device=src.device).scatter_add_(0, index, ones)
        cnt = cnt.clamp_(min=1)
        return res / _broadcast(cnt, res, dim)
    raise ValueError(reduce)
