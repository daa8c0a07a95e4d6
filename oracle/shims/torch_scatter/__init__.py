"""TEST INFRASTRUCTURE ONLY (oracle shim) - not part of the product path.

Minimal stand-in for the un-vendored `torch_scatter` wheel so that the
reference's own model source (/root/reference/src/ViSNet/model/*.py) can be
imported in this container.  Semantics restated from the published
torch_scatter API: `scatter(src, index, dim, dim_size, reduce='sum')` ==
`zeros(...).index_add_(dim, index, src)` (reduce='mean': divided by the per-index count, clamped to 1).

Call sites in the reference: ViSNet/model/visnet.py:146,
ViSNet/model/visnet_block.py:305-306, Calculators/combiner.py:39.
"""
import torch


def scatter(src, index, dim=0, out=None, dim_size=None, reduce="sum"):
    if reduce not in ("sum", "add", "mean"):
        raise NotImplementedError(f"oracle shim: reduce={reduce}")
    if dim < 0:
        dim = src.dim() + dim
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() > 0 else 0
    shape = list(src.shape)
    shape[dim] = dim_size
    res = torch.zeros(shape, dtype=src.dtype, device=src.device)
    res = res.index_add(dim, index, src)
    if reduce == "mean":  # torch_scatter: sum / count, the count clamped to >= 1
        cnt = torch.zeros(dim_size, dtype=src.dtype, device=src.device).index_add(0, index, torch.ones_like(index, dtype=src.dtype))
        view = [1] * res.dim()
        view[dim] = dim_size
        res = res / cnt.clamp(min=1).view(view)
    return res


def scatter_add(src, index, dim=0, out=None, dim_size=None):
    return scatter(src, index, dim=dim, dim_size=dim_size, reduce="sum")
