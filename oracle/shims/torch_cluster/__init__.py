"""TEST INFRASTRUCTURE ONLY (oracle shim) - not part of the product path.

Stand-in for the un-vendored `torch_cluster.radius_graph` (call site:
/root/reference/src/ViSNet/model/utils.py:260-266).  Published semantics
restated: pairs (j -> i) inside the same batch id with squared distance
STRICTLY < r^2, self loops kept when loop=True, at most `max_num_neighbors`
sources per target.  When a target has more candidates the CUDA kernel of
torch_cluster scans sources in ascending index and keeps the first
`max_num_neighbors`; this shim does the same (the CPU kd-tree path of the
real wheel keeps an implementation-defined subset - see SURVEY.md 8a/a4).
Returns int64 [2, E] = [source j ; target i], grouped by target ascending.
"""
import torch


def radius_graph(x, r, batch=None, loop=False, max_num_neighbors=32,
                 flow="source_to_target", num_workers=1):
    assert flow == "source_to_target"
    n = x.size(0)
    if batch is None:
        batch = torch.zeros(n, dtype=torch.long, device=x.device)
    src_all, tgt_all = [], []
    # dense per-fragment evaluation (fragments are tiny)
    uniq, counts = torch.unique_consecutive(batch, return_counts=True)
    start = 0
    for c in counts.tolist():
        p = x[start:start + c].detach()
        diff = p[:, None, :] - p[None, :, :]
        d2 = (diff * diff).sum(-1)
        adj = d2 < (r * r)
        if not loop:
            adj = adj & ~torch.eye(c, dtype=torch.bool, device=x.device)
        # adj[i, j]: j is a source for target i.  keep first max_nb per row
        rank = torch.cumsum(adj.to(torch.long), dim=1)
        adj = adj & (rank <= max_num_neighbors)
        ti, sj = torch.nonzero(adj, as_tuple=True)  # row-major: by target asc
        src_all.append(sj + start)
        tgt_all.append(ti + start)
        start += c
    if not src_all:
        return torch.zeros(2, 0, dtype=torch.long, device=x.device)
    return torch.stack([torch.cat(src_all), torch.cat(tgt_all)])
