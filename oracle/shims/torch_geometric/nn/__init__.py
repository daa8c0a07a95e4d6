"""TEST INFRASTRUCTURE ONLY (oracle shim) - not part of the product path.

Stand-in for `torch_geometric.nn.MessagePassing` restricted to what the
reference uses (ViSNet/model/visnet_block.py:145-312, utils.py:279-341):
`propagate` gathers `<name>_i` by edge_index[1] (target) and `<name>_j` by
edge_index[0] (source) (flow='source_to_target'), calls `message`, then
`aggregate(inputs, index=edge_index[1], ...)`, then `update`.
`edge_updater` gathers the same way and calls `edge_update`.
`jittable()` returns self.
"""
import inspect

import torch


class MessagePassing(torch.nn.Module):
    def __init__(self, aggr="add", flow="source_to_target", node_dim=-2):
        super().__init__()
        self.aggr = aggr
        self.flow = flow
        self.node_dim = node_dim

    def jittable(self, *a, **k):
        return self

    def _collect(self, fn, edge_index, kwargs):
        out = {}
        for name in inspect.signature(fn).parameters:
            if name.endswith("_i") or name.endswith("_j"):
                base = name[:-2]
                if base in kwargs:
                    idx = edge_index[1] if name.endswith("_i") else edge_index[0]
                    out[name] = kwargs[base].index_select(self.node_dim, idx)
                    continue
            if name in kwargs:
                out[name] = kwargs[name]
        return out

    def propagate(self, edge_index, size=None, **kwargs):
        dim_size = None
        for v in kwargs.values():
            if isinstance(v, torch.Tensor) and v.dim() > 0:
                dim_size = v.size(self.node_dim)
                break
        msg = self.message(**self._collect(self.message, edge_index, kwargs))
        params = list(inspect.signature(self.aggregate).parameters)
        index = edge_index[1]
        if params == ["features", "index"]:
            agg = self.aggregate(msg, index)
        elif params == ["features", "index", "ptr", "dim_size"]:
            agg = self.aggregate(msg, index, None, dim_size)
        else:
            agg = self.aggregate(msg, index, dim_size=dim_size)
        return self.update(agg)

    def aggregate(self, inputs, index, ptr=None, dim_size=None):
        dim = self.node_dim if self.node_dim >= 0 else inputs.dim() + self.node_dim
        shape = list(inputs.shape)
        shape[dim] = dim_size
        out = torch.zeros(shape, dtype=inputs.dtype, device=inputs.device)
        return out.index_add(dim, index, inputs)

    def update(self, inputs):
        return inputs

    def edge_updater(self, edge_index, **kwargs):
        return self.edge_update(**self._collect(self.edge_update, edge_index, kwargs))
