"""Stand-in for torch_geometric.nn: MessagePassing.propagate (gather by
edge_index[0] for `_j`, by edge_index[1] for `_i`, aggregate by edge_index[1])."""
import inspect
import torch


class MessagePassing(torch.nn.Module):
    def __init__(self, aggr="add", node_dim=0, flow="source_to_target"):
        super().__init__()
        assert flow == "source_to_target"
        self.aggr = aggr
        self.node_dim = node_dim

    def propagate(self, edge_index, size=None, **kwargs):
        params = list(inspect.signature(self.message).parameters)
        args = []
        dim_size = None
        for p in params:
            if p.endswith("_j") and p[:-2] in kwargs:
                src = kwargs[p[:-2]]
                dim_size = src.size(self.node_dim)
                args.append(src.index_select(self.node_dim, edge_index[0]))
            elif p.endswith("_i") and p[:-2] in kwargs:
                src = kwargs[p[:-2]]
                dim_size = src.size(self.node_dim)
                args.append(src.index_select(self.node_dim, edge_index[1]))
            else:
                args.append(kwargs[p])
        msg = self.message(*args)
        out = self.aggregate(msg, edge_index[1], None, dim_size)
        return self.update(out)

    def update(self, inputs):
        return inputs


def radius_graph(*a, **k):  # import-only placeholder (HermNet/data.py:9)
    raise RuntimeError("torch_cluster is not available")
