"""Graph container and neighbour-list transform for the hot path.

The reference passes `torch_geometric.data.Data` objects around
(`HermNet/hermnet.py:37,118-152`, `HermNet/utils.py:11-24`) and touches them by
attribute (`data.pos`), by item (`data['atomic_number']`, `hermnet.py:53`) and
through `.get(key)` (`hermnet.py:138`).  PyG is not available on the MI355X
image, so the build owns a minimal duck-type compatible container; a real PyG
`Data` works equally well with `HVNet.forward` because only those three access
forms are used.
"""
import torch

from .neighbor import neighbor_search  # noqa: F401  (re-export, `HermNet/data.py:14`)


class Data(object):
    """Attribute bag of tensors; unset keys read as None (PyG behaviour for
    `pos`, `batch`, `edge_index`, ...)."""

    def __init__(self, **kwargs):
        object.__setattr__(self, "_store", dict(kwargs))

    def __getattr__(self, key):
        if key.startswith("__"):
            raise AttributeError(key)
        return object.__getattribute__(self, "_store").get(key, None)

    def __setattr__(self, key, value):
        self._store[key] = value

    def __delattr__(self, key):
        self._store.pop(key, None)

    def __getitem__(self, key):
        return self._store[key]

    def __setitem__(self, key, value):
        self._store[key] = value

    def __contains__(self, key):
        return key in self._store

    def get(self, key, default=None):
        return self._store.get(key, default)

    def keys(self):
        return list(self._store.keys())

    def __iter__(self):
        for k in list(self._store.keys()):
            yield k, self._store[k]

    def __copy__(self):
        return Data(**self._store)

    @property
    def num_nodes(self):
        for k in ("pos", "atomic_number", "x"):
            v = self._store.get(k)
            if v is not None:
                return int(v.size(0))
        return 0

    @property
    def num_edges(self):
        ei = self._store.get("edge_index")
        return 0 if ei is None else int(ei.size(1))

    @property
    def num_graphs(self):
        b = self._store.get("batch")
        return 1 if b is None or b.numel() == 0 else int(b.max()) + 1

    def to(self, device, non_blocking=False):
        for k, v in self._store.items():
            if isinstance(v, torch.Tensor):
                self._store[k] = v.to(device, non_blocking=non_blocking)
        return self

    def __repr__(self):
        parts = []
        for k, v in self._store.items():
            parts.append("%s=%s" % (k, list(v.shape) if isinstance(v, torch.Tensor) else v))
        return "Data(%s)" % ", ".join(parts)


def transform(data, rc, reference_compat=False):
    """`HermNet/data.py:27-35`: attach `edge_index` (and `edge_shift`) for cutoff rc."""
    assert data.pos is not None
    if data.get("cell") is None:
        data.edge_index = neighbor_search(data.pos, rc)
    else:
        data.edge_index, data.edge_shift = neighbor_search(
            data.pos, rc, data.cell, reference_compat=reference_compat)
    if data.get("batch") is None:
        data.batch = torch.zeros(data.pos.size(0), dtype=torch.long, device=data.pos.device)
    return data
