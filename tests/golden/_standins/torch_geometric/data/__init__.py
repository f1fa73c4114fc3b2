"""Stand-in for torch_geometric.data: attribute-bag `Data`."""
import copy
import torch

_EDGE_KEYS = ("edge_embed", "edge_vec", "edge_dist", "edge_shift", "edge_attr", "edge_weight")


class Data(object):
    def __init__(self, **kwargs):
        object.__setattr__(self, "_store", {})
        for k, v in kwargs.items():
            self._store[k] = v

    # attribute / item / get access; unset well-known keys read as None
    def __getattr__(self, key):
        if key.startswith("__"):
            raise AttributeError(key)
        store = object.__getattribute__(self, "_store")
        return store.get(key, None)

    def __setattr__(self, key, value):
        self._store[key] = value

    def __getitem__(self, key):
        return self._store[key]

    def __setitem__(self, key, value):
        self._store[key] = value

    def get(self, key, default=None):
        return self._store.get(key, default)

    def __iter__(self):
        for k, v in list(self._store.items()):
            yield k, v

    def __copy__(self):
        out = Data()
        out._store.update(self._store)
        return out

    @property
    def num_edges(self):
        ei = self._store.get("edge_index")
        return 0 if ei is None else int(ei.size(1))

    @property
    def num_nodes(self):
        return int(self._store["pos"].size(0))

    def is_edge_attr(self, key):
        v = self._store.get(key)
        return (key in _EDGE_KEYS and isinstance(v, torch.Tensor)
                and v.dim() > 0 and v.size(0) == self.num_edges)

    def to(self, device):
        for k, v in self._store.items():
            if isinstance(v, torch.Tensor):
                self._store[k] = v.to(device)
        return self


class InMemoryDataset(object):  # import-only placeholder (HermNet/data.py:7)
    def __init__(self, *a, **k):
        pass


def download_url(*a, **k):  # import-only placeholder
    raise RuntimeError("no network")
