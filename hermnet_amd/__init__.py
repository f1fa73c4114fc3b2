"""hermnet_amd: MI355X-native engine for HermNet's heterogeneous relational
message-passing hot path."""
from .data import Data, neighbor_search, transform  # noqa: F401
from .hermnet import HVNet, HeteroVertexConv, HTNet, HeteroTriadicConv  # noqa: F401
from .rmnet import PaiNNModule, PaiNNMessage, PaiNNUpdate, ScaledSiLU, RadialBasis  # noqa: F401

__version__ = "0.1.0"
