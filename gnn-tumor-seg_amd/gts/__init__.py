"""gts — MI355X-native graph ops for the GNN-Tumor-Seg node-classification path.

Replaces the slice of DGL the reference reaches (SAGEConv, GATConv, from_networkx, batch,
graph.to/in_degrees/ndata) with an int32-CSR graph type and hand-written gfx950 kernels
behind a C ABI (include/gts_hip.h).  HIP-only: no CPU or PyTorch fallback.
"""
from .graph import Graph, batch, from_networkx, graph  # noqa: F401
from ._lib import GtsError  # noqa: F401

__all__ = ["Graph", "batch", "from_networkx", "graph", "GtsError"]
