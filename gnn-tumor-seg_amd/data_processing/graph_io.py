"""Graph (de)serialisation and the node -> voxel projection.

Counterpart of /root/reference/data_processing/graph_io.py:21-37.  The projection runs as
the K12 HIP kernel (gts.ops.project_rows); inputs may be numpy arrays (what the reference's
callers pass) or torch tensors already on the GPU, and the result comes back in kind.
"""
import json

import networkx as nx
import numpy as np
import torch

from gts import ops


def _to_device(array_or_tensor, device):
    if isinstance(array_or_tensor, torch.Tensor):
        return array_or_tensor.to(device)
    return torch.from_numpy(np.ascontiguousarray(array_or_tensor)).to(device)


def _hip_device():
    if not torch.cuda.is_available():
        raise RuntimeError("node->voxel projection runs on the MI355X HIP path only: no GPU visible")
    return torch.device("cuda", torch.cuda.current_device())


def project_nodes_to_img(svs, node_labels):
    """Give every voxel the label of its supervoxel; voxels with id -1 (background) get 0.

    Same result, shape and dtype as `np.append(node_labels, 0)[svs]` (reference
    graph_io.py:21-24): integer labels come back as int64, floating ones as float64."""
    as_numpy = not isinstance(svs, torch.Tensor)
    lab_dtype = node_labels.dtype if not isinstance(node_labels, torch.Tensor) else \
        np.dtype(str(node_labels.dtype).replace("torch.", ""))
    out_dtype = np.result_type(lab_dtype, np.int64)
    torch_dtype = torch.int64 if out_dtype == np.int64 else torch.float64
    dev = svs.device if isinstance(svs, torch.Tensor) and svs.is_cuda else _hip_device()
    svs_d = _to_device(svs, dev)
    if svs_d.dtype != torch.int16:
        if svs_d.numel() and (int(svs_d.min()) < -32768 or int(svs_d.max()) > 32767):
            raise ValueError("supervoxel ids do not fit int16")
        svs_d = svs_d.to(torch.int16)
    table = _to_device(node_labels, dev).reshape(-1).to(torch_dtype)
    bg = torch.zeros(1, dtype=torch_dtype, device=dev)
    out = ops.project_rows(svs_d, table, bg)
    return out.cpu().numpy() if as_numpy else out


def project_node_logits_to_img(svs, node_logits, background_logits):
    """Voxel logits [*svs.shape, C] = concat(node_logits, background_logits)[svs]
    (reference scripts/generate_gnn_predictions.py:58-61).  Stays on the GPU and in fp32
    when given tensors; numpy in -> numpy float64 out, as numpy's promotion gives there."""
    as_numpy = not isinstance(node_logits, torch.Tensor)
    dev = node_logits.device if (not as_numpy and node_logits.is_cuda) else _hip_device()
    table = _to_device(node_logits, dev).to(torch.float32).contiguous()
    if table.dim() != 2 or table.shape[1] * 4 not in (4, 8, 16):
        raise ValueError("node_logits must be [N, C] with C in {1, 2, 4}")
    bg = torch.tensor(background_logits, dtype=torch.float32, device=dev).reshape(-1)
    svs_d = _to_device(svs, dev)
    if svs_d.dtype != torch.int16:
        svs_d = svs_d.to(torch.int16)
    out = ops.project_rows(svs_d, table, bg)
    return out.cpu().numpy().astype(np.float64) if as_numpy else out


def save_networkx_graph(G, fp):
    with open(fp, "w") as f:
        json.dump(nx.readwrite.json_graph.node_link_data(G), f)


def load_networkx_graph(fp):
    with open(fp, "r") as f:
        return nx.readwrite.json_graph.node_link_graph(json.load(f))
