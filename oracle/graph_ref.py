"""Oracle (test infrastructure): graph construction rules restated with numpy.

Follows the reference call sites
  data_processing/data_loader.py:67-83   (get_graph: networkx -> DGL graph, 'norm')
  data_processing/data_loader.py:165-169 (minibatch_graphs: dgl.batch + concat)
and the published behaviour of the third-party functions they call (DGL,
version unpinned, "DGL>=0.4" README.md:20; not importable here, so this part is
PARITY UNPINNED):

  dgl.from_networkx(g):  g = nx.convert_node_labels_to_integers(g, ordering='sorted')
                         g = g.to_directed(); COO = list(g.edges()) in that order.
  CSC (in-edge CSR) is built from the COO by a stable sort on destination;
  the out-edge CSR by a stable sort on source.
  dgl.batch(gs): node ids of graph j shifted by sum_{i<j} N_i, edges concatenated
                 in graph order, ndata concatenated.
"""
import numpy as np


class RefGraph:
    """COO in insertion order + in-edge CSR (by destination) + out-edge CSR (by source)."""

    def __init__(self, src, dst, n_nodes, batch_num_nodes=None):
        self.src = np.asarray(src, dtype=np.int64)
        self.dst = np.asarray(dst, dtype=np.int64)
        self.n = int(n_nodes)
        self.batch_num_nodes = list(batch_num_nodes) if batch_num_nodes is not None else [self.n]
        # in-edge CSR: stable sort by destination
        order = np.argsort(self.dst, kind="stable")
        self.in_eid = order
        self.indices = self.src[order]
        self.indptr = np.zeros(self.n + 1, dtype=np.int64)
        np.add.at(self.indptr, self.dst + 1, 1)
        self.indptr = np.cumsum(self.indptr)
        # out-edge CSR: stable sort by source
        torder = np.argsort(self.src, kind="stable")
        self.out_eid = torder
        self.t_indices = self.dst[torder]
        self.t_indptr = np.zeros(self.n + 1, dtype=np.int64)
        np.add.at(self.t_indptr, self.src + 1, 1)
        self.t_indptr = np.cumsum(self.t_indptr)

    def number_of_edges(self):
        return int(self.src.shape[0])

    def number_of_nodes(self):
        return self.n

    def in_degrees(self):
        return np.diff(self.indptr)

    def out_degrees(self):
        return np.diff(self.t_indptr)


def from_networkx_ref(nx_graph):
    """dgl.from_networkx restated (data_loader.py:72).  Edge attributes ignored."""
    import networkx as nx

    g = nx.convert_node_labels_to_integers(nx_graph, ordering="sorted")
    if not g.is_directed():
        g = g.to_directed()
    src, dst = [], []
    for u, v in g.edges():
        src.append(u)
        dst.append(v)
    return RefGraph(np.array(src, dtype=np.int64), np.array(dst, dtype=np.int64), g.number_of_nodes())


def batch_ref(graphs):
    """dgl.batch restated (data_loader.py:168)."""
    off = 0
    srcs, dsts, sizes = [], [], []
    for g in graphs:
        srcs.append(g.src + off)
        dsts.append(g.dst + off)
        sizes.extend(g.batch_num_nodes)
        off += g.n
    return RefGraph(np.concatenate(srcs), np.concatenate(dsts), off, batch_num_nodes=sizes)


def norm_ref(g):
    """ndata['norm'] = in_deg^-0.5 with inf -> 0, shape [N,1] fp32 (data_loader.py:75-78)."""
    deg = g.in_degrees().astype(np.float32)
    with np.errstate(divide="ignore"):
        norm = np.power(deg, np.float32(-0.5))
    norm[np.isinf(norm)] = 0
    return norm[:, None]


# ---------------------------------------------------------------- scatter
def project_nodes_to_img_ref(svs, node_labels):
    """graph_io.py:21-24 — labels with a trailing 0 for background (-1 indexes it)."""
    table = np.append(node_labels, 0)
    return table[svs]


BACKGROUND_NODE_LOGITS = [[1.0, -1.0, -1.0, -1.0]]  # utils/hyperparam_helpers.py:25


def project_logits_to_img_ref(svs, node_logits):
    """scripts/generate_gnn_predictions.py:55-61 — float64 result (list promotes)."""
    table = np.concatenate([node_logits, BACKGROUND_NODE_LOGITS])
    return table[svs]


def uncrop_to_brats_size_ref(crop, voxel_preds):
    """data_processing/image_processing.py:21-25."""
    out = np.zeros((240, 240, 155), dtype=np.int16)
    out[crop] = voxel_preds
    return out


def swap_labels_to_brats_ref(preds):
    """scripts/preprocess_dataset.py:159-169 (LABEL_MAP :15) — 3->4, 1->2, 2->1; int16 out;
    RuntimeError('unexpected label') for anything outside {0,1,2,3}."""
    preds = np.asarray(preds)
    if np.setdiff1d(np.unique(preds), [0, 1, 2, 3]).size:
        raise RuntimeError("unexpected label")
    out = np.zeros_like(preds, dtype=np.int16)
    out[preds == 3] = 4
    out[preds == 1] = 2
    out[preds == 2] = 1
    return out


def label_confusion_ref(pred, truth):
    """int64 [5, 5] coincidence table of two integer label arrays: entry [cp, ct] counts the
    positions with class(pred) == cp and class(truth) == ct, class(v) = v for 0..3, else 4.
    (The integers behind /root/reference/model/evaluation.py:24-46,64-79,98-106: every mask
    there is a union of these classes.)"""
    p = np.asarray(pred).ravel().astype(np.int64)
    t = np.asarray(truth).ravel().astype(np.int64)
    cp = np.where((p >= 0) & (p <= 3), p, 4)
    ct = np.where((t >= 0) & (t <= 3), t, 4)
    return np.bincount(5 * cp + ct, minlength=25).reshape(5, 5).astype(np.int64)
