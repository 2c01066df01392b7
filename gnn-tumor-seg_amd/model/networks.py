"""GraphSAGE / GAT node classifiers on the MI355X-native layers of `gts.nn`.

Counterpart of /root/reference/model/networks.py:20-81 with the same public surface:
`init_graph_net(model_type, hp) -> nn.Module`, `net(graph, features) -> logits [N, n_classes]`,
modules registered under `layers.{i}` so checkpoints interchange.
`graph` is a `gts.Graph` (what data_processing.data_loader hands out) instead of a DGLGraph.
"""
import torch.nn as nn
import torch.nn.functional as F

from gts.nn import ActLink, GATConv, SAGEConv, sage_pool_stack

_SAGE_AGGREGATORS = {"GSpool": "pool", "GSgcn": "gcn", "GSmean": "mean"}


class GraphSage(nn.Module):
    """len(layer_sizes)+1 SAGEConv layers; ReLU and `dropout` on all but the last
    (reference: model/networks.py:21-36)."""

    def __init__(self, in_feats, layer_sizes, n_classes, aggregator_type, dropout):
        super().__init__()
        widths = [in_feats, *layer_sizes]
        hidden = [SAGEConv(w_in, w_out, aggregator_type, feat_drop=dropout, activation=F.relu)
                  for w_in, w_out in zip(widths[:-1], widths[1:])]
        head = SAGEConv(widths[-1], n_classes, aggregator_type, feat_drop=0, activation=None)
        self.layers = nn.ModuleList([*hidden, head])
        self.fuse_layers = True

    def forward(self, graph, features):
        # pool stacks run as one fused autograd node (identical arithmetic, see gts.nn)
        fused = sage_pool_stack(graph, features, list(self.layers)) if self.fuse_layers else None
        if fused is not None:
            return fused
        h = features
        for conv in self.layers:
            h = conv(graph, h)
        return h


class GAT(nn.Module):
    """Multi-head GAT stack; hidden outputs are flattened over heads, the single-head output
    layer is averaged over its head axis (reference: model/networks.py:39-66).
    As in the reference, hidden layer i takes `residuals[i]` and the output layer reads
    `heads[-1]`, so len(heads) must equal len(layer_sizes) for the widths to line up."""

    def __init__(self, in_feats, layer_sizes, n_classes, heads, residuals,
                 activation=F.elu, feat_drop=0, attn_drop=0, negative_slope=0.2):
        super().__init__()
        self.activation = activation
        common = (feat_drop, attn_drop, negative_slope)
        convs = [GATConv(in_feats, layer_sizes[0], heads[0], *common, False, activation)]
        for i in range(1, len(layer_sizes)):
            convs.append(GATConv(layer_sizes[i - 1] * heads[i - 1], layer_sizes[i], heads[i],
                                 *common, residuals[i], activation))
        convs.append(GATConv(layer_sizes[-1] * heads[-1], n_classes, 1, *common, False, None))
        self.layers = nn.ModuleList(convs)

    def forward(self, g, inputs):
        # every hidden output feeds the next layer and nothing else: its ELU backward and bias gradient ride in the next
        # layer's input-gradient GEMM (gts.nn.ActLink; same values as layer-by-layer)
        links = [ActLink() for _ in self.layers[:-1]]
        h = inputs
        for i, conv in enumerate(self.layers[:-1]):
            h = conv(g, h, below=links[i - 1] if i else None, above=links[i]).flatten(1)
        return self.layers[-1](g, h, below=links[-1]).mean(1)


class CnnRefinementNet(nn.Module):
    """Two 5x5x5 replicate-padded Conv3d layers with a ReLU in between, refining
    [1, in_feats, x, y, z] (image modalities ++ GNN voxel logits) into class scores
    (reference: model/networks.py:84-94; checkpoint keys `conv_layers.{0,1}.{weight,bias}`).
    Dense convolution is vendor-library territory: it runs on MIOpen through torch."""

    def __init__(self, in_feats, out_classes, layer_sizes):
        super().__init__()
        widths = [in_feats, layer_sizes[0], out_classes]
        self.conv_layers = nn.ModuleList(
            nn.Conv3d(in_channels=a, out_channels=b, kernel_size=5, stride=1, padding=2, padding_mode="replicate")
            for a, b in zip(widths[:-1], widths[1:]))

    def forward(self, comb_img_logits):
        return self.conv_layers[1](F.relu(self.conv_layers[0](comb_img_logits)))


def init_graph_net(model_type, hp):
    """'GSpool' | 'GSgcn' | 'GSmean' | 'GAT' -> module (reference: model/networks.py:68-81).
    `hp` is a FullParamSet / EvalParamSet; feature dropout only exists on the former and is
    only wired into GraphSAGE, exactly as in the reference."""
    dropout = hp.feature_dropout if "feature_dropout" in hp._fields else 0
    if model_type in _SAGE_AGGREGATORS:
        return GraphSage(in_feats=hp.in_feats, layer_sizes=hp.layer_sizes, n_classes=hp.out_classes,
                         aggregator_type=_SAGE_AGGREGATORS[model_type], dropout=dropout)
    if model_type == "GAT":
        return GAT(in_feats=hp.in_feats, layer_sizes=hp.layer_sizes, n_classes=hp.out_classes,
                   heads=hp.gat_heads, residuals=hp.gat_residuals)
    raise Exception(f"Unknown model type: {model_type}")
