"""Chain a trained GNN and refinement CNN over a preprocessed dataset and save the predicted
segmentations.

Command line and output files follow /root/reference/scripts/generate_joint_predictions.py:
27-110.  Per sample everything between the graph forward pass and the finished label volume
stays on the GPU:

    node logits --K12 (arg-max + projection + tumour-plane flags)--> crop box (3 small vectors
    to the host, 1-D dilation) --K16 (crop + image/logit concat, channels-first)--> Conv3d x2
    (MIOpen) --K17 (channel arg-max scattered into the zero volume, BraTS relabel)--> host.

The reference materialises the [X,Y,Z,4] voxel logits, copies them to the host for an argmax
and a 3-D dilation, and builds the [X,Y,Z,8] concatenation before cropping.
"""
import argparse
import os

import numpy as np
import torch

import Filepaths
from data_processing import data_loader, nifti_io
from data_processing.image_processing import tumor_crop_from_plane_flags, uncrop_to_brats_size
from data_processing.labels import INTERNAL_TO_BRATS
from gts import ops
from model.cnn_model import combine_node_logits_and_image
from model.networks import CnnRefinementNet, init_graph_net
from utils.hyperparam_helpers import DEFAULT_BACKGROUND_NODE_LOGITS, EvalParamSet

output_dir = None


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("generate_joint_predictions needs an AMD GPU (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def load_nets(gnn_type, gnn_weights, cnn_weights,
              gnn_hp=EvalParamSet(in_feats=20, out_classes=4, layer_sizes=[256] * 4, gat_heads=None,
                                  gat_residuals=None),
              cnn_hp=EvalParamSet(in_feats=8, out_classes=4, layer_sizes=[16], gat_heads=None,
                                  gat_residuals=None)):
    """Architectures must correspond to the weight files (the reference hard-codes these two)."""
    device = _device()
    graph_net = init_graph_net(gnn_type, gnn_hp).to(device)
    conv_net = CnnRefinementNet(cnn_hp.in_feats, cnn_hp.out_classes, cnn_hp.layer_sizes).to(device)
    graph_net.load_state_dict(torch.load(gnn_weights, map_location=device, weights_only=True))
    conv_net.load_state_dict(torch.load(cnn_weights, map_location=device, weights_only=True))
    graph_net.eval()
    conv_net.eval()
    return graph_net, conv_net


def predict_one_sample(graph_net, conv_net, graph, node_feats, img, supervoxel_partitioning, relabel=None):
    """int16 label volume of the partitioning's shape: GNN prediction refined by the CNN inside
    the (dilated) box around the GNN-predicted tumour, healthy outside (reference :59-73).
    `relabel` (optional int16 table on the device) maps the labels on the way out."""
    device = _device()
    with torch.no_grad():
        graph = graph.to(device)
        node_feats = torch.as_tensor(np.asarray(node_feats), dtype=torch.float32).to(device)
        img = torch.as_tensor(np.asarray(img), dtype=torch.float32).to(device)
        svs = torch.as_tensor(np.asarray(supervoxel_partitioning)).to(device)
        node_logits = graph_net(graph, node_feats).float()
        _, plane_flags = ops.project_argmax_occupancy(svs, node_logits)                     # K12
        crop = tumor_crop_from_plane_flags(*[f.cpu().numpy() for f in plane_flags])
        box = ops.CropBox(*[c.reshape(-1) for c in crop], svs.shape, device)
        cnn_in = combine_node_logits_and_image(node_logits, DEFAULT_BACKGROUND_NODE_LOGITS, svs, img, box)  # K16
        refined_voxel_logits = conv_net(cnn_in)
        return ops.argmax_scatter(refined_voxel_logits.float(), box, relabel).cpu().numpy()  # K17


def save_predictions(graph_net, conv_net, dataset):
    relabel = torch.from_numpy(INTERNAL_TO_BRATS).to(_device())
    for mri, graph, node_feats, img in dataset:
        try:
            supervoxel_partitioning = dataset.get_supervoxel_partitioning(mri)
            raw_data_crop = dataset.get_crop(mri)
        except FileNotFoundError as e:
            raise FileNotFoundError(f"Couldnt predict {mri} because couldn't read in a required file: {e}")
        pred = predict_one_sample(graph_net, conv_net, graph, node_feats, img, supervoxel_partitioning, relabel)
        nifti_io.save_as_nifti(uncrop_to_brats_size(raw_data_crop, pred), f"{output_dir}{os.sep}{mri}.nii.gz")


_ARGUMENTS = [
    ("-d", "--data_dir", Filepaths.PROCESSED_DATA_DIR, "path to the directory where data is stored"),
    ("-p", "--data_prefix", "", "A prefix that all data folders share, i.e. BraTS2021."),
    ("-o", "--output_dir", None, "Directory to save predictions to"),
    ("-m", "--gnn_type", "GSpool", "What graph learning layer the saved model uses. GSpool, GSmean, GSgcn, GAT"),
    ("-c", "--cnn_weights", "", "Path to weights file for convolutional net"),
    ("-g", "--gnn_weights", "", "Path to weights file for graph net"),
]


def main(argv=None):
    global output_dir
    parser = argparse.ArgumentParser()
    for short, long_, default, text in _ARGUMENTS:
        parser.add_argument(short, long_, default=default, help=text, type=str)
    args = parser.parse_args(argv)
    # the reference falls back on an attribute its parser never defines (:97); predictions go to PRED_DIR
    output_dir = os.path.expanduser(args.output_dir if args.output_dir else Filepaths.PRED_DIR)
    if not os.path.isdir(output_dir):
        print(f"Creating save directory: {output_dir}")
        os.makedirs(output_dir)
    dataset = data_loader.ImageGraphDataset(os.path.expanduser(args.data_dir), args.data_prefix,
                                            read_image=True, read_graph=True, read_label=False)
    graph_net, conv_net = load_nets(args.gnn_type, os.path.expanduser(args.gnn_weights),
                                    os.path.expanduser(args.cnn_weights))
    save_predictions(graph_net, conv_net, dataset)
    print(f"Finished saving predictions generated by {args.gnn_weights} and {args.cnn_weights} "
          f"in folder {output_dir}")


if __name__ == "__main__":
    main()
