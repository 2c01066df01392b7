"""TEST INFRASTRUCTURE ONLY — CPU restatement of the joint GNN -> CNN predictor.

Follows /root/reference/scripts/generate_joint_predictions.py:59-73 (predict_one_sample),
model/cnn_model.py:85-88 (combine_logits_and_image), data_processing/image_processing.py:8-17
(determine_tumor_crop) and model/networks.py:84-94 (CnnRefinementNet) step by step with
numpy / torch-CPU, materialising every intermediate the reference materialises.  Only tests
may import this module; the product path never does.  The integer steps (projection, crop,
scatter) are pinned by the reference-generated fixture tests/golden/ref_project.npz /
ref_tumor_crop.npz; the convolution is torch's own CPU Conv3d ("parity unpinned" beyond that).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from scipy import ndimage


def determine_tumor_crop_ref(preds):
    """image_processing.py:8-17."""
    mask = ndimage.binary_dilation(preds != 0)
    if np.all(~mask):
        mask = ~mask
    return np.ix_(mask.any(axis=(1, 2)), mask.any(axis=(0, 2)), mask.any(axis=(0, 1)))


class RefCnnRefinementNet(nn.Module):
    """networks.py:84-94."""

    def __init__(self, in_feats, out_classes, layer_sizes):
        super().__init__()
        self.conv_layers = nn.ModuleList()
        self.conv_layers.append(nn.Conv3d(in_feats, layer_sizes[0], 5, 1, 2, padding_mode="replicate"))
        self.conv_layers.append(nn.Conv3d(layer_sizes[0], out_classes, 5, 1, 2, padding_mode="replicate"))

    def forward(self, x):
        return self.conv_layers[1](F.relu(self.conv_layers[0](x)))


def combine_logits_and_image_ref(voxel_logits, img, crop):
    """cnn_model.py:85-88 on torch CPU tensors."""
    return torch.cat([img, voxel_logits], dim=-1)[crop].movedim(-1, 0).unsqueeze(0)


def predict_one_sample_ref(node_logits, conv_net, img, svs, background_logits):
    """generate_joint_predictions.py:59-73 from the GNN's node logits on (numpy fp32 [N,C]).
    Returns (label volume int16 like svs, crop, cnn input, refined logits) for inspection."""
    table = torch.cat([torch.from_numpy(node_logits), torch.FloatTensor(background_logits)], dim=0)
    voxel_logits = table[svs.astype(np.int64)]                       # negative ids wrap to the background row
    crop = determine_tumor_crop_ref(voxel_logits.numpy().argmax(axis=-1))
    cnn_in = combine_logits_and_image_ref(voxel_logits, torch.from_numpy(img), crop)
    with torch.no_grad():
        refined = conv_net(cnn_in)
    cropped = torch.argmax(refined.squeeze(0), dim=0).numpy()
    volume = np.zeros_like(svs, dtype=np.int16)
    volume[crop] = cropped
    return volume, crop, cnn_in, refined
