"""Input assembly of the refinement CNN.

Counterpart of `combine_logits_and_image` in /root/reference/model/cnn_model.py:85-88, plus the
fused form the joint predictor uses on the GPU.  Training the CNN itself (RefinementModel,
reference :23-82) is dense Conv3d work on the vendor library and is not part of this build.
"""
import torch

from gts import ops


def combine_logits_and_image(gnn_out, img, tumor_crop):
    """[1, C_img + C_logits, cx, cy, cz] from channels-last voxel logits [X,Y,Z,Cl], image
    [X,Y,Z,Ci] and an np.ix_ crop — the reference's signature, for callers that already hold the
    voxel-logit volume (e.g. read back from disk)."""
    combined = torch.cat([img, gnn_out], dim=-1)[tumor_crop]
    return combined.movedim(-1, 0).unsqueeze(0)


def combine_node_logits_and_image(node_logits, background_logits, supervoxel_partitioning, img, box):
    """Same tensor straight from the NODE logits, in one K16 pass on the GPU:
    cat([img, cat(node_logits, background)[svs]], -1)[box] moved to channels-first; neither the
    voxel-logit volume nor the concatenation is materialised.  `box` is a gts.ops.CropBox."""
    bg = torch.as_tensor(background_logits, dtype=torch.float32, device=node_logits.device).reshape(-1)
    return ops.crop_concat(img, supervoxel_partitioning, node_logits.float(), bg, box)
