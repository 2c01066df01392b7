"""Node- and voxel-level BraTS metrics (Dice for WT/CT/ET, 95th-percentile Hausdorff).

Host-side numpy/scipy, counterpart of the functions of /root/reference/model/evaluation.py
that GNN.evaluate reaches (:24-46, :64-106, :112-189).  Labels: 0 healthy, 1 edema,
2 non-enhancing tumour, 3 enhancing tumour.
"""
import numpy as np
from scipy.ndimage import binary_erosion, distance_transform_edt, generate_binary_structure

HEALTHY, EDEMA, NET, ET = 0, 1, 2, 3


def count_node_labels(preds_or_labels):
    counts = np.zeros(4)
    values, n = np.unique(preds_or_labels, return_counts=True)
    counts[values] = n
    return counts


def _regions(x):
    """Binary masks of whole tumour, tumour core and enhancing tumour."""
    x = np.asarray(x)
    return (x != HEALTHY).astype(int), np.isin(x, [NET, ET]).astype(int), (x == ET).astype(int)


def calculate_dice_from_logical_array(binary_predictions, binary_ground_truth):
    p, g = binary_predictions == 1, binary_ground_truth == 1
    tp = np.count_nonzero(p & g)
    fp = np.count_nonzero(p & (binary_ground_truth == 0))
    fn = np.count_nonzero((binary_predictions == 0) & g)
    if tp + fp + fn == 0:
        return 1
    return (2 * tp) / (2 * tp + fp + fn)


def _region_members(n_classes=5):
    """Class-index sets of WT / CT / ET over the 5 classes of a gts.ops.label_confusion table
    (0..3 = the labels, 4 = any other value: tumour for WT's `!= 0`, outside CT and ET)."""
    return ([c for c in range(n_classes) if c != HEALTHY], [NET, ET], [ET])


def dices_from_confusion(confusion):
    """[WT, CT, ET] Dice from a [5, 5] coincidence table (rows: predicted class, columns: true
    class).  Same integers tp / fp / fn as calculate_dice_from_logical_array counts on the
    binarised arrays, hence the same quotient."""
    m = np.asarray(confusion, dtype=np.int64)
    dices = []
    for members in _region_members(m.shape[0]):
        inside = np.zeros(m.shape[0], dtype=bool)
        inside[members] = True
        tp = int(m[np.ix_(inside, inside)].sum())
        fp = int(m[np.ix_(inside, ~inside)].sum())
        fn = int(m[np.ix_(~inside, inside)].sum())
        dices.append(1 if tp + fp + fn == 0 else (2 * tp) / (2 * tp + fp + fn))
    return dices


def label_counts_from_confusion(confusion):
    """count_node_labels(preds) ++ count_node_labels(labels) from the coincidence table."""
    m = np.asarray(confusion, dtype=np.int64)
    if m[4].sum() or m[:, 4].sum():
        raise IndexError("labels outside 0..3 (count_node_labels indexes a 4-entry table)")
    return np.concatenate([m[:4, :4].sum(axis=1), m[:4, :4].sum(axis=0)]).astype(np.float64)


def calculate_node_dices(preds, labels):
    return [calculate_dice_from_logical_array(p, g) for p, g in zip(_regions(preds), _regions(labels))]


def _surface_distances(result, reference, connectivity=1):
    result = np.atleast_1d(result.astype(bool))
    reference = np.atleast_1d(reference.astype(bool))
    if not result.any():
        raise RuntimeError("The first supplied array does not contain any binary object.")
    if not reference.any():
        raise RuntimeError("The second supplied array does not contain any binary object.")
    footprint = generate_binary_structure(result.ndim, connectivity)
    result_border = result ^ binary_erosion(result, structure=footprint, iterations=1)
    reference_border = reference ^ binary_erosion(reference, structure=footprint, iterations=1)
    return distance_transform_edt(~reference_border)[result_border]


def hd95(result, reference, connectivity=1):
    """Symmetric 95th-percentile Hausdorff distance between two binary objects (unit spacing)."""
    both = np.hstack((_surface_distances(result, reference, connectivity),
                      _surface_distances(reference, result, connectivity)))
    return np.percentile(both, 95)


def calculate_hd95_from_logical_array(pred, gt):
    """0 when the region is absent from both, 300 when absent from exactly one."""
    try:
        return hd95(pred, gt)
    except RuntimeError:
        return 0 if (1 not in pred and 1 not in gt) else 300


def calculate_hd95s(predicted_voxels, true_voxels):
    """[WT, CT, ET] HD95 for one volume (the host half of calculate_brats_metrics)."""
    return [calculate_hd95_from_logical_array(p, g)
            for p, g in zip(_regions(predicted_voxels), _regions(true_voxels))]


def calculate_brats_metrics(predicted_voxels, true_voxels):
    """[WT, CT, ET Dice, WT, CT, ET HD95] for one volume."""
    pairs = list(zip(_regions(predicted_voxels), _regions(true_voxels)))
    dices = [calculate_dice_from_logical_array(p, g) for p, g in pairs]
    hds = [calculate_hd95_from_logical_array(p, g) for p, g in pairs]
    return dices + hds
