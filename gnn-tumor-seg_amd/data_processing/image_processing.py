"""Crop helpers used around the GNN path (counterpart of the functions of
/root/reference/data_processing/image_processing.py that the GNN scripts call)."""
import numpy as np
from scipy import ndimage

BRATS_SHAPE = (240, 240, 155)


def uncrop_to_brats_size(crop, voxel_preds):
    """Paste cropped predictions back into a zero (healthy) BraTS-sized int16 volume
    (reference image_processing.py:21-25)."""
    full = np.zeros(BRATS_SHAPE, dtype=np.int16)
    full[crop] = voxel_preds
    return full


def _any_along_other_axes(mask):
    return np.ix_(mask.any(axis=(1, 2)), mask.any(axis=(0, 2)), mask.any(axis=(0, 1)))


def determine_tumor_crop(preds):
    """Tightest box around the (dilated) predicted tumour; whole image when nothing is predicted
    (reference image_processing.py:8-17)."""
    mask = ndimage.binary_dilation(preds != 0)
    if not mask.any():
        print("No GNN predicted tumor, not cropping image")
        mask = ~mask
    return _any_along_other_axes(mask)


def determine_brain_crop(multi_modal_data):
    """Box of all planes that are not entirely black (reference image_processing.py:31-42)."""
    if multi_modal_data.ndim == 4:
        intensity = np.amax(multi_modal_data, axis=3)
    elif multi_modal_data.ndim == 3:
        intensity = multi_modal_data
    else:
        raise Exception(f"Expected input shape of either nxmxr or nxmxrxC. Instead got {multi_modal_data.shape}")
    return _any_along_other_axes(intensity > 0.01)
