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


def tumor_crop_from_plane_flags(any_x, any_y, any_z):
    """determine_tumor_crop from the three per-axis `any` vectors of the UNdilated tumour mask
    (what K12's occupancy variant reports).  The 6-neighbourhood dilation of the reference adds,
    to the planes that hold tumour, exactly their two neighbours along the same axis, so the
    crop is the 1-D dilation of each vector; no tumour at all selects the whole volume."""
    flags = [np.asarray(v).astype(bool) for v in (any_x, any_y, any_z)]
    if not any(f.any() for f in flags):
        print("No GNN predicted tumor, not cropping image")
        return np.ix_(*[np.ones(len(f), dtype=bool) for f in flags])
    grown = []
    for f in flags:
        g = f.copy()
        g[1:] |= f[:-1]
        g[:-1] |= f[1:]
        grown.append(g)
    return np.ix_(*grown)


def determine_brain_crop(multi_modal_data):
    """Box of all planes that are not entirely black (reference image_processing.py:31-42)."""
    if multi_modal_data.ndim == 4:
        intensity = np.amax(multi_modal_data, axis=3)
    elif multi_modal_data.ndim == 3:
        intensity = multi_modal_data
    else:
        raise Exception(f"Expected input shape of either nxmxr or nxmxrxC. Instead got {multi_modal_data.shape}")
    return _any_along_other_axes(intensity > 0.01)
