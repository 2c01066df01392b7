"""BraTS <-> internal label maps (reference scripts/preprocess_dataset.py:15,146-169).

BraTS files use {0, 1 NCR/NET, 2 edema, 4 enhancing}; the pipeline trains on
{0, 1 edema, 2 NET, 3 ET}."""
import numpy as np

LABEL_MAP = {4: 3, 2: 1, 1: 2}          # BraTS -> internal
INTERNAL_TO_BRATS = np.array([0, 2, 1, 4], dtype=np.int16)  # index = internal label


def _check(values, allowed):
    if np.setdiff1d(np.unique(values), allowed).size:
        raise RuntimeError("unexpected label")


def swap_labels_from_brats(label_data):
    _check(label_data, [0, 1, 2, 4])
    out = np.zeros_like(label_data, dtype=np.int16)
    for brats, internal in LABEL_MAP.items():
        out[label_data == brats] = internal
    return out


def swap_labels_to_brats(label_data):
    _check(label_data, [0, 1, 2, 3])
    return INTERNAL_TO_BRATS[np.asarray(label_data, dtype=np.int64)]
