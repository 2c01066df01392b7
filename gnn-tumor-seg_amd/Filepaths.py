"""Default directories for the CLI entry points (`import Filepaths` in scripts/train_gnn.py).

Same five names as the reference's Filepaths.py:7-11 (all empty there); here each may also be
supplied through an environment variable of the same name prefixed with GTS_.
"""
import os as _os


def _default(name):
    return _os.environ.get("GTS_" + name, "")


GNN_LOGIT_DIR = _default("GNN_LOGIT_DIR")
INPUT_MRI_DIR = _default("INPUT_MRI_DIR")
PROCESSED_DATA_DIR = _default("PROCESSED_DATA_DIR")
PRED_DIR = _default("PRED_DIR")
LOG_DIR = _default("LOG_DIR")
