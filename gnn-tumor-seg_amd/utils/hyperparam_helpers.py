"""Hyper-parameter sets for GNN / CNN training.

Counterpart of /root/reference/utils/hyperparam_helpers.py:12-72: the same two namedtuple
types (field names and order), the same defaults and the same random-search draws in the
same RNG call order, so a run seeded identically selects identical settings.
"""
from collections import namedtuple
from time import time

from numpy import random

FullParamSet = namedtuple(
    "FullParamSet",
    "n_epochs in_feats out_classes lr lr_decay w_decay class_weights layer_sizes "
    "feature_dropout gat_heads gat_residuals")
EvalParamSet = namedtuple("EvalParamSet", "in_feats out_classes layer_sizes gat_heads gat_residuals")

DEFAULT_N_CLASSES = 4
DEFAULT_LR = 0.0001
DEFAULT_LR_DECAY = 0.98
DEFAULT_WEIGHT_DECAY = 0.0001
DEFAULT_FEATURE_DROPOUT = 0

DEFAULT_GNN_IN_FEATS = 20
DEFAULT_CNN_IN_FEATS = 8

# logits assigned to voxels outside every supervoxel (id -1): confidently healthy
DEFAULT_BACKGROUND_NODE_LOGITS = [[1.0, -1.0, -1.0, -1.0]]

# (n_epochs, in_feats, class_weights, layer_sizes) per family
_HARDCODED = {
    "CNN": (1, DEFAULT_CNN_IN_FEATS, [0.1, 5, 15, 15], [16]),
    "GNN": (10, DEFAULT_GNN_IN_FEATS, [0.1, 1, 2, 2], [256] * 4),
}
_HARDCODED_GAT_HEADS = [4, 4, 3, 3, 4, 4]
_HARDCODED_GAT_RESIDUALS = [False, False, True, False, False, True]


def populate_hardcoded_hyperparameters(model_type):
    print("Using hardcoded hyperparameters")
    n_epochs, in_feats, class_weights, layer_sizes = _HARDCODED["CNN" if model_type == "CNN" else "GNN"]
    return FullParamSet(n_epochs, in_feats, DEFAULT_N_CLASSES, DEFAULT_LR, DEFAULT_LR_DECAY,
                        DEFAULT_WEIGHT_DECAY, list(class_weights), list(layer_sizes),
                        DEFAULT_FEATURE_DROPOUT, list(_HARDCODED_GAT_HEADS),
                        list(_HARDCODED_GAT_RESIDUALS))


def generate_random_hyperparameters(model_type, seed=None):
    """Random search point.  `seed=None` reproduces the reference: the last three characters
    of str(time()).  Draw order: lr, w_decay, class weights x3, depth, width, heads, residuals."""
    print("Generated Random Hyperparameters")
    rng = random.RandomState(int(str(time())[-3:]) if seed is None else seed)
    lr = rng.choice([0.0001, 0.0005, 0.001])
    w_decay = rng.choice([0.0001, 0])
    feature_dropout = 0.0
    n_epochs = 3
    if model_type == "CNN":
        in_feats = DEFAULT_CNN_IN_FEATS
        class_weights = [0.1, rng.normal(5, 1), rng.normal(10, 2), rng.normal(10, 2)]
        layer_sizes = [16]
    else:
        in_feats = DEFAULT_GNN_IN_FEATS
        class_weights = [0.1, rng.normal(1, 0.2), rng.normal(2, 0.2), rng.normal(2, 0.2)]
        depth = rng.choice([3, 4, 5])
        width = int(rng.choice([64, 128, 256]))
        layer_sizes = depth * [width]
    heads = rng.randint(4, size=len(layer_sizes)) + 3
    residuals = [bool(flag == 1) for flag in rng.binomial(1, p=0.3, size=len(layer_sizes))]
    return FullParamSet(n_epochs, in_feats, DEFAULT_N_CLASSES, lr, DEFAULT_LR_DECAY, w_decay,
                        class_weights, layer_sizes, feature_dropout, heads, residuals)
