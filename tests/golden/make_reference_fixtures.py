"""Generate golden fixtures by IMPORTING the reference's pure-numpy modules.

Run in the build container only (needs /root/reference):   python tests/golden/make_reference_fixtures.py
Writes data files (inputs + expected outputs) next to this script; no reference source is
copied.  Importable reference modules: data_processing.graph_io,
utils.hyperparam_helpers, utils.training_helpers, data_processing.image_processing,
model.evaluation (the last one needs the
`np.bool` alias that NumPy 2 removed: added here as an environment shim, the reference file
is untouched).  Everything that imports dgl / nibabel is NOT importable and is not used.
"""
import io
import json
import os
import sys
from contextlib import redirect_stdout

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
if not hasattr(np, "bool"):
    np.bool = bool  # environment shim for model/evaluation.py:170 under NumPy 2

from data_processing import graph_io as ref_graph_io          # noqa: E402
from data_processing import image_processing as ref_img       # noqa: E402
from model import evaluation as ref_eval                      # noqa: E402
from utils import hyperparam_helpers as ref_hp                # noqa: E402
from utils import training_helpers as ref_th                  # noqa: E402


def scatter_fixture():
    rng = np.random.default_rng(20240607)
    n_nodes = 37
    svs = rng.integers(-1, n_nodes, size=(12, 10, 8)).astype(np.int16)
    svs[0, :, :] = -1
    labels = rng.integers(0, 4, size=n_nodes).astype(np.int64)
    logits = rng.standard_normal((n_nodes, 4)).astype(np.float32)
    out_labels = ref_graph_io.project_nodes_to_img(svs, labels)
    # scripts/generate_gnn_predictions.py:60-61, executed with the reference's own constant
    table = np.concatenate([logits, ref_hp.DEFAULT_BACKGROUND_NODE_LOGITS])
    out_logits = table[svs]
    np.savez_compressed(os.path.join(OUT, "ref_project.npz"), svs=svs, labels=labels, logits=logits,
                        out_labels=out_labels, out_logits=out_logits,
                        background=np.array(ref_hp.DEFAULT_BACKGROUND_NODE_LOGITS))


def _plain(x):
    if isinstance(x, np.ndarray):
        return x.tolist()
    if isinstance(x, (np.floating, np.integer, np.bool_)):
        return x.item()
    if isinstance(x, (list, tuple)):
        return [_plain(v) for v in x]
    return x


def hyperparam_fixture():
    cases = []
    with redirect_stdout(io.StringIO()):
        for model_type in ("GSpool", "GAT", "CNN"):
            hp = ref_hp.populate_hardcoded_hyperparameters(model_type)
            cases.append({"kind": "hardcoded", "model_type": model_type, "fields": list(hp._fields),
                          "values": _plain(list(hp))})
        for fake_time in (1700000000.123, 1700000001.907, 1712345678.555, 1699999999.042):
            for model_type in ("GSpool", "CNN"):
                ref_hp.time = lambda t=fake_time: t
                hp = ref_hp.generate_random_hyperparameters(model_type)
                cases.append({"kind": "random", "model_type": model_type,
                              "seed": int(str(fake_time)[-3:]), "fields": list(hp._fields),
                              "values": _plain(list(hp))})
    consts = {k: _plain(getattr(ref_hp, k)) for k in dir(ref_hp) if k.startswith("DEFAULT_")}
    consts["EvalParamSet_fields"] = list(ref_hp.EvalParamSet._fields)
    with open(os.path.join(OUT, "ref_hyperparams.json"), "w") as f:
        json.dump({"cases": cases, "constants": consts}, f, indent=1)


class _FakeModel:
    def __init__(self, losses):
        self.losses, self.saved, self.i = list(losses), [], 0

    def run_epoch(self):
        self.i += 1
        return self.losses[self.i - 1]

    def save_weights(self, folder, name):
        self.saved.append([self.i, folder, name])


def training_fixture():
    out = {}
    tmp = os.path.join(OUT, "_tmp_progress.txt")
    with redirect_stdout(io.StringIO()):
        for model_type in ("GSpool", "GAT"):
            hp = ref_hp.populate_hardcoded_hyperparameters(model_type)
            ref_th.create_run_progress_file(tmp, model_type, hp)
            ref_th.update_progress_file(tmp, "run_f1_train", 0.4321, [0.9, 0.8, 0.7])
            ref_th.update_progress_file(tmp, "run_f1_val", np.float64(0.5), np.array([0.5, 0.25, 0.125]))
            out[f"progress_{model_type}"] = open(tmp).read()
    os.remove(tmp)
    out["folds"] = [{"n": n, "k": k, "folds": ref_th.chunk_dataset_into_folds(range(n), k)}
                    for n, k in ((10, 5), (11, 3), (7, 2), (5, 5), (3, 1))]
    traces = []
    for losses in ([1.0, 0.9, 0.8, 0.7, 0.6, 0.5], [1.0, 0.5, 0.6, 0.7, 0.4, 0.9, 0.95, 0.2],
                   [0.5, 0.5005, 0.5009, 0.5011, 0.3], [2000.0, 1500.0, 1200.0, 1100.0],
                   [0.3, 0.2, 0.1, 0.1, 0.1, 0.1005, 0.2]):
        m = _FakeModel(losses)
        buf = io.StringIO()
        with redirect_stdout(buf):
            ref_th.train_on_fold(m, "ckpt/", len(losses), "runA", 3)
        traces.append({"losses": losses, "epochs_run": m.i, "saved": m.saved, "stdout": buf.getvalue()})
    out["train_on_fold"] = traces
    with open(os.path.join(OUT, "ref_training.json"), "w") as f:
        json.dump(out, f, indent=1)


def evaluation_fixture():
    rng = np.random.default_rng(77)
    cases = {}
    for i, shape in enumerate([(14, 12, 10), (9, 9, 9), (6, 7, 8)]):
        pred = rng.choice(4, size=shape, p=[0.6, 0.2, 0.1, 0.1])
        true = rng.choice(4, size=shape, p=[0.6, 0.2, 0.1, 0.1])
        if i == 1:
            pred[pred == 3] = 0          # ET absent from the prediction only -> HD95 300
        if i == 2:
            pred[pred == 3] = 0
            true[true == 3] = 0          # ET absent from both -> Dice 1, HD95 0
        cases[f"pred{i}"], cases[f"true{i}"] = pred, true
        cases[f"brats{i}"] = np.array(ref_eval.calculate_brats_metrics(pred, true), dtype=np.float64)
        cases[f"node_dice{i}"] = np.array(ref_eval.calculate_node_dices(pred.ravel(), true.ravel()),
                                          dtype=np.float64)
        cases[f"counts{i}"] = ref_eval.count_node_labels(pred.ravel())
    np.savez_compressed(os.path.join(OUT, "ref_evaluation.npz"), **cases)


def tumor_crop_fixture():
    """determine_tumor_crop on label volumes: one blob, tumour on the border, two separated
    blobs (gaps in the selected planes), isolated voxels, nothing predicted."""
    rng = np.random.default_rng(4242)
    shape = (18, 15, 13)
    volumes = []
    v = np.zeros(shape, dtype=np.int64)
    v[5:9, 4:7, 6:8] = 2
    volumes.append(v)
    v = np.zeros(shape, dtype=np.int64)
    v[0:2, 12:15, 0] = 1
    v[17, 0, 12] = 3
    volumes.append(v)
    v = np.zeros(shape, dtype=np.int64)
    v[2:4, 2:4, 2:4] = 1
    v[10:13, 9:11, 8:11] = 3
    volumes.append(v)
    v = (rng.random(shape) < 0.004).astype(np.int64) * rng.integers(1, 4, shape)
    volumes.append(v)
    volumes.append(np.zeros(shape, dtype=np.int64))
    cases = {}
    with redirect_stdout(io.StringIO()):
        for i, vol in enumerate(volumes):
            ix = ref_img.determine_tumor_crop(vol)
            cases[f"preds{i}"] = vol.astype(np.int16)
            for name, idx in zip("xyz", ix):
                cases[f"{name}{i}"] = np.asarray(idx).reshape(-1).astype(np.int64)
    np.savez_compressed(os.path.join(OUT, "ref_tumor_crop.npz"), **cases)


if __name__ == "__main__":
    if sys.argv[1:] == ["tumor_crop"]:      # add this one fixture without touching the others
        tumor_crop_fixture()
        sys.exit(0)
    scatter_fixture()
    hyperparam_fixture()
    training_fixture()
    evaluation_fixture()
    tumor_crop_fixture()
    print("fixtures written to", OUT)
