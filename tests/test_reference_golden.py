"""Oracle AND product host code against fixtures produced by the reference's own functions
(tests/golden/make_reference_fixtures.py).  CPU only."""
import io
import json
import os
from contextlib import redirect_stdout

import numpy as np
import pytest

from oracle import graph_ref


@pytest.fixture(scope="module")
def project_fx(golden_dir):
    return np.load(os.path.join(golden_dir, "ref_project.npz"))


def test_oracle_project_labels_matches_reference(project_fx):
    out = graph_ref.project_nodes_to_img_ref(project_fx["svs"], project_fx["labels"])
    assert out.dtype == project_fx["out_labels"].dtype
    assert np.array_equal(out, project_fx["out_labels"])


def test_oracle_project_logits_matches_reference(project_fx):
    out = graph_ref.project_logits_to_img_ref(project_fx["svs"], project_fx["logits"])
    assert out.dtype == project_fx["out_logits"].dtype == np.float64
    assert np.array_equal(out, project_fx["out_logits"])
    assert np.array_equal(np.array(graph_ref.BACKGROUND_NODE_LOGITS), project_fx["background"])


def _plain(x):
    if isinstance(x, np.ndarray):
        return x.tolist()
    if isinstance(x, (np.floating, np.integer, np.bool_)):
        return x.item()
    if isinstance(x, (list, tuple)):
        return [_plain(v) for v in x]
    return x


def test_hyperparameters_match_reference(golden_dir):
    from utils import hyperparam_helpers as hp_mod

    fx = json.load(open(os.path.join(golden_dir, "ref_hyperparams.json")))
    for name, value in fx["constants"].items():
        if name == "EvalParamSet_fields":
            assert list(hp_mod.EvalParamSet._fields) == value
        else:
            assert _plain(getattr(hp_mod, name)) == value, name
    with redirect_stdout(io.StringIO()):
        for case in fx["cases"]:
            if case["kind"] == "hardcoded":
                hp = hp_mod.populate_hardcoded_hyperparameters(case["model_type"])
            else:
                hp = hp_mod.generate_random_hyperparameters(case["model_type"], seed=case["seed"])
            assert list(hp._fields) == case["fields"]
            assert _plain(list(hp)) == case["values"], case


def test_random_hyperparameters_default_seed_rule(monkeypatch):
    """seed=None takes the last three characters of str(time()), like the reference."""
    from utils import hyperparam_helpers as hp_mod

    monkeypatch.setattr(hp_mod, "time", lambda: 1700000001.907)
    with redirect_stdout(io.StringIO()):
        a = hp_mod.generate_random_hyperparameters("GSpool")
        b = hp_mod.generate_random_hyperparameters("GSpool", seed=907)
    assert _plain(list(a)) == _plain(list(b))


class _FakeModel:
    def __init__(self, losses):
        self.losses, self.saved, self.i = list(losses), [], 0

    def run_epoch(self):
        self.i += 1
        return self.losses[self.i - 1]

    def save_weights(self, folder, name):
        self.saved.append([self.i, folder, name])


def test_training_helpers_match_reference(golden_dir, tmp_path):
    from utils import hyperparam_helpers as hp_mod
    from utils import training_helpers as th

    fx = json.load(open(os.path.join(golden_dir, "ref_training.json")))
    with redirect_stdout(io.StringIO()):
        for model_type in ("GSpool", "GAT"):
            fp = tmp_path / f"{model_type}.txt"
            th.create_run_progress_file(fp, model_type, hp_mod.populate_hardcoded_hyperparameters(model_type))
            th.update_progress_file(fp, "run_f1_train", 0.4321, [0.9, 0.8, 0.7])
            th.update_progress_file(fp, "run_f1_val", np.float64(0.5), np.array([0.5, 0.25, 0.125]))
            assert fp.read_text() == fx[f"progress_{model_type}"]
    for case in fx["folds"]:
        assert [list(f) for f in th.chunk_dataset_into_folds(range(case["n"]), case["k"])] == case["folds"]
    for trace in fx["train_on_fold"]:
        m = _FakeModel(trace["losses"])
        buf = io.StringIO()
        with redirect_stdout(buf):
            th.train_on_fold(m, "ckpt/", len(trace["losses"]), "runA", 3)
        assert m.i == trace["epochs_run"]
        assert m.saved == trace["saved"]
        assert buf.getvalue() == trace["stdout"]


def test_evaluation_matches_reference(golden_dir):
    from model import evaluation

    fx = np.load(os.path.join(golden_dir, "ref_evaluation.npz"))
    for i in range(3):
        pred, true = fx[f"pred{i}"], fx[f"true{i}"]
        got = np.array(evaluation.calculate_brats_metrics(pred, true), dtype=np.float64)
        assert np.array_equal(got, fx[f"brats{i}"]), (i, got, fx[f"brats{i}"])
        nd = np.array(evaluation.calculate_node_dices(pred.ravel(), true.ravel()), dtype=np.float64)
        assert np.array_equal(nd, fx[f"node_dice{i}"])
        assert np.array_equal(evaluation.count_node_labels(pred.ravel()), fx[f"counts{i}"])


def test_confusion_table_route_matches_reference(golden_dir):
    """The Dice numbers GNN.evaluate derives from 5x5 coincidence tables (counted on the GPU by
    K15, here by the numpy oracle) are the reference's, on the reference's own outputs."""
    from model import evaluation

    fx = np.load(os.path.join(golden_dir, "ref_evaluation.npz"))
    for i in range(3):
        table = graph_ref.label_confusion_ref(fx[f"pred{i}"], fx[f"true{i}"])
        dices = np.array(evaluation.dices_from_confusion(table), dtype=np.float64)
        assert np.array_equal(dices, fx[f"brats{i}"][:3]) and np.array_equal(dices, fx[f"node_dice{i}"])
        counts = evaluation.label_counts_from_confusion(table)
        assert np.array_equal(counts[:4], fx[f"counts{i}"])
        assert np.array_equal(counts[4:], evaluation.count_node_labels(fx[f"true{i}"].ravel()))
        hds = np.array(evaluation.calculate_hd95s(fx[f"pred{i}"], fx[f"true{i}"]), dtype=np.float64)
        assert np.array_equal(hds, fx[f"brats{i}"][3:])
    # labels outside 0..3: whole tumour for `!= 0`, outside CT / ET; count_node_labels has no slot
    rng = np.random.default_rng(0)
    pred, true = rng.integers(-2, 7, 5000), rng.integers(-2, 7, 5000)
    table = graph_ref.label_confusion_ref(pred, true)
    assert table.sum() == 5000
    assert evaluation.dices_from_confusion(table) == evaluation.calculate_node_dices(pred, true)
    with pytest.raises(IndexError):
        evaluation.label_counts_from_confusion(table)
    assert evaluation.dices_from_confusion(np.zeros((5, 5), dtype=np.int64)) == [1, 1, 1]


def test_tumor_crop_matches_reference(golden_dir):
    """determine_tumor_crop three ways — the host mirror, the oracle restatement, and the route
    the joint predictor takes (per-axis `any` of the undilated mask, as K12 reports them, then
    1-D dilation) — against the reference's own outputs."""
    from data_processing import image_processing
    from oracle import joint_ref

    fx = np.load(os.path.join(golden_dir, "ref_tumor_crop.npz"))
    for i in range(5):
        preds = fx[f"preds{i}"]
        want = [fx[f"{a}{i}"] for a in "xyz"]
        mask = preds != 0
        flags = (mask.any(axis=(1, 2)), mask.any(axis=(0, 2)), mask.any(axis=(0, 1)))
        with redirect_stdout(io.StringIO()):
            routes = (image_processing.determine_tumor_crop(preds), joint_ref.determine_tumor_crop_ref(preds),
                      image_processing.tumor_crop_from_plane_flags(*flags))
        for ix in routes:
            assert [a.shape for a in ix] == [(len(want[0]), 1, 1), (1, len(want[1]), 1), (1, 1, len(want[2]))]
            assert all(np.array_equal(np.asarray(a).reshape(-1), w) for a, w in zip(ix, want)), i
    assert len(fx["x4"]) == 18 and len(fx["y4"]) == 15 and len(fx["z4"]) == 13      # nothing predicted: whole volume
    assert np.any(np.diff(fx["x2"]) > 1)                                           # separated blobs leave gaps


def test_label_maps():
    from data_processing import labels

    internal = np.array([[0, 1], [2, 3]])
    assert np.array_equal(labels.swap_labels_to_brats(internal), graph_ref.swap_labels_to_brats_ref(internal))
    assert labels.swap_labels_to_brats(internal).dtype == np.int16
    brats = np.array([0, 1, 2, 4, 4, 2])
    assert np.array_equal(labels.swap_labels_to_brats(labels.swap_labels_from_brats(brats)), brats)
    with pytest.raises(RuntimeError):
        labels.swap_labels_to_brats(np.array([0, 5]))
    with pytest.raises(RuntimeError):
        labels.swap_labels_from_brats(np.array([0, 3]))


def test_uncrop_matches_oracle():
    from data_processing.image_processing import uncrop_to_brats_size

    rng = np.random.default_rng(1)
    preds = rng.integers(0, 4, size=(5, 6, 7))
    mask = np.zeros((240, 240, 155), dtype=bool)
    mask[10:15, 20:26, 30:37] = True
    crop = np.ix_(mask.any(axis=(1, 2)), mask.any(axis=(0, 2)), mask.any(axis=(0, 1)))
    a = uncrop_to_brats_size(crop, preds)
    assert a.dtype == np.int16 and a.shape == (240, 240, 155)
    assert np.array_equal(a, graph_ref.uncrop_to_brats_size_ref(crop, preds))
