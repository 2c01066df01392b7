"""End to end through the reference's command-line surface on a synthetic dataset: train
(k-fold and full), checkpoints, progress file, then logits / prediction volumes."""
import io
import os
from contextlib import redirect_stdout

import numpy as np
import pytest
import torch

from tests.dataset_util import write_dataset

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _lib(hip_lib):
    assert torch.cuda.is_available()
    return hip_lib


def test_train_and_predict_cli(tmp_path, monkeypatch):
    from data_processing import nifti_io
    from data_processing.data_loader import ImageGraphDataset
    from oracle import graph_ref
    from scripts import generate_gnn_predictions as gen
    from scripts import train_gnn
    from utils import hyperparam_helpers as hp_mod

    data = str(tmp_path / "data") + "/"
    logs = str(tmp_path / "logs")
    os.makedirs(logs)
    write_dataset(data, 6)
    # 2 epochs of the hard-coded configuration are enough here
    real = hp_mod.populate_hardcoded_hyperparameters
    monkeypatch.setattr(train_gnn, "populate_hardcoded_hyperparameters",
                        lambda m: real(m)._replace(n_epochs=2, layer_sizes=[256] * 4))
    buf = io.StringIO()
    with redirect_stdout(buf):
        train_gnn.main(["-d", data, "-o", logs, "-r", "runA", "-m", "GSpool", "-k", "2", "-p", "BraTS_"])
    out = buf.getvalue()
    assert "Fold contains 3 examples" in out and "#runA_f2_val Results#" in out
    progress = open(os.path.join(logs, "runA.txt")).read().splitlines()
    assert progress[0] == "----Model Parameters----" and progress[1] == "Model\tGSpool"
    rows = [line.split("\t") for line in progress if line.startswith("runA_f")]
    assert [r[0] for r in rows] == ["runA_f1_train", "runA_f1_val", "runA_f2_train", "runA_f2_val"]
    assert all(len(r) == 5 and np.isfinite(float(r[1])) for r in rows)
    ckpt = os.path.join(logs, "runA_f1.pt")
    assert os.path.exists(ckpt)

    preds_dir, logits_dir = str(tmp_path / "preds"), str(tmp_path / "logits")
    with redirect_stdout(io.StringIO()):
        gen.main(["-d", data, "-p", "BraTS_", "-o", logits_dir, "-w", ckpt, "-f", "logits"])
        gen.main(["-d", data, "-p", "BraTS_", "-o", preds_dir, "-w", ckpt, "-f", "preds"])
    ds = ImageGraphDataset(data, "BraTS_", read_image=False, read_graph=True, read_label=False)
    net = gen.load_net_and_weights(ckpt).to("cuda")
    for mri_id, graph, feats in ds:
        with torch.no_grad():
            logits = net(graph.to("cuda"), torch.FloatTensor(feats).to("cuda")).cpu().numpy()
        svs = ds.get_supervoxel_partitioning(mri_id)
        vox = nifti_io.read_nifti(os.path.join(logits_dir, f"{mri_id}_logits.nii.gz"), np.float64)
        assert np.array_equal(vox, graph_ref.project_logits_to_img_ref(svs, logits))
        pred = nifti_io.read_nifti(os.path.join(preds_dir, f"{mri_id}.nii.gz"), np.int16)
        want = graph_ref.project_nodes_to_img_ref(svs, logits.argmax(1))
        want = graph_ref.uncrop_to_brats_size_ref(ds.get_crop(mri_id), want)
        assert pred.shape == (240, 240, 155)
        assert np.array_equal(pred, graph_ref.swap_labels_to_brats_ref(want))
    with pytest.raises(ValueError):
        gen.save_predictions(net, ds, "bogus")


def test_evaluate_on_device_counts_equal_the_host_route(tmp_path):
    """GNN.evaluate (arg-max + projection in K12, coincidence tables in K15, Dice from the
    tables) returns exactly what the reference-shaped host helper computes from the same
    predictions with numpy masks."""
    from data_processing.data_loader import ImageGraphDataset
    from model.gnn_model import GNN
    from utils.hyperparam_helpers import FullParamSet

    data = str(tmp_path / "data") + "/"
    write_dataset(data, 4)
    with redirect_stdout(io.StringIO()):
        ds = ImageGraphDataset(data, "BraTS_", read_image=False, read_graph=True, read_label=True)
        hp = FullParamSet(3, 20, 4, 5e-3, 0.98, 1e-4, [0.1, 1, 2, 2], [64, 64], 0, None, None)
        torch.manual_seed(0)
        model = GNN("GSpool", hp, ds, batch_size=2)
    for _ in range(3):
        model.run_epoch()
    subset = torch.utils.data.Subset(ds, [0, 1, 2, 3])
    metrics, counts = model.evaluate(subset)                     # one batched forward of the four samples
    for batch_size in (1, 3):                                    # one at a time / a ragged last batch: same doubles
        again = model.evaluate(subset, batch_size=batch_size)
        assert np.array_equal(again[0], metrics) and np.array_equal(again[1], counts)
    rows, count_rows = [], []
    for mri_id, graph, feats, labels in subset:
        with torch.no_grad():
            logits = model.net(graph.to("cuda"), torch.FloatTensor(feats).to("cuda"))
            loss = model.loss_fcn(logits, torch.LongTensor(labels).to("cuda"))
        c, m = model.calculate_all_metrics_for_brain(mri_id, subset, logits.argmax(1).cpu().numpy(),
                                                     np.asarray(labels))
        rows.append(np.concatenate([[loss.item()], m]))
        count_rows.append(c)
    assert metrics.shape == (10,) and counts.shape == (8,)
    assert np.array_equal(metrics, np.mean(np.array(rows), axis=0))
    assert np.array_equal(counts, np.sum(np.array(count_rows), axis=0))
    assert len(set(np.concatenate(count_rows)[:4] > 0)) >= 1 and counts[4:].sum() == sum(len(s[3]) for s in subset)
