"""Two data-parallel ranks sharing the one GPU of the test box (gloo carries the all-reduce;
RCCL needs one GPU per rank) train exactly like one process on the global batches."""
import io
import socket
from contextlib import redirect_stdout

import pytest
import torch
import torch.multiprocessing as mp

import gts
from tests import dp_worker

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(600)
def test_two_rank_epoch_matches_single_process_global_batches(tmp_path, hip_lib):
    from gts import dist as gdist
    from model.gnn_model import GNN
    from model.networks import init_graph_net

    n_samples, world = 4, 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(dp_worker.run_rank, args=(world, port, n_samples, str(tmp_path)), nprocs=world, join=True)
    ranks = [torch.load(tmp_path / f"rank{r}.pt", weights_only=True) for r in range(world)]
    for k in ranks[0]["state"]:
        assert torch.equal(ranks[0]["state"][k], ranks[1]["state"][k]), k      # replicas stay identical
    assert ranks[0]["loss"] == ranks[1]["loss"]

    # the same epoch in one process: global batch s = graphs perm[2s], perm[2s+1]
    hp = dp_worker.hyperparams()
    data = dp_worker.MemDataset(n_samples)
    torch.manual_seed(100)                                   # rank 0's initialisation
    with redirect_stdout(io.StringIO()):
        single = GNN("GSpool", hp, None)
    gen = torch.Generator()
    gen.manual_seed(0)
    perm = torch.randperm(n_samples, generator=gen).tolist()
    single.net.train()
    losses = []
    for step in range(n_samples // world):
        idx = [i for r in range(world) for i in gdist.shard_indices(perm, step, 1, r, world)]
        items = [data[i] for i in idx]
        g = gts.batch([it[1] for it in items]).to(single.device)
        feats = torch.cat([torch.FloatTensor(it[2]) for it in items]).to(single.device)
        labels = torch.cat([torch.LongTensor(it[3]) for it in items]).to(single.device)
        losses.append(float(single.train_step(g, feats, labels)))
    assert abs(sum(losses) / len(losses) - ranks[0]["loss"]) < 1e-5 * max(1.0, abs(ranks[0]["loss"]))
    for k, v in single.net.state_dict().items():
        assert torch.allclose(v.cpu(), ranks[0]["state"][k], rtol=1e-4, atol=1e-6), k
    assert isinstance(init_graph_net("GSpool", hp), torch.nn.Module)


@pytest.mark.timeout(600)
def test_two_rank_evaluate_and_predict_equal_the_single_process_run(tmp_path, hip_lib):
    """GNN.evaluate shards the samples over the ranks (dataset[rank::W]) and gathers the per-sample rows on the host:
    every rank returns the doubles of the single-process evaluation (reference loop:
    /root/reference/model/gnn_model.py:51-74); the prediction script deals the volumes the same way and the union of
    what the ranks wrote equals the single-process output, byte for byte."""
    import os

    import numpy as np

    from data_processing.data_loader import ImageGraphDataset
    from model.gnn_model import GNN
    from scripts import generate_gnn_predictions as gen
    from tests.dataset_util import write_dataset
    from utils.hyperparam_helpers import FullParamSet

    data = str(tmp_path / "data") + "/"
    write_dataset(data, 5)
    with redirect_stdout(io.StringIO()):
        ds = ImageGraphDataset(data, "BraTS_", read_image=False, read_graph=True, read_label=True)
        hp = FullParamSet(3, 20, 4, 5e-3, 0.98, 1e-4, [0.1, 1, 2, 2], [64, 64], 0, None, None)
        torch.manual_seed(0)
        model = GNN("GSpool", hp, ds, batch_size=2)
        for _ in range(2):
            model.run_epoch()
        ckpt = str(tmp_path / "w.pt")
        torch.save(model.net.state_dict(), ckpt)
        want = model.evaluate(torch.utils.data.Subset(ds, list(range(len(ds)))))
        single_dir = str(tmp_path / "single")
        os.makedirs(single_dir)
        gen.output_dir = single_dir
        gen.save_predictions(model.net, ImageGraphDataset(data, "BraTS_", read_image=False, read_graph=True,
                                                          read_label=False), "preds")
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out_dir = str(tmp_path / "ranks")
    os.makedirs(out_dir)
    mp.spawn(dp_worker.run_eval_rank, args=(world, port, data, ckpt, out_dir), nprocs=world, join=True)
    for r in range(world):
        got = np.load(os.path.join(out_dir, f"eval{r}.npz"))
        assert np.array_equal(got["metrics"], want[0]) and np.array_equal(got["counts"], want[1])
    files = sorted(os.listdir(single_dir))
    assert len(files) == 5 and sorted(os.listdir(os.path.join(out_dir, "preds"))) == files
    import gzip
    for f in files:      # the gzip header carries a time stamp: compare the volumes inside
        assert gzip.open(os.path.join(single_dir, f)).read() == gzip.open(os.path.join(out_dir, "preds", f)).read()
