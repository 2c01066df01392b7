"""Worker for the two-rank data-parallel test (spawned; must be importable)."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "gnn-tumor-seg_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


class MemDataset(torch.utils.data.Dataset):
    """In-memory stand-in for ImageGraphDataset (same item layout)."""

    def __init__(self, n_samples, n=300, in_feats=20):
        from gts import synth

        self.items = []
        for i in range(n_samples):
            g = synth.random_graph(n=n, n_pairs=2 * n, seed=1000 + i)
            self.items.append((f"s{i}", g, synth.node_features(n, in_feats, 1000 + i).astype(np.float64),
                               synth.node_labels(n, 1000 + i)))
        self.read_label = True

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


class UnequalDataset(MemDataset):
    """Graphs of unequal size (50 .. 70 nodes: the 5 - 7k spread of real supervoxel graphs, scaled down) that know their
    costs, as ImageGraphDataset does through its file sizes."""

    def __init__(self, n_samples, in_feats=20, seed=0):
        from gts import synth

        sizes = np.random.default_rng(seed).integers(50, 71, size=n_samples)
        self.items = []
        for i, n in enumerate(sizes):
            g = synth.random_graph(n=int(n), n_pairs=2 * int(n), seed=1000 + i)
            self.items.append((f"s{i}", g, synth.node_features(int(n), in_feats, 1000 + i).astype(np.float64),
                               synth.node_labels(int(n), 1000 + i)))
        self.read_label = True

    def sample_costs(self):
        return [item[1].n for item in self.items]


def hyperparams():
    from utils.hyperparam_helpers import FullParamSet

    return FullParamSet(1, 20, 4, 1e-3, 0.98, 1e-4, [0.1, 1, 2, 2], [64, 64], 0, None, None)


def run_rank(rank, world, port, n_samples, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), GTS_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import io
    from contextlib import redirect_stdout

    from gts import dist as gdist
    from model.gnn_model import GNN

    gdist.init_from_env()
    torch.manual_seed(100 + rank)            # different init per rank: rank 0's must win (broadcast)
    with redirect_stdout(io.StringIO()):
        model = GNN("GSpool", hyperparams(), MemDataset(n_samples), batch_size=1)
    loss = model.run_epoch()
    torch.save({"loss": float(loss), "state": {k: v.cpu() for k, v in model.net.state_dict().items()}},
               os.path.join(out_dir, f"rank{rank}.pt"))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def run_eval_rank(rank, world, port, data_dir, ckpt, out_dir):
    """One rank of a sharded GNN.evaluate + prediction run over the file dataset in `data_dir` (two ranks share
    the one GPU; gloo carries the host-side gather)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), GTS_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import io
    from contextlib import redirect_stdout

    from data_processing.data_loader import ImageGraphDataset
    from gts import dist as gdist
    from model.gnn_model import GNN
    from scripts import generate_gnn_predictions as gen
    from utils.hyperparam_helpers import FullParamSet

    gdist.init_from_env()
    with redirect_stdout(io.StringIO()):
        ds = ImageGraphDataset(data_dir, "BraTS_", read_image=False, read_graph=True, read_label=True)
        hp = FullParamSet(3, 20, 4, 5e-3, 0.98, 1e-4, [0.1, 1, 2, 2], [64, 64], 0, None, None)
        model = GNN("GSpool", hp, None)
        model.net.load_state_dict(torch.load(ckpt, map_location=model.device, weights_only=True))
        metrics, counts = model.evaluate(torch.utils.data.Subset(ds, list(range(len(ds)))), batch_size=2)
        unl = ImageGraphDataset(data_dir, "BraTS_", read_image=False, read_graph=True, read_label=False)
        gen.output_dir = os.path.join(out_dir, "preds")
        os.makedirs(gen.output_dir, exist_ok=True)
        gen.save_predictions(model.net, unl, "preds")
    np.savez(os.path.join(out_dir, f"eval{rank}.npz"), metrics=metrics, counts=counts)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
