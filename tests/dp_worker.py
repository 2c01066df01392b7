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
