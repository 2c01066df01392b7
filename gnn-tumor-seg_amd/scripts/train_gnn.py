"""Train the supervoxel GNN (k-fold validation or full dataset) on MI355X.

Command line, outputs and file names match /root/reference/scripts/train_gnn.py:64-89:
  python -m scripts.train_gnn -d DATA -o LOGDIR -r RUN [-m GSpool|GSmean|GSgcn|GAT] [-k FOLDS]
                              [-p PREFIX] [-x]
Launch under `torchrun --nproc-per-node N` for data-parallel training (one rank per GPU).  By default every
rank then takes the reference's batch of 6 graphs per step (global batch 6 N at the reference's learning rate:
weak scaling); `--keep_global_batch` splits the reference's 6 over the ranks instead (ceil(6 / N) each), which is
the optimisation problem of the single-GPU run.  Evaluation is sharded over the ranks either way.
"""
import argparse
import os

from numpy import around, r_
from torch.utils.data import Subset

import Filepaths
from data_processing.data_loader import ImageGraphDataset
from gts import dist as gdist
from model.gnn_model import GNN
from utils.hyperparam_helpers import generate_random_hyperparameters, populate_hardcoded_hyperparameters
from utils.training_helpers import (chunk_dataset_into_folds, create_run_progress_file, train_on_fold,
                                    update_progress_file)


def document_metrics(fp, description, results):
    """Console report + one progress-file row (loss and the three voxel Dice scores)."""
    metrics, counts = around(results[0], 4), results[1]
    report = [
        f"\n#{description} Results#",
        f"Loss: {metrics[0]}",
        f"Predicted Node Counts: {counts[0:4]}",
        f"Label Node Counts: {counts[4:8]}",
        "WT Node Dice: {}, CT Node Dice: {}, ET Node Dice: {}".format(*metrics[1:4]),
        "WT Voxel Dice: {}, CT Voxel Dice: {}, ET Voxel Dice: {}".format(*metrics[4:7]),
        "WT HD95: {}, CT HD95: {}, ET HD95: {}".format(*metrics[7:10]),
    ]
    print("\n".join(report))
    if gdist.world()[0] == 0:
        update_progress_file(fp, description, metrics[0], metrics[4:7])


def train_on_full_dataset(args, hyperparams, progress_file_fd, dataset):
    print("Training on full dataset")
    model = GNN(args.model_type, hyperparams, dataset, keep_global_batch=args.keep_global_batch)
    train_on_fold(model, args.output_dir + os.sep, hyperparams.n_epochs, args.run_name, 1)
    whole = Subset(dataset, range(len(dataset)))
    document_metrics(progress_file_fd, f"{args.run_name}_full", model.evaluate(whole))


def run_k_fold_val(args, hyperparams, progress_file_fd, dataset, k):
    """k-fold cross-validation: fold f trains on everything outside [start_f, end_f)."""
    assert k > 1
    everything = len(dataset)
    for fold, (start, end) in enumerate(chunk_dataset_into_folds(dataset, k), start=1):
        held_out = Subset(dataset, range(start, end))
        training = Subset(dataset, list(r_[0:start, end:everything]))
        print(f"Fold contains {len(training)} examples")
        model = GNN(args.model_type, hyperparams, training, keep_global_batch=args.keep_global_batch)
        train_on_fold(model, args.output_dir + os.sep, hyperparams.n_epochs, args.run_name, fold)
        for split, subset in (("train", training), ("val", held_out)):
            document_metrics(progress_file_fd, f"{args.run_name}_f{fold}_{split}", model.evaluate(subset))


_FLAGS = (
    # short, long, default, type, help
    ("-d", "--data_dir", Filepaths.PROCESSED_DATA_DIR, str, "preprocessed dataset directory"),
    ("-o", "--output_dir", Filepaths.LOG_DIR, str, "where checkpoints and the progress file go"),
    ("-r", "--run_name", None, str, "name under which results are saved"),
    ("-m", "--model_type", "GSpool", str, "GSpool | GSmean | GSgcn | GAT"),
    ("-k", "--num_folds", 5, int, "folds for cross-validation; 1 trains on the full dataset"),
    ("-p", "--data_prefix", "", str, "common prefix of the sample folders, e.g. BraTS2021"),
)


def build_parser():
    """Same flags as the reference CLI (scripts/train_gnn.py:66-76)."""
    parser = argparse.ArgumentParser(description="Train the supervoxel GNN on MI355X")
    for short, long_name, default, kind, text in _FLAGS:
        parser.add_argument(short, long_name, default=default, type=kind, help=text)
    parser.add_argument("-x", "--random_hyperparams", default=False, action="store_true",
                        help="draw random hyper-parameters")
    parser.add_argument("--keep_global_batch", default=False, action="store_true",
                        help="data-parallel runs: split the reference's batch of 6 graphs over the ranks instead of "
                             "giving every rank 6 (not a flag of the reference, which is single-device)")
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.num_folds < 1:
        raise ValueError("Number of folds must be a positive integer")
    rank = gdist.init_from_env()[0]
    args.output_dir = os.path.expanduser(args.output_dir)
    dataset = ImageGraphDataset(os.path.expanduser(args.data_dir), args.data_prefix, read_image=False,
                                read_graph=True, read_label=True)
    pick = generate_random_hyperparameters if args.random_hyperparams else populate_hardcoded_hyperparameters
    # ONE draw for the whole job: the random search seeds itself from the wall clock of the calling
    # process (utils/hyperparam_helpers.py), so ranks drawing on their own would build different
    # networks.  Rank 0 draws, every rank receives its tuple.
    hyperparams = gdist.broadcast_object(pick(args.model_type) if rank == 0 else None)
    progress_file_fd = f"{args.output_dir}{os.sep}{args.run_name}.txt"
    if rank == 0:
        create_run_progress_file(progress_file_fd, args.model_type, hyperparams)
    if args.num_folds == 1:
        train_on_full_dataset(args, hyperparams, progress_file_fd, dataset)
    else:
        run_k_fold_val(args, hyperparams, progress_file_fd, dataset, args.num_folds)


if __name__ == "__main__":
    main()
