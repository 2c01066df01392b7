"""Train the supervoxel GNN (k-fold validation or full dataset) on MI355X.

Command line, outputs and file names match /root/reference/scripts/train_gnn.py:64-89:
  python -m scripts.train_gnn -d DATA -o LOGDIR -r RUN [-m GSpool|GSmean|GSgcn|GAT] [-k FOLDS]
                              [-p PREFIX] [-x]
Launch under `torchrun --nproc-per-node N` for data-parallel training (one rank per GPU).
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
    metrics, counts = around(results[0], 4), results[1]
    print(f"\n#{description} Results#")
    print("Loss:", metrics[0])
    print("Predicted Node Counts:", counts[0:4])
    print("Label Node Counts:", counts[4:8])
    print(f"WT Node Dice: {metrics[1]}, CT Node Dice: {metrics[2]}, ET Node Dice: {metrics[3]}")
    print(f"WT Voxel Dice: {metrics[4]}, CT Voxel Dice: {metrics[5]}, ET Voxel Dice: {metrics[6]}")
    print(f"WT HD95: {metrics[7]}, CT HD95: {metrics[8]}, ET HD95: {metrics[9]}")
    if gdist.world()[0] == 0:
        update_progress_file(fp, description, metrics[0], metrics[4:7])


def train_on_full_dataset(args, hyperparams, progress_file_fd, dataset):
    print("Training on full dataset")
    model = GNN(args.model_type, hyperparams, dataset)
    train_on_fold(model, args.output_dir + os.sep, hyperparams.n_epochs, args.run_name, 1)
    everything = Subset(dataset, range(0, len(dataset)))
    document_metrics(progress_file_fd, f"{args.run_name}_full", model.evaluate(everything))


def run_k_fold_val(args, hyperparams, progress_file_fd, dataset, k):
    assert k > 1
    for fold, (start, end) in enumerate(chunk_dataset_into_folds(dataset, k), start=1):
        val_dataset = Subset(dataset, range(start, end))
        train_dataset = Subset(dataset, list(r_[0:start, end:len(dataset)]))
        print(f"Fold contains {len(train_dataset)} examples")
        model = GNN(args.model_type, hyperparams, train_dataset)
        train_on_fold(model, args.output_dir + os.sep, hyperparams.n_epochs, args.run_name, fold)
        document_metrics(progress_file_fd, f"{args.run_name}_f{fold}_train", model.evaluate(train_dataset))
        document_metrics(progress_file_fd, f"{args.run_name}_f{fold}_val", model.evaluate(val_dataset))


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument("-d", "--data_dir", default=Filepaths.PROCESSED_DATA_DIR, type=str,
                        help="path to the directory where data is stored")
    parser.add_argument("-o", "--output_dir", default=Filepaths.LOG_DIR, type=str, help="Log directory")
    parser.add_argument("-r", "--run_name", default=None, type=str, help="A unique name to save results under")
    parser.add_argument("-m", "--model_type", default="GSpool", type=str,
                        help="What graph learning layer to use. GSpool, GSmean, GSgcn, GAT")
    parser.add_argument("-k", "--num_folds", default=5, type=int,
                        help="How many folds to run k fold validation on. 1== train on full dataset")
    parser.add_argument("-p", "--data_prefix", default="", type=str,
                        help="A prefix that all data folders share, i.e. BraTS2021.")
    parser.add_argument("-x", "--random_hyperparams", default=False, action="store_true",
                        help="whether to generate random hyperparameters")
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    rank, _, _ = gdist.init_from_env()
    dataset = ImageGraphDataset(os.path.expanduser(args.data_dir), args.data_prefix, read_image=False,
                                read_graph=True, read_label=True)
    hyperparams = generate_random_hyperparameters(args.model_type) if args.random_hyperparams \
        else populate_hardcoded_hyperparameters(args.model_type)
    args.output_dir = os.path.expanduser(args.output_dir)
    progress_file_fd = f"{args.output_dir}{os.sep}{args.run_name}.txt"
    if rank == 0:
        create_run_progress_file(progress_file_fd, args.model_type, hyperparams)
    if args.num_folds == 1:
        train_on_full_dataset(args, hyperparams, progress_file_fd, dataset)
    elif args.num_folds > 1:
        run_k_fold_val(args, hyperparams, progress_file_fd, dataset, args.num_folds)
    else:
        raise ValueError("Number of folds must be a positive integer")


if __name__ == "__main__":
    main()
