"""Epoch loop, fold bookkeeping and the run-progress text file.

Counterpart of /root/reference/utils/training_helpers.py:7-57; file format, checkpoint
naming (`{run_name}_f{fold}`) and the early-stop rule are kept exactly.
"""

_HEADER_ROWS = (
    ("Epochs", "n_epochs"),
    ("Input Features", "in_feats"),
    ("LR", "lr"),
    ("L2Reg", "w_decay"),
    ("LR Decay", "lr_decay"),
    ("Layer Sizes", "layer_sizes"),
)
_GAT_ROWS = (("Heads", "gat_heads"), ("Residuals", "gat_residuals"))


def create_run_progress_file(fp, model_type, hp):
    """Start the progress file with the hyper-parameters of the run (tab separated)."""
    lines = ["----Model Parameters----", f"Model\t{model_type}"]
    lines += [f"{title}\t{getattr(hp, field)}" for title, field in _HEADER_ROWS]
    if model_type == "GAT":
        lines += [f"{title}\t{getattr(hp, field)}" for title, field in _GAT_ROWS]
    lines.append("Fold\tLoss\tWT_Dice\tCT_Dice\tET_Dice\n")
    with open(fp, "w") as f:
        f.write("\n".join(lines) + "\n")


def chunk_dataset_into_folds(dataset, k):
    """k equal [start, end) index ranges; a remainder of len(dataset) % k samples is unused."""
    size = len(dataset) // k
    return [(i * size, (i + 1) * size) for i in range(k)]


def update_progress_file(fp, description, loss, dices):
    with open(fp, "a") as f:
        f.write("\t".join(str(x) for x in (description, loss, dices[0], dices[1], dices[2])) + "\n")


def train_on_fold(model, checkpoint_dir, n_epoch, run_name, fold):
    """Run `model.run_epoch()` up to n_epoch times; checkpoint whenever the epoch loss is a new
    minimum; after half of the epochs stop once the loss exceeds the minimum by > 0.001."""
    lowest_loss = 1000
    for epoch in range(1, n_epoch + 1):
        epoch_loss = model.run_epoch()
        print(f"____Epoch {epoch}_____")
        print(epoch_loss)
        if epoch > n_epoch / 2 and epoch_loss > lowest_loss + 0.001:
            print("Fold terminated early due to converged train loss")
            print(f"Ran for {epoch} epochs")
            return
        if epoch_loss < lowest_loss:
            lowest_loss = epoch_loss
            model.save_weights(checkpoint_dir, f"{run_name}_f{fold}")
    print(f"Finished fold {fold} for run {run_name}")
