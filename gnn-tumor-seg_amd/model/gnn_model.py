"""Training / evaluation harness of the GNN on MI355X.

Counterpart of /root/reference/model/gnn_model.py:21-90 — same class, constructor
`GNN(model_type, hyperparameters, train_dataset)`, attributes (`net`, `optimizer`,
`lr_decay`, `loss_fcn`, `train_loader`, `device`) and methods (`run_epoch() -> float`,
`evaluate(Subset) -> (metrics[10], counts[8])`, `save_weights(folder, name)`), so
scripts/train_gnn.py drives it unchanged.

Differences that do not change results:
  * the device must be an AMD GPU — there is no CPU path (the HIP library is the product);
  * AdamW (:28) runs as one HIP launch over a flat parameter buffer (`gts.optim.FlatAdamW`, a
    `torch.optim.Optimizer` with the same update rule and a single param group);
  * `ExponentialLR(..., verbose=False)` (reference :29) raises on current PyTorch; the kwarg
    is dropped, behaviour is the same;
  * the per-step `loss.item()` host sync (:43) becomes one read-back per epoch (same values), and
    the next batch is collated and uploaded by a worker thread while the current step runs
    (`prefetch=True`; same batches, same order);
  * when torch.distributed is initialised with world_size > 1, every rank trains on its
    share of each global batch and gradients are combined by gts.dist.FlatGradSync (exact
    weighted-CE normalisation); world_size == 1 follows the reference's arithmetic exactly.
"""
import os
import sys

import numpy as np
import torch
from torch.utils.data import DataLoader

from data_processing.data_loader import ImageGraphDataset, minibatch_graphs
from data_processing.graph_io import project_nodes_to_img
from gts import collate as gcollate
from gts import dense as gdense
from gts import dist as gdist
from gts import nn as gnn
from gts import ops as gops
from gts.graph import PinnedRing, _upload, uploads_through
from gts.optim import FlatAdamW

from . import evaluation
from .networks import init_graph_net

BATCH_SIZE = 6
EVAL_BATCH_SIZE = 8     # samples per batched evaluation forward (the reference evaluates one at a time)


class _ShardedBatches:
    """Epoch iterator for data-parallel runs: one shared shuffle, per-rank shares of every global
    batch of `per_rank * world_size` samples (dealt by cost when the dataset knows its samples' sizes).  Like the reference's `DataLoader(drop_last=False)`
    (model/gnn_model.py:31) the trailing partial global batch IS trained on: it is dealt round-robin
    too, and a rank whose share of it is empty yields None (it then contributes zero gradients and
    zero loss weight to that step's all-reduce, so the weighted-CE normalisation stays exact)."""

    def __init__(self, dataset, per_rank, rank, world_size, seed=0):
        self.dataset, self.per_rank, self.rank, self.world = dataset, per_rank, rank, world_size
        self.seed, self.epoch = seed, 0
        if len(dataset) == 0:
            raise ValueError("the training dataset is empty: an epoch would have no step")
        # graphs of unequal size are dealt to the ranks by cost inside every global batch (gts.dist.shard_indices)
        self.costs = gdist.sample_costs(dataset)
        if self.costs is not None and len(self.costs) != len(dataset):
            raise ValueError("sample_costs() must return one cost per dataset item")

    def __len__(self):
        g = self.per_rank * self.world
        return (len(self.dataset) + g - 1) // g

    def raw(self):
        """One epoch as lists of samples (None: no share of a short global batch), not yet collated."""
        gen = torch.Generator()
        gen.manual_seed(self.seed + self.epoch)
        self.epoch += 1
        perm = torch.randperm(len(self.dataset), generator=gen).tolist()
        for step in range(len(self)):
            idx = gdist.shard_indices(perm, step, self.per_rank, self.rank, self.world, self.costs)
            yield [self.dataset[i] for i in idx] if idx else None

    def __iter__(self):
        return (None if samples is None else minibatch_graphs(samples) for samples in self.raw())


class GNN:
    """`batch_size` is the number of graphs ONE rank puts into a step (the reference's
    BATCH_SIZE = 6, model/gnn_model.py:12).  Under torch.distributed with W ranks a step therefore
    trains on a global batch of `batch_size * W` graphs (weak scaling, `global_batch_size`) with the
    learning rate unchanged; pass `keep_global_batch=True` to split the reference's batch over the
    ranks instead (per-rank batch ceil(batch_size / W), same optimisation problem as one GPU)."""

    def __init__(self, model_type, hyperparameters, train_dataset, batch_size=BATCH_SIZE, prefetch=True,
                 keep_global_batch=False, host_collate=True):
        if not torch.cuda.is_available():
            raise RuntimeError("GNN needs an AMD GPU (MI355X): the HIP kernels have no CPU fallback")
        self.rank, self.world_size = gdist.world()
        self.prefetch = prefetch
        # True: the loader's batches are assembled by gts_collate_batch (one host C call, one upload: gts/collate.py);
        # False: by minibatch_graphs + per-array uploads, the Python path whose bytes the C path is tested against
        self.host_collate = host_collate and os.environ.get("GTS_HOST_COLLATE", "1") != "0"
        self.device = torch.device("cuda", torch.cuda.current_device())
        print("Using device", self.device)
        class_weights = torch.FloatTensor(hyperparameters.class_weights).to(self.device)
        self.class_weights = class_weights
        self.net = init_graph_net(model_type, hyperparameters)
        self.net.to(self.device)
        # same AdamW update as the reference's torch.optim.AdamW (:28), as one HIP pass over a
        # flat parameter buffer (gts_adamw_f32); a torch Optimizer, so the scheduler below drives it
        self.optimizer = FlatAdamW(self.net.parameters(), lr=hyperparameters.lr,
                                   weight_decay=hyperparameters.w_decay)
        self.lr_decay = torch.optim.lr_scheduler.ExponentialLR(self.optimizer, hyperparameters.lr_decay,
                                                               last_epoch=-1)
        # same function as torch.nn.CrossEntropyLoss(weight=class_weights) (reference :30), as one
        # fused HIP pass (gts_weighted_ce_f32)
        self.loss_fcn = lambda logits, labels: gops.weighted_cross_entropy(logits, labels, class_weights)
        self.grad_sync = None
        self._pinned_ring = None
        self._copy_stream = None
        # the fused layer stack writes its weight gradients straight into one flat buffer laid out like the optimizer's
        # parameters (no per-parameter gradient tensors, nothing to concatenate before the AdamW launch)
        self.grad_sink = gnn.GradSink(self.optimizer._params)
        self.global_batch_size = batch_size
        if train_dataset is None:
            self.train_loader = None
        elif self.world_size > 1:
            # identical replicas: rank 0's initialisation wins (after checking that every rank built
            # the same architecture — ranks that drew different hyper-parameters must not reach RCCL)
            gdist.broadcast_parameters(self.net.parameters(), src=0)
            self.grad_sync = gdist.FlatGradSync(self.net.parameters())
            per_rank = -(-batch_size // self.world_size) if keep_global_batch else batch_size
            self.global_batch_size = per_rank * self.world_size
            self.train_loader = _ShardedBatches(train_dataset, per_rank, self.rank, self.world_size)
        else:
            self.train_loader = DataLoader(train_dataset, batch_size=batch_size, shuffle=True,
                                           num_workers=0, collate_fn=minibatch_graphs)

    def train_step(self, graph, features, labels):
        """One optimizer step on device-resident inputs; returns the loss as a 0-dim tensor."""
        logits = self.net(graph, features)
        if self.grad_sync is None:
            # loss_fcn(logits, labels).backward() without the autograd node of the loss: the fused pass returns
            # d(numerator)/d(logits) and [numerator, denominator, mean]; one in-place division makes it the mean's
            grad, stats = gops.weighted_ce_numerator_grad(logits, labels, self.class_weights)
            grad.div_(stats[1])
            self.optimizer.zero_grad()
            flat = self.grad_sink.new_buffer()
            with gnn.grad_sink(self.grad_sink):
                logits.backward(grad)
            if self.grad_sink.filled:          # the whole network was one fused stack: its gradients are in `flat`
                self.optimizer.step(flat_grad=flat)
            else:
                self.optimizer.step()
            return stats[2]
        self.grad_sync.zero_grad()
        self.grad_sync.weighted_ce_backward(logits, labels, self.class_weights)
        loss = self.grad_sync.all_reduce_and_normalise()
        if isinstance(self.optimizer, FlatAdamW):
            self.optimizer.step(flat_grad=self.grad_sync.flat_gradients())
        else:
            self.optimizer.step()
        return loss                    # a fresh 0-dim tensor (numerator / denominator), nothing aliases it

    def empty_step(self):
        """Data-parallel step of a rank without samples (short last global batch): no forward, zero
        contribution to the all-reduce, the same optimizer update as every other rank."""
        if self.grad_sync is None:
            raise RuntimeError("empty_step() only exists in data-parallel runs")
        self.grad_sync.empty_step(self.device)
        loss = self.grad_sync.all_reduce_and_normalise()
        if isinstance(self.optimizer, FlatAdamW):
            self.optimizer.step(flat_grad=self.grad_sync.flat_gradients())
        else:
            self.optimizer.step()
        return loss.clone()

    def _to_device(self, graph, features, labels=None):
        graph = graph.to(self.device)

        def up(t, dtype):      # in the prefetch thread: through its page-locked ring, asynchronously on the copy stream
            t = torch.as_tensor(np.asarray(t), dtype=dtype) if not isinstance(t, torch.Tensor) else t.to(dtype)
            return t.to(self.device) if t.is_cuda else _upload(t.contiguous(), self.device)
        features = up(features, torch.float32)
        if labels is None:
            return graph, features
        return graph, features, up(labels, torch.int64)

    def _schedules_wanted(self, n_rows):
        """Cluster row schedules the training step of this network will ask the graph for ('in' / 'out': the max-pool
        reducers of SAGEConv-pool at 256 features; 'gat_*': GATConv at 256 features per head): the prefetch thread builds
        and uploads them with the batch instead of leaving that to the first kernel call."""
        from gts import schedule
        from gts.nn import GATConv, SAGEConv

        wanted = ()
        if schedule.ENABLED and any(isinstance(m, SAGEConv) and m._aggre_type == "pool" and m._in_src_feats == 256
                                    for m in self.net.modules()):
            wanted += ("out", "in") if n_rows >= schedule.MIN_ROWS_FORWARD else ("out",)
        if schedule.ENABLED_GAT and n_rows >= schedule.MIN_ROWS_GAT and any(
                isinstance(m, GATConv) and m._out_feats == 256 for m in self.net.modules()):
            wanted += ("gat_in", "gat_edge_in", "gat_out")
        return wanted

    def _raw_batches(self):
        """The training loader's batches as lists of samples (None: an empty share), in the loader's order and drawing
        from the same random streams, or None when the loader is not one this class built (then it is iterated as is)."""
        loader = self.train_loader
        if isinstance(loader, _ShardedBatches):
            return loader.raw()
        if isinstance(loader, DataLoader) and loader.collate_fn is minibatch_graphs and loader.num_workers == 0:
            # the iterator binds the collate function when it is created: the loader itself is left as it was
            loader.collate_fn = list
            try:
                return iter(loader)
            finally:
                loader.collate_fn = minibatch_graphs
        return None

    def _device_batches(self):
        """The loader's batches, already on the GPU, prepared one step ahead: a worker thread
        collates batch i+1 (JSON / cache read, block-diagonal union) and uploads it on a copy
        stream while the main thread enqueues step i.  Same batches in the same order as iterating
        the loader directly (`prefetch=False`).

        With `host_collate` (the default) a batch is assembled by ONE host C call into a page-locked slab and uploaded
        by ONE copy (gts/collate.py): the worker holds the interpreter lock for tens of microseconds per batch."""
        raw = self._raw_batches() if self.host_collate else None
        if self._pinned_ring is None and (raw is not None or self.prefetch):
            self._pinned_ring = PinnedRing()       # page-locked staging slabs: allocated once per GNN, reused every batch
        ring = self._pinned_ring
        collator = gcollate.HostCollator(self.device, ring, self._schedules_wanted) if raw is not None else None
        if not self.prefetch:
            if collator is not None:
                for samples in raw:
                    if samples is None:
                        yield None
                        continue
                    _ids, graph, feats, labels = collator(samples)
                    ring.next_batch()
                    yield graph, feats, labels
                return
            for item in self.train_loader:
                yield None if item is None else self._to_device(*item[1:])
            return
        import queue
        import threading

        slots, stop = queue.Queue(maxsize=2), threading.Event()
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=self.device)
        copy_stream = self._copy_stream
        device_index = self.device.index

        import time
        trace = [] if os.environ.get("GTS_PREFETCH_TRACE") else None     # diagnostic: seconds per stage of the loader thread
        t_prev = [time.perf_counter()]

        def produce():
            try:
                torch.cuda.set_device(device_index)
                uploads_through(ring)
                for item in (raw if raw is not None else self.train_loader):
                    if stop.is_set():
                        return
                    if item is None:                       # no share of this (short) global batch
                        slots.put((None, None))
                        continue
                    t0 = time.perf_counter()
                    with torch.cuda.stream(copy_stream):
                        if collator is not None:
                            batch = collator(item)[1:]
                            t1 = t2 = t3 = time.perf_counter()
                        else:
                            _ids, graph, feats, labels = item
                            batch = self._to_device(graph, feats, labels)
                            t1 = time.perf_counter()
                            batch[0].dev()                     # CSR upload belongs to the copy as well
                            t2 = time.perf_counter()
                            for which in self._schedules_wanted(batch[0].n):
                                batch[0].dev_schedule(which)   # ... and so do the cluster row schedules the step will read
                            t3 = time.perf_counter()
                        ready = torch.cuda.Event()
                        ready.record(copy_stream)
                        ring.next_batch()
                        t4 = time.perf_counter()
                    slots.put((batch, ready))
                    if trace is not None:
                        trace.append((t0 - t_prev[0], t1 - t0, t2 - t1, t3 - t2, t4 - t3, time.perf_counter() - t4))
                        t_prev[0] = time.perf_counter()
                slots.put(None)
            except BaseException as exc:                   # noqa: BLE001 - re-raised in the consumer
                slots.put(exc)

        worker = threading.Thread(target=produce, name="gts-batch-prefetch", daemon=True)
        worker.start()
        try:
            while True:
                item = slots.get()
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                if item[0] is None:
                    yield None
                    continue
                (graph, feats, labels), ready = item
                main = torch.cuda.current_stream()
                main.wait_event(ready)
                graph.dev().record_stream(main)            # allocated on the copy stream, read here
                feats.record_stream(main)
                labels.record_stream(main)
                yield graph, feats, labels
        finally:
            if trace:
                mean = [1e3 * sum(col) / len(trace) for col in zip(*trace)]
                print("[gts prefetch] ms per batch: loader %.2f | %s %.2f | csr %.2f | schedules %.2f | event+ring %.2f | "
                      "queue %.2f  (%d batches)" % (mean[0], "host collate + upload" if collator is not None else "features+labels",
                                                    *mean[1:], len(trace)), file=sys.stderr, flush=True)
            stop.set()
            while worker.is_alive():                       # unblock a producer waiting on a full queue
                try:
                    slots.get_nowait()
                except queue.Empty:
                    worker.join(timeout=0.05)

    def run_epoch(self):
        """One pass over the training loader; returns the mean of the per-step losses
        (reference :34-48).  Losses stay on the device until the epoch ends."""
        self.net.train()
        step_losses = [self.empty_step() if batch is None else self.train_step(*batch)
                       for batch in self._device_batches()]
        self.lr_decay.step()
        return np.mean(torch.stack(step_losses).cpu().double().numpy())

    def evaluate(self, dataset: ImageGraphDataset, batch_size=EVAL_BATCH_SIZE):
        """`dataset` is a torch Subset of an ImageGraphDataset with labels (what
        scripts/train_gnn.py passes).  Per sample: no-grad forward, weighted CE, arg-max; node
        metrics, then voxel metrics after projecting the predictions onto the supervoxel
        partitioning (reference :51-87).  Returns (mean of the [n,10] metric rows, sum of the
        [n,8] label-count rows); metric columns: loss | node Dice WT,CT,ET | voxel Dice WT,CT,ET |
        voxel HD95 WT,CT,ET.

        The forwards are BATCHED (`batch_size` samples per block-diagonal union, one launch chain
        instead of one per sample); rows of a block-diagonal batch never mix, and the GEMMs are pinned
        to the tile family whose rounding does not depend on the row count, so every logit — hence
        every loss, count and Dice — is identical for ANY `batch_size` of this method, `batch_size=1` (the
        reference's one-sample-at-a-time route, under the same tile pin) included (tests/test_gpu_cli.py).
        A forward OUTSIDE this method runs under the automatic tile choice and may round differently.  Everything up to the Dice quotients stays on the GPU:
        arg-max + projection is one K12 pass per sample, the node- and voxel-level label
        coincidences are counted by K15, and only the two 5x5 integer tables, the loss and the
        predicted volume (for the scipy distance transforms of HD95) travel to the host.

        Under torch.distributed the samples are dealt round-robin over the ranks (rank r evaluates r, r + W, ...:
        forwards AND the host-side HD95 transforms), the per-sample rows are gathered on the host and every rank
        returns the same two arrays — the doubles a single process computes, because a sample's row does not
        depend on what it is batched with (tests/test_gpu_dp.py)."""
        assert dataset.dataset.read_label == True  # noqa: E712
        self.net.eval()
        source = dataset.dataset        # Subset -> underlying ImageGraphDataset
        n_samples = len(dataset)
        share = gdist.rank_share(n_samples)
        rows = {}
        step = max(1, int(batch_size))
        for first in range(0, len(share), step):
            indices = share[first:first + step]
            samples = [dataset[i] for i in indices]
            ids, graph, feats, labels = minibatch_graphs(samples)
            graph, feats, labels = self._to_device(graph, feats, labels)
            with torch.no_grad(), gdense.row_count_invariant():
                all_logits = self.net(graph, feats)
            ends = np.cumsum([s[1].number_of_nodes() for s in samples])
            for j, mri_id in enumerate(ids):
                lo, hi = int(ends[j - 1]) if j else 0, int(ends[j])
                logits, node_labels = all_logits[lo:hi], labels[lo:hi]
                partitioning = torch.from_numpy(source.get_supervoxel_partitioning(mri_id)).to(self.device)
                true_voxels = source.get_voxel_labels(mri_id)
                with torch.no_grad():
                    loss = self.loss_fcn(logits, node_labels)
                    predicted = torch.max(logits, dim=1)[1]
                    node_table = gops.label_confusion(predicted.to(torch.int16), node_labels.to(torch.int16))
                    predicted_voxels = gops.project_argmax(partitioning, logits)          # K12 + arg-max
                    voxel_table = gops.label_confusion(                                  # K15
                        predicted_voxels, torch.from_numpy(true_voxels).to(self.device).contiguous())
                tables = torch.stack([node_table, voxel_table]).cpu().numpy()
                hd95s = evaluation.calculate_hd95s(predicted_voxels.cpu().numpy(), true_voxels)
                rows[indices[j]] = (np.concatenate([[loss.item()], evaluation.dices_from_confusion(tables[0]),
                                                    evaluation.dices_from_confusion(tables[1]), hd95s]),
                                    evaluation.label_counts_from_confusion(tables[0]))
        rows = gdist.gather_rows_in_order(n_samples, rows)
        metrics = np.array([r[0] for r in rows]).reshape(n_samples, 10)
        counts = np.array([r[1] for r in rows]).reshape(n_samples, 8)
        return np.mean(metrics, axis=0), np.sum(counts, axis=0)

    def calculate_all_metrics_for_brain(self, mri_id, dataset, node_preds, node_labels):
        """(label counts [pred x4, truth x4], [node Dice x3, voxel Dice x3, voxel HD95 x3]) from
        host arrays of node predictions / labels — the reference's helper (:76-87), kept for its
        callers; `evaluate` computes the same numbers from device-side counts."""
        source = dataset.dataset        # Subset -> underlying ImageGraphDataset
        label_counts = np.concatenate([evaluation.count_node_labels(node_preds),
                                       evaluation.count_node_labels(node_labels)])
        node_dices = evaluation.calculate_node_dices(node_preds, node_labels)
        predicted_voxels = project_nodes_to_img(source.get_supervoxel_partitioning(mri_id), node_preds)  # K12
        voxel_metrics = evaluation.calculate_brats_metrics(predicted_voxels, source.get_voxel_labels(mri_id))
        return label_counts, np.concatenate([node_dices, voxel_metrics])

    def save_weights(self, folder, name):
        if self.rank == 0:
            torch.save(self.net.state_dict(), f"{folder}{name}.pt")
