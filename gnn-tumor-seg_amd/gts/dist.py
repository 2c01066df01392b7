"""Data-parallel training support: one process per MI355X, RCCL over xGMI.

The reference is single-device (model/gnn_model.py:23); this is the new exchange step of
the path.  Graphs are independent samples, so ranks never exchange
activations — only one all-reduce per optimizer step over ONE flat fp32 buffer that holds
every parameter gradient plus two scalars:

    flat = [ grad(theta) ... | sum_i w[y_i] | sum_i w[y_i] * nll_i ]

`CrossEntropyLoss(weight=w)` (model/gnn_model.py:30) is a weighted mean, so the global-batch
gradient is  (sum over ranks of grad(numerator_r)) / (sum over ranks of denominator_r), NOT
the average of per-rank mean-loss gradients.  Each rank therefore back-propagates its local
numerator (reduction='sum'), the numerators/denominators ride in the same all-reduce, and
the division happens once afterwards.  The payload is ~5 MB: latency-bound, a single
collective, no bucketing needed.  The flat buffer is built by ONE concatenation kernel after
backward (gradients are not accumulated into pre-attached views: that would cost one small
add kernel per parameter per step).
"""
import os

import torch
import torch.distributed as dist


# bench.py installs a callable here (launch -> result) to bracket the gradient all-reduce with HIP
# events; None in normal use.
COLLECTIVE_TIMER = None


def world():
    """(rank, world_size) of the default process group, (0, 1) when not initialised."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's environment (RANK / WORLD_SIZE / LOCAL_RANK /
    MASTER_ADDR / MASTER_PORT).  backend defaults to 'nccl' (= RCCL on ROCm) when a GPU is
    visible, else 'gloo'.  Returns (rank, world_size, local_rank)."""
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_gpu = torch.cuda.is_available()
    if use_gpu:
        # one GPU per rank; ranks wrap around only in single-GPU rehearsals (GTS_DIST_BACKEND=gloo)
        local_rank = local_rank % torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
    if world_size > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = backend or os.environ.get("GTS_DIST_BACKEND") or ("nccl" if use_gpu else "gloo")
        kwargs = {}
        if use_gpu and backend == "nccl":
            kwargs["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world_size, **kwargs)
    return rank, world_size, local_rank


def shard_indices(perm, step, per_rank, rank, world_size, costs=None):
    """Indices of rank `rank` for global step `step`: the global batch is
    perm[step*G:(step+1)*G] with G = per_rank*world_size.  The last global batch of an epoch may be
    short; a rank whose share of it is empty gets [].

    costs is None: dealt round-robin (r::W) — right when every graph costs the same (the synthetic
    15k-node graphs of the benchmark).
    costs[i] = cost of dataset item i (node count, or a proxy such as the size of its graph file): the
    global batch — the SAME set of graphs, hence the same exact gradient after the all-reduce — is
    dealt by longest-processing-time: graphs in order of falling cost (ties: position in the batch) go
    to the rank with the least cost so far that still has room (at most `per_rank` graphs; ties: lowest
    rank).  Real supervoxel graphs have 5 - 7k nodes (/root/reference/mri2graph/graphgen.py:210-211): dealt
    blindly, every step waits for the rank that drew the largest ones.  Deterministic, no communication:
    every rank computes the whole deal.  A rank's graphs keep their order in the global batch."""
    g = per_rank * world_size
    chunk = list(perm[step * g:(step + 1) * g])
    if costs is None or world_size == 1:
        return chunk[rank::world_size]
    order = sorted(range(len(chunk)), key=lambda k: (-costs[chunk[k]], k))
    load, held = [0] * world_size, [[] for _ in range(world_size)]
    for k in order:
        r = min((r for r in range(world_size) if len(held[r]) < per_rank), key=lambda r: (load[r], r))
        held[r].append(k)
        load[r] += costs[chunk[k]]
    return [chunk[k] for k in sorted(held[rank])]


def sample_costs(dataset):
    """Per-item cost of `dataset` for shard_indices, identical on every rank, WITHOUT loading the samples: the
    dataset's own `sample_costs()` (ImageGraphDataset: bytes of each graph file; in-memory datasets: node counts),
    followed through torch Subsets.  None when the dataset offers none (round-robin dealing then)."""
    indices = None
    while hasattr(dataset, "dataset") and hasattr(dataset, "indices"):        # torch.utils.data.Subset
        own = list(dataset.indices)
        indices = own if indices is None else [own[i] for i in indices]
        dataset = dataset.dataset
    fn = getattr(dataset, "sample_costs", None)
    if fn is None:
        return None
    costs = list(fn())
    return costs if indices is None else [costs[i] for i in indices]


def rank_share(n_items):
    """Indices of this rank when `n_items` independent items (evaluation samples, volumes to predict) are dealt
    round-robin over the ranks: rank r takes r, r + W, ...  Everything when torch.distributed is not initialised."""
    rank, w = world()
    return list(range(rank, n_items, w))


def gather_rows_in_order(n_items, mine):
    """`mine` maps the indices of rank_share(n_items) to picklable rows (metric vectors of this rank's samples).
    Returns the list of all n_items rows in index order, identical on every rank (one all_gather_object on
    the host side: evaluation has no device collective)."""
    _, w = world()
    if w == 1:
        everyone = [mine]
    else:
        everyone = [None] * w
        dist.all_gather_object(everyone, mine)
    rows = {}
    for part in everyone:
        rows.update(part)
    missing = [i for i in range(n_items) if i not in rows]
    if missing:
        raise RuntimeError(f"items {missing[:8]} were evaluated by no rank")
    return [rows[i] for i in range(n_items)]


def broadcast_object(obj, src=0):
    """`obj` of rank `src` on every rank (anything picklable: hyper-parameter tuples, seeds).
    Identity when torch.distributed is not initialised."""
    if world()[1] == 1:
        return obj
    box = [obj]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def assert_same_layout(params, what="parameters"):
    """Every rank must hold the same number of tensors with the same shapes before a collective
    walks over them: a mismatch (e.g. ranks that drew different hyper-parameters) would otherwise
    hang RCCL or silently train diverging replicas.  Raises RuntimeError on EVERY rank."""
    rank, w = world()
    if w == 1:
        return
    mine = [tuple(p.shape) for p in params]
    everyone = [None] * w
    dist.all_gather_object(everyone, mine)
    bad = [r for r, shapes in enumerate(everyone) if shapes != everyone[0]]
    if bad:
        raise RuntimeError(f"{what} differ between ranks: rank 0 holds {len(everyone[0])} tensors, "
                           f"ranks {bad} hold {[len(everyone[r]) for r in bad]} (or other shapes); "
                           "build the model from ONE set of hyper-parameters (gts.dist.broadcast_object)")


def broadcast_parameters(params, src=0):
    """Identical replicas: rank `src`'s values win.  Checks the layouts first."""
    params = list(params)
    assert_same_layout(params)
    for p in params:
        dist.broadcast(p.data, src=src)


class FlatGradSync:
    """Packs every parameter gradient plus the two loss scalars into ONE flat fp32 buffer per
    step, all-reduces it, normalises, and hands the parameters views into it as their .grad.

    Gradients are left to autograd as separate tensors (`.grad = None` before backward, so no
    accumulate kernel runs per parameter); one concatenation kernel builds the flat buffer."""

    def __init__(self, params, group=None):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        for p in self.params:
            if p.dtype != torch.float32 or p.device != dev:
                raise ValueError("FlatGradSync needs fp32 parameters on one device")
        self.n_grad = sum(p.numel() for p in self.params)
        self.group = group
        self.flat = None
        self._scalars = None
        self._sink = None                # gts.nn.GradSink over the same parameters (+ 2 floats), made on first use

    def flat_gradients(self):
        """The normalised gradients of the last step as one contiguous fp32 tensor (parameter
        order), for optimizers that consume a flat buffer (gts.optim.FlatAdamW)."""
        if self.flat is None:
            raise RuntimeError("call all_reduce_and_normalise() first")
        return self.flat[:self.n_grad]

    def zero_grad(self):
        """Replaces optimizer.zero_grad(): drops the gradients (nothing is launched)."""
        for p in self.params:
            p.grad = None

    def weighted_ce_backward(self, logits, labels, class_weights, cross_entropy_sum=None):
        """Back-propagate the local numerator and keep (denominator, numerator) for the exchange.
        The loss is the fused HIP pass (numerator with gradient + denominator at once); there is no
        CPU path here — the gloo tests, which drive this class with the oracle network on the CPU,
        inject theirs through `cross_entropy_sum(logits, labels, class_weights) -> (num, den)`."""
        if cross_entropy_sum is not None:
            num, den = cross_entropy_sum(logits, labels, class_weights)
            num.backward()
            self._scalars = (den.detach().reshape(1), num.detach().reshape(1))
            return
        from . import nn, ops
        grad, stats = ops.weighted_ce_numerator_grad(logits, labels, class_weights)
        if self._sink is None:
            self._sink = nn.GradSink(self.params, extra=2)
        self._sink.new_buffer()
        with nn.grad_sink(self._sink):                          # a fused layer stack writes its gradients into the flat buffer
            logits.backward(grad)                               # d(numerator): the kernel's gradient as it is
        self._scalars = (stats[1:2], stats[0:1])                # (denominator, numerator): views, no launch

    def empty_step(self, device=None):
        """This rank has no sample in the (short, last) global batch of the epoch: it contributes
        zero gradients, numerator and denominator, and still takes part in the collective."""
        for p in self.params:
            p.grad = None
        if self._sink is not None:
            self._sink.filled = False
        zeros = torch.zeros(2, dtype=torch.float32, device=device or self.params[0].device)
        self._scalars = (zeros[0:1], zeros[1:2])

    def all_reduce_and_normalise(self):
        """One concatenation, one collective, one division; afterwards every rank holds the exact
        global-batch gradient in `p.grad` (views of the flat buffer).  Returns the global
        weighted-mean loss (0-dim tensor, no host sync)."""
        if self._scalars is None:
            raise RuntimeError("call weighted_ce_backward() first")
        if self._sink is not None and self._sink.filled:      # gradients are already in place: add the two scalars
            self.flat = self._sink.flat
            self.flat[self.n_grad:self.n_grad + 1].copy_(self._scalars[0])
            self.flat[self.n_grad + 1:].copy_(self._scalars[1])
            self._sink.filled = False
        else:
            pieces = [(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in self.params]
            self.flat = torch.cat(pieces + [t.to(pieces[0].dtype) for t in self._scalars])
        self._scalars = None
        _, w = world()
        if w > 1:
            reduce = lambda: dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)  # noqa: E731
            if COLLECTIVE_TIMER is not None:
                COLLECTIVE_TIMER(reduce)
            else:
                reduce()
        den = self.flat[self.n_grad]
        loss = self.flat[self.n_grad + 1] / den
        self.flat[:self.n_grad].div_(den)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        return loss
