"""AdamW as one HIP launch over a flat parameter buffer.

Same update rule and constructor meaning as `torch.optim.AdamW(params, lr, weight_decay)`
(the reference's optimizer, model/gnn_model.py:28): a `torch.optim.Optimizer` with ONE param
group, so `lr_scheduler.ExponentialLR` (gnn_model.py:29) drives it unchanged.  The network's
1.25 M parameters live in 32 tensors; at construction they are re-pointed at consecutive
slices of one flat fp32 buffer (values preserved, `state_dict` / `load_state_dict` unaffected)
and every step is a single elementwise pass (gts_adamw_f32) instead of multi-tensor launches
that keep a fifth of the chip busy.
"""
import torch

from . import _lib
from ._lib import check, current_stream, ptr


class FlatAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        params = [p for p in params if p.requires_grad]
        if not params:
            raise ValueError("optimizer got an empty parameter list")
        dev = params[0].device
        for p in params:
            if p.dtype != torch.float32 or p.device != dev or not p.is_cuda:
                raise _lib.GtsError("FlatAdamW needs fp32 parameters on one AMD GPU (no CPU fallback)")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._params = params
        self._sizes = [p.numel() for p in params]
        self.n = sum(self._sizes)
        self.flat_param = torch.cat([p.detach().reshape(-1) for p in params])
        off = 0
        for p, n in zip(params, self._sizes):           # parameters become views of the flat buffer
            p.data = self.flat_param[off:off + n].view_as(p)
            off += n
        self.exp_avg = torch.zeros_like(self.flat_param)
        self.exp_avg_sq = torch.zeros_like(self.flat_param)
        self.steps = 0

    def _still_flat(self):
        """Every parameter is still the slice of the flat buffer it was made (re-allocations such as net.to(...) would
        break that).  `Tensor._version`-free and cheap: storage addresses only."""
        base, off = self.flat_param.data_ptr(), 0
        for p, n in zip(self._params, self._sizes):
            if p.data_ptr() != base + 4 * off:
                return False
            off += n
        return True

    @torch.no_grad()
    def step(self, closure=None, flat_grad=None):
        """One update.  `flat_grad` (optional): the gradients already laid out like the
        parameters in one contiguous fp32 tensor (gts.dist.FlatGradSync.flat_gradients());
        otherwise the `.grad` tensors are concatenated by one kernel."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if not self._still_flat():
            raise _lib.GtsError("parameters were re-allocated after FlatAdamW was built (e.g. net.to(...)): "
                                "create the optimizer after moving the network")
        if flat_grad is None:
            missing = [i for i, p in enumerate(self._params) if p.grad is None]
            if missing:
                raise _lib.GtsError(f"parameters {missing} received no gradient this step")
            flat_grad = torch.cat([p.grad.reshape(-1) for p in self._params])
        if flat_grad.numel() != self.n or flat_grad.dtype != torch.float32 or not flat_grad.is_contiguous():
            raise _lib.GtsError("flat_grad must be one contiguous fp32 tensor with one entry per parameter")
        group = self.param_groups[0]
        self.steps += 1
        check(_lib.load().gts_adamw_f32(ptr(self.flat_param), ptr(flat_grad), ptr(self.exp_avg),
                                        ptr(self.exp_avg_sq), self.n, float(group["lr"]),
                                        float(group["betas"][0]), float(group["betas"][1]), float(group["eps"]),
                                        float(group["weight_decay"]), self.steps, current_stream()),
              "gts_adamw_f32")
        return loss
