"""Optimiser step of the reference's recipe on the MI355X path: ``clip_grad_norm_`` (run.py:125) and
``torch.optim.Adam(lr, weight_decay)`` (cnn14rnn_trm.yaml:42-46; the decay is L2 added to the gradient, not AdamW).

``FusedAdam`` is a ``torch.optim.Optimizer`` (same param_groups / ``state`` / ``state_dict`` layout as torch's Adam:
``step``, ``exp_avg``, ``exp_avg_sq`` per parameter, so the reference's LR schedulers and checkpointing work on it).
When parameters and gradients are the views of the flat buffers that ``audiocaption_amd.train.TrainEngine`` creates,
the whole update is ONE launch over the flat range (csrc/train.hip adam_kernel), otherwise one launch per tensor.
``clip_grad_norm_`` leaves the norm on the device (no host synchronisation): sum of squares, clip coefficient and
scaling are three launches, and ``FusedAdam.step(clip=...)`` folds the scaling into the update.
"""
import torch

from . import _lib
from ._lib import check, stream


NORM_STATE_FLOATS = 1032   # include/audiocaption_hip.h: 4 results + the scratch of the deterministic sum of squares


def _flat_span(tensors):
    """(storage_tensor_ptr, first_byte, n_floats) if the tensors are fp32 views of ONE storage, else None."""
    if not tensors:
        return None
    st = tensors[0].untyped_storage().data_ptr()
    lo, hi = None, None
    for t in tensors:
        if t.dtype != torch.float32 or not t.is_contiguous() or t.untyped_storage().data_ptr() != st:
            return None
        a, b = t.data_ptr(), t.data_ptr() + 4 * t.numel()
        lo = a if lo is None else min(lo, a)
        hi = b if hi is None else max(hi, b)
    return lo, (hi - lo) // 4


class GradClip:
    """Device-side result of ``clip_grad_norm_``: ``state`` = [sum of squares, total norm, coefficient]."""

    def __init__(self, state):
        self.state = state

    @property
    def total_norm(self):
        return self.state[1]


def clip_grad_norm_(parameters, max_norm, grad_div=1.0, scale_now=True):
    """Total L2 norm of the gradients (of ``grad / grad_div``: pass the world size when the gradients hold an
    all-reduce SUM) and, with ``scale_now``, the in-place scaling by min(1, max_norm / (norm + 1e-6)) / grad_div
    (torch.nn.utils.clip_grad_norm_, run.py:125).  Returns a ``GradClip``; ``.total_norm`` is a device scalar."""
    lib = _lib.load()
    if isinstance(parameters, torch.Tensor):
        parameters = [parameters]
    grads = [p.grad for p in parameters if p.grad is not None]
    if not grads:
        raise ValueError("no gradients to clip")
    dev = grads[0].device
    state = torch.zeros(NORM_STATE_FLOATS, device=dev, dtype=torch.float32)
    s = stream()
    span = _flat_span(grads)
    if span is not None:
        check(lib.ac_grad_sumsq(span[0], span[1], state.data_ptr(), s), "ac_grad_sumsq")
    else:
        for g in grads:
            g = g if g.is_contiguous() else g.contiguous()
            check(lib.ac_grad_sumsq(g.data_ptr(), g.numel(), state.data_ptr(), s), "ac_grad_sumsq")
    check(lib.ac_clip_coef(state.data_ptr(), float(max_norm), float(grad_div), s), "ac_clip_coef")
    if scale_now:
        if span is not None:
            check(lib.ac_scale_by_coef(span[0], span[1], state.data_ptr(), s), "ac_scale_by_coef")
        else:
            for g in grads:
                check(lib.ac_scale_by_coef(g.data_ptr(), g.numel(), state.data_ptr(), s), "ac_scale_by_coef")
    return GradClip(state)


class FusedAdam(torch.optim.Optimizer):

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._flat_state = {}
        # Per group, the number of updates APPLIED so far lives on the device: a step whose gradient is not finite is
        # skipped by the kernels (run.py:123) and must not advance Adam's bias correction either - the reference does
        # not call optimizer.step() at all in that case.  ``state[p]["step"]`` is refreshed from it on ``state_dict()``.
        self._step_dev = {}

    def _group_step_dev(self, gi, params):
        dev = params[0].device
        t = self._step_dev.get(gi)
        if t is None or t.device != dev:
            done = max([int(self.state[p].get("step", 0)) for p in params] + [0])
            t = self._step_dev[gi] = torch.full((1,), done, device=dev, dtype=torch.int32)
        return t

    def _group_flat(self, gi, params):
        """Flat moment buffers for a group whose parameters (and gradients) are views of one storage each."""
        pspan = _flat_span([p.data for p in params])
        gspan = _flat_span([p.grad for p in params])
        if pspan is None or gspan is None or pspan[1] != gspan[1]:
            return None
        # same layout in both storages?
        if any((p.data_ptr() - pspan[0]) != (p.grad.data_ptr() - gspan[0]) for p in params):
            return None
        fs = self._flat_state.get(gi)
        if fs is None or fs["pspan"] != pspan:
            dev = params[0].device
            m = torch.zeros(pspan[1], device=dev, dtype=torch.float32)
            v = torch.zeros(pspan[1], device=dev, dtype=torch.float32)
            for p in params:  # adopt moments that already exist (e.g. loaded from a checkpoint)
                o = (p.data_ptr() - pspan[0]) // 4
                st = self.state[p]
                if "exp_avg" in st:
                    m[o:o + p.numel()].copy_(st["exp_avg"].reshape(-1))
                    v[o:o + p.numel()].copy_(st["exp_avg_sq"].reshape(-1))
                st.setdefault("step", 0)
                st["exp_avg"] = m[o:o + p.numel()].view(p.shape)
                st["exp_avg_sq"] = v[o:o + p.numel()].view(p.shape)
            fs = self._flat_state[gi] = {"pspan": pspan, "m": m, "v": v}
        fs["gspan"] = gspan
        return fs

    def sync_step_counts(self):
        """Copy the device-side counts of applied updates into ``state[p]["step"]`` (one small device->host read per
        group).  Called by ``state_dict()``; not needed for stepping."""
        for gi, group in enumerate(self.param_groups):
            t = self._step_dev.get(gi)
            if t is None:
                continue
            n = int(t.item())
            for p in group["params"]:
                if p in self.state:
                    self.state[p]["step"] = n

    def state_dict(self):
        self.sync_step_counts()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        # the loaded moments and step counts replace whatever flat buffers / device counters existed: they are re-adopted
        # from ``self.state`` by the next step
        self._flat_state = {}
        self._step_dev = {}

    @torch.no_grad()
    def step(self, closure=None, clip=None):
        """One Adam update.  ``clip``: a ``GradClip`` from ``clip_grad_norm_(..., scale_now=False)`` whose coefficient
        is applied to the gradients inside the update (and whose non-finite flag skips it)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        s = stream()
        coef = clip.state.data_ptr() if clip is not None else None
        touched = []
        for gi, group in enumerate(self.param_groups):
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            b1, b2 = group["betas"]
            fs = self._group_flat(gi, params)
            step_dev = self._group_step_dev(gi, params)
            touched += params
            if fs is not None:
                check(lib.ac_adam_step(fs["pspan"][0], fs["gspan"][0], fs["m"].data_ptr(), fs["v"].data_ptr(),
                                       fs["pspan"][1], coef, float(group["lr"]), b1, b2, group["eps"],
                                       group["weight_decay"], 0, step_dev.data_ptr(), s), "ac_adam_step")
            else:
                for p in params:
                    st = self.state[p]
                    if "exp_avg" not in st:
                        st["step"] = 0
                        st["exp_avg"] = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.contiguous_format)
                        st["exp_avg_sq"] = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.contiguous_format)
                    if p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
                        raise NotImplementedError("FusedAdam: contiguous fp32 parameters and gradients only")
                    check(lib.ac_adam_step(p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(),
                                           st["exp_avg_sq"].data_ptr(), p.numel(), coef, float(group["lr"]), b1, b2,
                                           group["eps"], group["weight_decay"], 0, step_dev.data_ptr(), s), "ac_adam_step")
            check(lib.ac_adam_commit(step_dev.data_ptr(), coef, s), "ac_adam_commit")
        # packed inference weights of exactly these parameters must be rebuilt
        _lib.bump_param_generation(touched)
        return loss
