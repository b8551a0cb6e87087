"""Trainer state machines around the training step: the scheduled-sampling ratio
(run.py:55-65), stochastic weight averaging (``AveragedModel``, train_util.py:233-253; run.py:303-305,350-355) and the
NaN-loss skip (run.py:123, done on the device by csrc/train.hip's clip / Adam kernels).  Host-side arithmetic except
the SWA update, which is one launch over the flat parameter buffer."""
import torch

from . import _lib
from ._lib import check, stream


class ScheduledSampling:
    """``Runner._update_ss_ratio`` (run.py:55-65): called once per iteration BEFORE the forward; ``ratio`` starts at 1
    and decays exponentially to 0.01 or linearly to ``final_ratio`` over ``total_iters`` iterations."""

    def __init__(self, use=True, mode="linear", final_ratio=0.7, total_iters=1):
        if mode not in ("exponential", "linear"):
            raise Exception(f"mode {mode} not supported")
        self.use, self.mode, self.final_ratio, self.total_iters = use, mode, final_ratio, total_iters
        self.ratio = 1.0

    def step(self):
        if not self.use:
            return self.ratio
        if self.mode == "exponential":
            self.ratio *= 0.01 ** (1.0 / self.total_iters)
        else:
            self.ratio -= (1.0 - self.final_ratio) / self.total_iters
        return self.ratio


class SwaAverager:
    """Running average of a model's parameters and buffers.  ``update_parameters(model)`` follows
    ``AveragedModel.update_parameters`` with torch's default ``avg_fn`` (avg + (p - avg) / (n + 1)); ``state_dict()``
    returns tensors under the model's own keys (what run.py:350-355 saves as ``swa.pth``).  When the parameters are the
    views of a TrainEngine's flat buffer the whole update is one launch."""

    def __init__(self, model):
        self.keys = [k for k, _ in model.named_parameters()] + [k for k, _ in model.named_buffers()]
        self.n_averaged = 0
        self.avg = None
        self._flat_src = None

    def _tensors(self, model):
        return [p.detach() for _, p in model.named_parameters()] + [b.detach() for _, b in model.named_buffers()]

    @torch.no_grad()
    def update_parameters(self, model):
        lib = _lib.load()
        tensors = self._tensors(model)
        if self.avg is None:
            self.avg = [t.clone() for t in tensors]
            self.n_averaged = 1
            return
        engine = getattr(model, "_train_engine", None)
        flat_ids = set()
        if engine is not None and engine.flat is not None and engine.flat.intact():
            fp = engine.flat
            if self._flat_src is None:   # re-point the averaged copies of the flat parameters into one buffer as well
                index = {id(p): i for i, p in enumerate(model.parameters())}
                self._flat_avg = torch.zeros_like(fp.flat)
                for p, off in zip(fp.params, fp.offsets):
                    i = index[id(p)]
                    view = self._flat_avg[off:off + p.numel()].view(p.shape)
                    view.copy_(self.avg[i])
                    self.avg[i] = view
                self._flat_src = fp
            check(lib.ac_swa_update(self._flat_avg.data_ptr(), fp.flat.data_ptr(), fp.total, self.n_averaged, stream()),
                  "ac_swa_update")
            flat_ids = {id(p) for p in fp.params}
        for a, t, src in zip(self.avg, tensors, list(model.parameters()) + list(model.buffers())):
            if id(src) in flat_ids:
                continue
            if not t.dtype.is_floating_point:
                a.copy_(t)                                   # num_batches_tracked and friends
            elif t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and a.is_contiguous():
                check(lib.ac_swa_update(a.data_ptr(), t.data_ptr(), t.numel(), self.n_averaged, stream()), "ac_swa_update")
            else:
                a.add_((t - a) / (self.n_averaged + 1))      # CPU tensors: host arithmetic
        self.n_averaged += 1

    def state_dict(self):
        return {k: a.clone() for k, a in zip(self.keys, self.avg)}


def with_next(batches):
    """Iterate ``(batch, next_batch)`` over a data loader (``next_batch`` is None behind the last one): what
    ``TrainEngine.step(batch, optimizer, next_batch=next_batch)`` wants in order to run the frozen Cnn14 forward of the
    following iteration on a side stream under the current one (train.py ``prefetch_cnn``; run.py:77-148 hands its batches
    over one at a time)."""
    it = iter(batches)
    try:
        cur = next(it)
    except StopIteration:
        return
    for nxt in it:
        yield cur, nxt
        cur = nxt
    yield cur, None
