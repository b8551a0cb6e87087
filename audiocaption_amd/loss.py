"""Label-smoothing cross entropy on the MI355X path.  Plugin-compatible with the reference class
``captioning.losses.loss.LabelSmoothingLoss`` (loss.py:40-74): same constructor, ``forward(output_dict)`` reading
``logit`` (N, T, V), ``tgt`` (N, T) and ``tgt_len`` (N,), reductions "mean" / "sum" / "none".

Forward and backward are the one-pass kernel of csrc/train.hip (log-sum-exp, smoothed target term and
``softmax - q`` per row); there is no PyTorch fallback.
"""
import torch
import torch.nn as nn

from . import _lib
from ._lib import check, ptr, stream


def _launch(logit, tgt, tgt_len_dev, smoothing, inv_count, dlogit, gscale, gscale_dev):
    lib = _lib.load()
    N, T, V = logit.shape
    row_loss = torch.empty(N * T, device=logit.device, dtype=torch.float32)
    loss = torch.empty(1, device=logit.device, dtype=torch.float32)
    check(lib.ac_label_smoothing_loss(ptr(logit), ptr(tgt), tgt.stride(0), ptr(tgt_len_dev), N, T, V, float(smoothing),
                                      float(inv_count), ptr(row_loss), ptr(loss), ptr(dlogit), float(gscale),
                                      ptr(gscale_dev), stream()), "ac_label_smoothing_loss")
    return loss, row_loss


class _LabelSmoothingFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, logit, tgt, tgt_len_dev, smoothing, inv_count, reduction):
        loss, row_loss = _launch(logit, tgt, tgt_len_dev, smoothing, inv_count, None, 0.0, None)
        ctx.save_for_backward(logit, tgt, tgt_len_dev)
        ctx.args = (smoothing, inv_count, reduction)
        if reduction == "none":
            return row_loss.view(logit.shape[0], logit.shape[1])
        return loss[0]

    @staticmethod
    def backward(ctx, grad_out):
        logit, tgt, tgt_len_dev = ctx.saved_tensors
        smoothing, inv_count, reduction = ctx.args
        if reduction == "none":
            raise NotImplementedError("LabelSmoothingLoss (HIP path): backward of reduction='none' is not built")
        dlogit = torch.empty_like(logit)
        g = grad_out.reshape(1).to(device=logit.device, dtype=torch.float32)
        _launch(logit, tgt, tgt_len_dev, smoothing, inv_count, dlogit, inv_count, g)
        return dlogit, None, None, None, None, None


class LabelSmoothingLoss(nn.Module):

    def __init__(self, smoothing=0.0, dim=-1, reduction="mean", logit_name="logit", target_name="tgt"):
        super().__init__()
        self.confidence = 1.0 - smoothing
        self.smoothing = smoothing
        if dim not in (-1, 2):
            raise NotImplementedError("LabelSmoothingLoss (HIP path): the class axis must be the last one")
        self.dim = dim
        self.reduction = reduction
        self.logit_name = logit_name
        self.target_name = target_name

    def forward(self, output):
        logit = output[self.logit_name]
        tgt = output[self.target_name]
        tgt_len = torch.as_tensor(output[f"{self.target_name}_len"])
        if logit.dim() != 3:
            raise ValueError("logit must be (batch, length, classes)")
        if logit.dtype != torch.float32 or not logit.is_contiguous():
            logit = logit.float().contiguous()
        dev = logit.device
        T = logit.shape[1]
        tgt = tgt.to(device=dev, dtype=torch.int64)
        if tgt.stride(1) != 1:
            tgt = tgt.contiguous()
        # generate_length_mask(tgt_len) (model_util.py:29-38) has max(tgt_len) columns: every row below T counts
        count = float(torch.clamp(tgt_len.cpu(), max=T).sum())
        inv_count = {"mean": 1.0 / count, "sum": 1.0, "none": 1.0}[self.reduction]
        tgt_len_dev = tgt_len.to(device=dev, dtype=torch.int32)
        return _LabelSmoothingFn.apply(logit, tgt, tgt_len_dev, self.smoothing, inv_count, self.reduction)
