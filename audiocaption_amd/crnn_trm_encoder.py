"""CNN -> RNN composite encoder.  Plugin-compatible with the reference class
``captioning.models.crnn_trm_encoder.CrnnEncoder`` (crnn_trm_encoder.py:179-211).

``Cnn14RnnEncoder`` is accepted as an alias because the reference's own Clotho config names it
(eg_configs/clotho_v2/waveform/cnn14rnn_trm.yaml:9) although that class does not exist in the
reference module (SURVEY.md "Key facts").
"""
import inspect

import torch.nn as nn


def _accepts_skip_fc(cnn):
    """Whether the CNN's forward takes ``skip_fc`` (this package's Cnn14Encoder does; any other plugin encoder is
    called with the reference's plain ``forward(input_dict)``).  Decided once from the signature: a ``TypeError``
    raised INSIDE the encoder must surface, not trigger a silent second run."""
    try:
        return "skip_fc" in inspect.signature(cnn.forward).parameters
    except (TypeError, ValueError):
        return False


def _run_cnn(cnn, skip_fc, input_dict):
    out = cnn(input_dict, skip_fc=True) if skip_fc else cnn(input_dict)
    res = {"attn": out["attn_emb"], "attn_len": out["attn_emb_len"]}
    if "gru_algo" in input_dict:
        res["gru_algo"] = input_dict["gru_algo"]
    return res, out.get("f16_overflow")


class CrnnEncoder(nn.Module):

    def __init__(self, cnn, rnn, freeze_cnn=False, freeze_cnn_bn=False, **kwargs):
        super().__init__()
        self.cnn = cnn
        self.rnn = rnn
        self._skip_fc = _accepts_skip_fc(cnn)
        self.freeze_cnn_bn = False
        if freeze_cnn:
            for param in self.cnn.parameters():
                param.requires_grad = False
            self.freeze_cnn_bn = freeze_cnn_bn

    def train(self, mode=True):
        super().train(mode=mode)
        if self.freeze_cnn_bn:
            for module in self.cnn.modules():
                if module.__class__.__name__.find("BatchNorm") != -1:
                    module.eval()
        return self

    def forward(self, input_dict):
        return self.forward_back(self.forward_front(input_dict))

    # The two halves of forward(), so that a caller can put them on different streams (TransformerModel.forward_async:
    # the matrix-bound convolutions of the NEXT batch overlap the latency-bound recurrence of this one).
    def forward_front(self, input_dict):
        # Cnn14's own fc_emb is dead in this pipeline (the RNN recomputes it): skip its kernels.
        feats, overflow = _run_cnn(self.cnn, self._skip_fc, input_dict)
        return {"feats": feats, "overflow": overflow}

    def forward_back(self, front):
        out = self.rnn(front["feats"])
        if front["overflow"] is not None:
            out["f16_overflow"] = front["overflow"]   # fp16 range flag of the conv tier (see Cnn14Encoder.forward)
        return out


Cnn14RnnEncoder = CrnnEncoder


class Cnn14TransformerEncoder(nn.Module):
    """CNN -> Transformer composite encoder (reference crnn_trm_encoder.py:214-246)."""

    def __init__(self, cnn, transformer, freeze_cnn=False, freeze_cnn_bn=False, **kwargs):
        super().__init__()
        self.cnn = cnn
        self.trm = transformer
        self._skip_fc = _accepts_skip_fc(cnn)
        self.freeze_cnn_bn = False
        if freeze_cnn:
            for param in self.cnn.parameters():
                param.requires_grad = False
            self.freeze_cnn_bn = freeze_cnn_bn

    def train(self, mode=True):
        super().train(mode=mode)
        if self.freeze_cnn_bn:
            for module in self.cnn.modules():
                if module.__class__.__name__.find("BatchNorm") != -1:
                    module.eval()
        return self

    def forward(self, input_dict):
        feats, overflow = _run_cnn(self.cnn, self._skip_fc, input_dict)
        out = self.trm(feats)
        if overflow is not None:
            out["f16_overflow"] = overflow
        return out
