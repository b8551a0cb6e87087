"""CNN -> RNN composite encoder.  Plugin-compatible with the reference class
``captioning.models.crnn_trm_encoder.CrnnEncoder`` (crnn_trm_encoder.py:179-211).

``Cnn14RnnEncoder`` is accepted as an alias because the reference's own Clotho config names it
(eg_configs/clotho_v2/waveform/cnn14rnn_trm.yaml:9) although that class does not exist in the
reference module (SURVEY.md "Key facts").
"""
import torch.nn as nn


class CrnnEncoder(nn.Module):

    def __init__(self, cnn, rnn, freeze_cnn=False, freeze_cnn_bn=False, **kwargs):
        super().__init__()
        self.cnn = cnn
        self.rnn = rnn
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
        # Cnn14's own fc_emb is dead in this pipeline (the RNN recomputes it): skip its kernels.
        try:
            out = self.cnn(input_dict, skip_fc=True)
        except TypeError:
            out = self.cnn(input_dict)
        return self.rnn({"attn": out["attn_emb"], "attn_len": out["attn_emb_len"]})


Cnn14RnnEncoder = CrnnEncoder


class Cnn14TransformerEncoder(nn.Module):
    """CNN -> Transformer composite encoder (reference crnn_trm_encoder.py:214-246)."""

    def __init__(self, cnn, transformer, freeze_cnn=False, freeze_cnn_bn=False, **kwargs):
        super().__init__()
        self.cnn = cnn
        self.trm = transformer
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
        try:
            out = self.cnn(input_dict, skip_fc=True)
        except TypeError:
            out = self.cnn(input_dict)
        return self.trm({"attn": out["attn_emb"], "attn_len": out["attn_emb_len"]})
