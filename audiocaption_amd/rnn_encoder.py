"""Bidirectional-GRU temporal encoder, MI355X path.  Plugin-compatible with the reference class
``captioning.models.rnn_encoder.RnnEncoder`` (rnn_encoder.py:10-49): same constructor keywords,
``forward({"attn", "attn_len"}) -> {"attn_emb", "fc_emb", "attn_emb_len"}``, parameters under
``network.*`` with nn.GRU's names and gate order.

``self.network`` (an nn.GRU) only owns the parameters; the forward pass is one MFMA GEMM per layer
for the input projections of all time steps (csrc/gemm.hip) plus the persistent recurrence kernel of
csrc/gru.hip, which reproduces pack_padded_sequence semantics (model_util.py:10-27) from the lengths.
"""
import os

import torch
import torch.nn as nn

from . import kernels as K


class RnnEncoder(nn.Module):

    def __init__(self, spec_dim, fc_feat_dim, attn_feat_dim, pooling="mean", **kwargs):
        super().__init__()
        self.spec_dim, self.fc_feat_dim, self.attn_feat_dim = spec_dim, fc_feat_dim, attn_feat_dim
        self.pooling = pooling
        self.hidden_size = kwargs.get("hidden_size", 512)
        self.bidirectional = kwargs.get("bidirectional", False)
        self.num_layers = kwargs.get("num_layers", 1)
        self.dropout = kwargs.get("dropout", 0.2)
        self.rnn_type = kwargs.get("rnn_type", "GRU")
        self.in_bn = kwargs.get("in_bn", False)
        self.embed_dim = self.hidden_size * (self.bidirectional + 1)
        if self.rnn_type != "GRU" or not self.bidirectional or self.hidden_size != 256 or self.in_bn:
            raise NotImplementedError(
                "RnnEncoder (HIP path) is built for the configuration the reference configs use: "
                "rnn_type=GRU, bidirectional=True, hidden_size=256, in_bn=False "
                "(eg_configs/*/waveform/cnn14rnn_trm.yaml:18-27)")
        if self.pooling != "mean":
            raise NotImplementedError("RnnEncoder (HIP path): only pooling='mean' is built")
        self.network = nn.GRU(attn_feat_dim, self.hidden_size, num_layers=self.num_layers,
                              bidirectional=True, dropout=self.dropout, batch_first=True)
        self._packed = None
        self._packed_key = None
        # "split" (default): every (clip, direction) on four workgroups of 256 threads with W_hh register resident
        # (csrc/gru.hip gru_layer_split_kernel: 79 us per layer at 64 clips x 31 steps); "single": one 768-thread
        # workgroup per (clip, direction), 44 % of W_hh re-streamed per step (156 us).  They differ in summation order only
        # (~2e-6).  ``input_dict["gru_algo"]`` overrides it for one call.
        self.gru_algo = os.environ.get("AUDIOCAPTION_GRU_ALGO", "split")
        self._split_ws = None

    def _pack(self):
        ps = dict(self.network.named_parameters())
        key = tuple((t.data_ptr(), t._version) for t in ps.values()) + (K._lib.param_generation(),)
        if self._packed is not None and key == self._packed_key:
            return self._packed
        layers = []
        with torch.no_grad():
            for l in range(self.num_layers):
                f, r = f"_l{l}", f"_l{l}_reverse"
                w_ih = torch.cat([ps["weight_ih" + f], ps["weight_ih" + r]], 0).float().contiguous()
                b_ih = torch.cat([ps["bias_ih" + f], ps["bias_ih" + r]], 0).float().contiguous()
                whh = torch.stack([ps["weight_hh" + f], ps["weight_hh" + r]], 0).float().contiguous()
                whhT = K.gru_pack_whh(whh, self.hidden_size)   # [2][H/4][3H][4]
                bhh = torch.stack([ps["bias_hh" + f], ps["bias_hh" + r]], 0).float().contiguous()
                layers.append((w_ih, b_ih, whhT, bhh, whh))
        self._packed, self._packed_key = layers, key
        return layers

    def forward(self, input_dict):
        if self.training:
            raise NotImplementedError(
                "RnnEncoder (HIP path): in train mode the GRU only runs inside the whole-model training step "
                "(audiocaption_amd.train.TrainEngine / TransformerModel.forward with mode='train')")
        x = input_dict["attn"]
        lens = torch.as_tensor(input_dict["attn_len"]).cpu().long()
        B, T, _ = x.shape
        if int(lens.min()) < 1 or int(lens.max()) > T:
            raise ValueError("attn_len must lie in [1, attn.size(1)]")
        lens_dev = K.upload(lens, x.device, torch.int32)
        h = K.f32c(x).reshape(B * T, -1)
        algo = input_dict.get("gru_algo", self.gru_algo)
        split = algo == "split"
        for (w_ih, b_ih, whhT, bhh, whh) in self._pack():
            gx = K.linear(h, w_ih, b_ih)                       # (B*T, 2*3H): all steps, both directions
            if split:
                h, self._split_ws = K.gru_layer_split(gx, whh, bhh, lens_dev, B, T, self.hidden_size, self._split_ws)
                h = h.reshape(B * T, -1)
            else:
                h = K.gru_layer(gx, whhT, bhh, lens_dev, B, T, self.hidden_size).reshape(B * T, -1)
        out = h.reshape(B, T, self.embed_dim)
        t_out = int(lens.max())                                 # pad_packed_sequence truncates to max(len)
        if t_out < T:
            out = out[:, :t_out].contiguous()
        fc_emb = K.mean_with_lens(out, lens_dev)
        res = {"attn_emb": out, "fc_emb": fc_emb, "attn_emb_len": lens}
        if split:
            # sticky device word: non-zero if a workgroup's partner never started (the output is then invalid);
            # TransformerModel reads it where it synchronises anyway and raises
            res["gru_error"] = K.gru_split_error(self._split_ws, B)
        return res
