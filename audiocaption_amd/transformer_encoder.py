"""Transformer temporal encoder, MI355X path.  Plugin-compatible with the reference class
``captioning.models.transformer_encoder.TransformerEncoder`` (transformer_encoder.py:64-116): same constructor
keywords, ``forward({"attn", "attn_len"}) -> {"attn_emb", "fc_emb", "attn_emb_len"}``, parameters under
``attn_proj.*``, ``model.layers.N.*`` (nn.TransformerEncoderLayer names) and ``cls_token``.

The nn modules only own the parameters.  The forward runs attn_proj (Linear -> ReLU -> LayerNorm), prepends the learned
cls token and applies the post-LN encoder layers (self-attention with the key-padding mask of ``attn_len + 1``, ReLU
feed-forward) with the kernels of csrc/train.hip: ``ac_gemm``, ``ac_attn_seq_fwd``, ``ac_dropadd_ln_fwd``.
Like the reference (transformer_encoder.py:105) the length tensor is incremented IN PLACE.
"""
import torch
import torch.nn as nn

from . import _lib
from . import kernels as K
from ._lib import check, f32c, ptr, stream


class TransformerEncoder(nn.Module):

    def __init__(self, spec_dim, fc_feat_dim, attn_feat_dim, d_model, **kwargs):
        super().__init__()
        self.spec_dim, self.fc_feat_dim, self.attn_feat_dim = spec_dim, fc_feat_dim, attn_feat_dim
        self.d_model = d_model
        dropout = kwargs.get("dropout", 0.2)
        self.nhead = kwargs.get("nhead", self.d_model // 64)
        self.nlayers = kwargs.get("nlayers", 2)
        self.dim_feedforward = kwargs.get("dim_feedforward", self.d_model * 4)
        if d_model != 256 or self.nhead * 64 != d_model:
            raise NotImplementedError("TransformerEncoder (HIP path): d_model 256 with 64-wide heads only")
        self.attn_proj = nn.Sequential(nn.Linear(attn_feat_dim, d_model), nn.ReLU(), nn.Dropout(dropout),
                                       nn.LayerNorm(d_model))
        layer = nn.TransformerEncoderLayer(d_model=d_model, nhead=self.nhead, dim_feedforward=self.dim_feedforward,
                                           dropout=dropout)
        self.model = nn.TransformerEncoder(layer, self.nlayers, enable_nested_tensor=False)
        self.cls_token = nn.Parameter(torch.zeros(d_model))
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def forward(self, input_dict):
        if self.training:
            raise NotImplementedError("TransformerEncoder (HIP path): inference only (no backward is built)")
        lib = _lib.load()
        attn = f32c(input_dict["attn"])
        if not attn.is_cuda:
            raise _lib.HipLibraryError("the HIP path needs tensors on a ROCm device; there is no CPU fallback")
        lens = torch.as_tensor(input_dict["attn_len"])
        N, T, A = attn.shape
        d, F, nh = self.d_model, self.dim_feedforward, self.nhead
        dev = attn.device
        s = stream()
        f32 = dict(device=dev, dtype=torch.float32)

        def lin(x, w, b, y, M, Nn, Kd, relu=0):
            check(lib.ac_gemm(ptr(x), Kd, 1, ptr(w), 1, Kd, ptr(y), Nn, M, Nn, Kd, ptr(b), relu, 0.0, 1, 0.0, 0, None, 0,
                              None, 0, s), "ac_gemm")

        def add_ln(x, res, ln, y, rows):
            pre = torch.empty(rows, d, **f32)
            check(lib.ac_dropadd_ln_fwd(ptr(x), ptr(res), ptr(ln.weight), ptr(ln.bias), ptr(pre), ptr(y), 0, rows, 0, d,
                                        0.0, 0, None, float(ln.eps), s), "ac_dropadd_ln_fwd")

        a = torch.empty(N * T, d, **f32)
        lin(attn, f32c(self.attn_proj[0].weight), f32c(self.attn_proj[0].bias), a, N * T, d, A, relu=1)
        proj = torch.empty(N * T, d, **f32)
        add_ln(a, None, self.attn_proj[3], proj, N * T)
        L = T + 1
        x = torch.empty(N, L, d, **f32)
        x[:, 0] = self.cls_token.detach().float()
        x[:, 1:] = proj.view(N, T, d)
        lens += 1                                             # in place, as transformer_encoder.py:105
        rows = N * L
        row0 = (torch.arange(N, dtype=torch.int32) * L).to(dev)
        qlen = torch.full((N,), L, dtype=torch.int32, device=dev)
        valid = K.upload(lens, dev, torch.int32)
        qkv = torch.empty(rows, 3 * d, **f32)
        ctx = torch.empty(rows, d, **f32)
        sub = torch.empty(rows, d, **f32)
        hid = torch.empty(rows, F, **f32)
        P = torch.empty(N * nh * L * L, **f32)
        for layer in self.model.layers:
            sa = layer.self_attn
            lin(x, f32c(sa.in_proj_weight), f32c(sa.in_proj_bias), qkv, rows, 3 * d, d)
            q = qkv.data_ptr()
            check(lib.ac_attn_seq_fwd(q, 3 * d, q + 4 * d, 3 * d, q + 8 * d, 3 * d, ptr(ctx), d, ptr(P), L, L, ptr(row0),
                                      ptr(qlen), ptr(row0), ptr(qlen), ptr(valid), None, 0, 0, 0, N, nh, 64, L, L, 0.0,
                                      0, None, s), "ac_attn_seq_fwd")
            lin(ctx, f32c(sa.out_proj.weight), f32c(sa.out_proj.bias), sub, rows, d, d)
            y = torch.empty(N, L, d, **f32)
            add_ln(sub, x, layer.norm1, y, rows)
            lin(y, f32c(layer.linear1.weight), f32c(layer.linear1.bias), hid, rows, F, d, relu=1)
            lin(hid, f32c(layer.linear2.weight), f32c(layer.linear2.bias), sub, rows, d, F)
            x = torch.empty(N, L, d, **f32)
            add_ln(sub, y, layer.norm2, x, rows)
        return {"attn_emb": x, "fc_emb": x[:, 0], "attn_emb_len": lens}
