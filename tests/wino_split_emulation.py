"""Checker script (uses the oracle, hence under tests/): what would Winograd convolutions on SPLIT 16-bit operands cost in
accuracy?  CPU emulation on the oracle's conv stack: every 3x3 convolution after the first one is replaced by

    direct   : x (f32) * w, operands split hi + lo in <fmt>, products hi*hi + hi*lo + lo*hi     (today's "bf16x3" tier)
    wino2d   : F(2x2,3x3): V = B^T d B in f32 then split, U = G g G^T in f64 then split, 16 GEMMs of 3 products
    wino1d   : F(2,3) along the time axis only (4 positions x 3 mel taps), same splitting
    wino43   : F(4,3) along the time axis only (6 positions x 3 mel taps): the default tier's arithmetic
    nest43x23: F(4,3) along time x F(2,3) along mel (24 positions per 4 x 2 outputs: 3 products per output instead of 4.5) -
               the 2-D nest priced in DESIGN.md section 7 (round 6: accuracy is NOT what rules it out)
    nest43x43: F(4,3) x F(4,3) (36 positions per 16 outputs), for the trend

with f32 accumulation, and the greedy logits are compared with the plain fp32 oracle.

    python tests/wino_split_emulation.py [seconds=4] [max_blocks=6] [names, comma separated]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from audiocaption_amd import procedural as P
from oracle import cpu_path as O

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def split(x, fmt, flush=False):
    """x (f32 / f64) -> (hi, lo) as f32 tensors holding values representable in fmt."""
    x = x.float()
    dt = {"bf16": torch.bfloat16, "f16": torch.float16}[fmt]
    hi = x.to(dt).float()
    lo = (x - hi).to(dt).float()
    if flush and fmt == "f16":
        tiny = 2.0 ** -14
        hi = torch.where(hi.abs() < tiny, torch.zeros_like(hi), hi)
        lo = torch.where(lo.abs() < tiny, torch.zeros_like(lo), lo)
    return hi, lo


def mm3(a, b, fmt, flush):
    """sum_k a[..., m, k] b[..., k, n] with both operands split: 3 products, f32 accumulate."""
    ah, al = split(a, fmt, flush)
    bh, bl = split(b, fmt, flush)
    return ah @ bh + (ah @ bl + al @ bh)


def conv_direct(x, w, fmt, flush):
    B, C, H, W = x.shape
    cols = F.unfold(x, 3, padding=1)                     # (B, C*9, H*W)
    y = mm3(w.reshape(w.shape[0], -1)[None], cols, fmt, flush)
    return y.reshape(B, -1, H, W)


def conv_wino2d(x, w, fmt, flush):
    B, C, H, W = x.shape
    He, We = H + (H & 1), W + (W & 1)
    xp = F.pad(x, (1, 1 + We - W, 1, 1 + He - H))
    t = F.unfold(xp, 4, stride=2)                        # (B, C*16, nt)
    nt = t.shape[-1]
    d = t.reshape(B, C, 4, 4, nt)
    bt = BT.float()
    V = torch.einsum("ia,bcakn,jk->bijcn", bt, d, bt)    # (B, 4, 4, C, nt) f32 adds
    U = torch.einsum("ia,ocak,jk->ijoc", G, w.double(), G)   # (4, 4, O, C) f64
    M = mm3(U[None], V, fmt, flush)                      # (B, 4, 4, O, nt)
    at = AT.float()
    Y = torch.einsum("ai,bijon,cj->boacn", at, M, at)    # (B, O, 2, 2, nt)
    y = F.fold(Y.reshape(B, -1, nt), (He, We), 2, stride=2)
    return y[:, :, :H, :W]


def conv_wino1d(x, w, fmt, flush):
    """F(2,3) along H (time); the three W (mel) taps stay direct."""
    B, C, H, W = x.shape
    He = H + (H & 1)
    xp = F.pad(x, (1, 1, 1, 1 + He - H))                 # (B, C, He+2, W+2)
    bt = BT.float()
    # rows: tiles of 4 with stride 2
    rows = xp.unfold(2, 4, 2)                            # (B, C, nt, W+2, 4)
    V = torch.einsum("ia,bctwa->bitwc", bt, rows)        # (B, 4, nt, W+2, C)
    U = torch.einsum("ia,ocak->ikoc", G, w.double())     # (4 pos, 3 kx, O, C)
    nt = V.shape[2]
    M = 0
    for kx in range(3):
        Vk = V[:, :, :, kx:kx + W, :].reshape(B, 4, nt * W, C).transpose(2, 3)   # (B, 4, C, nt*W)
        M = M + mm3(U[None, :, kx], Vk, fmt, flush)      # (B, 4, O, nt*W)
    at = AT.float()
    Y = torch.einsum("ai,bion->boan", at, M)             # (B, O, 2, nt*W)
    Y = Y.reshape(B, -1, 2, nt, W).permute(0, 1, 3, 2, 4).reshape(B, -1, He, W)
    return Y[:, :, :H]


# F(4,3), points 0, +-1, +-2, inf (Lavin & Gray) - the matrices csrc/ac_wino43.h spells out row by row
BT4 = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                    [0, 4, 0, -5, 0, 1]], dtype=torch.float64)
G4 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
                   [0, 0, 1]], dtype=torch.float64)
AT4 = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)
# "F(1,3)": the direct form as a degenerate transform (three positions = the three taps)
BT1, G1, AT1 = torch.eye(3, dtype=torch.float64), torch.eye(3, dtype=torch.float64), torch.ones(1, 3, dtype=torch.float64)
XF = {1: (BT1, G1, AT1), 2: (BT, G, AT), 4: (BT4, G4, AT4)}


def conv_wino_nest(x, w, fmt, flush, mh, mw):
    """F(mh,3) along H (time) nested with F(mw,3) along W (mel); m = 1 is the direct form along that axis.  Input transform in
    f32 BEFORE the split, filter transform in f64 before the split, one split product per position, f32 accumulation."""
    B, C, H, W = x.shape
    (bth, gh, ath), (btw, gw, atw) = XF[mh], XF[mw]
    ah, aw = mh + 2, mw + 2
    He, We = -(-H // mh) * mh, -(-W // mw) * mw
    xp = F.pad(x, (1, 1 + We - W, 1, 1 + He - H))
    tiles = xp.unfold(2, ah, mh).unfold(3, aw, mw)       # (B, C, th, tw, ah, aw)
    th, tw = tiles.shape[2], tiles.shape[3]
    V = torch.einsum("ia,bctuak,jk->bijctu", bth.float(), tiles, btw.float()).reshape(B, ah, aw, C, th * tw)
    U = torch.einsum("ia,ocak,jk->ijoc", gh, w.double(), gw)   # (ah, aw, O, C) f64
    M = mm3(U[None], V, fmt, flush)                      # (B, ah, aw, O, nt)
    Y = torch.einsum("ai,bijon,cj->boacn", ath.float(), M, atw.float())   # (B, O, mh, mw, nt)
    Y = Y.reshape(B, -1, mh, mw, th, tw).permute(0, 1, 4, 2, 5, 3).reshape(B, -1, He, We)
    return Y[:, :, :H, :W]


def conv_stack(state, lms, conv, blocks, prefix="encoder.cnn."):
    x = lms.transpose(1, 2).unsqueeze(1)
    x = O._bn_eval(x.transpose(1, 3), state, prefix + "bn0").transpose(1, 3)
    for b in range(1, 7):
        p = f"{prefix}conv_block{b}."
        for j in (1, 2):
            w = state[p + f"conv{j}.weight"]
            if (b == 1 and j == 1) or b > blocks or conv is None:
                y = F.conv2d(x, w, padding=1)
            else:
                y = conv(x, w)
            x = F.relu(O._bn_eval(y, state, p + f"bn{j}"))
        if b < 6:
            x = F.avg_pool2d(x, 2)
    return x.mean(dim=3).transpose(1, 2)


def main():
    sec = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
    blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    vocab = 4368
    state = P.to_torch(P.cnn14rnn_trm_state(vocab))
    L = int(32000 * sec)
    lens = [L, int(L * 0.8)]
    wav = P.synthetic_wav(2, L, seed=9, varied=True)
    wav[1, lens[1]:] = 0
    wav = torch.from_numpy(wav)
    lms = O.logmel(wav, 32000)
    flen = O.cnn14_feat_len(lens)

    def logits(attn):
        enc = O.gru_forward(state, attn, flen)
        out = O.greedy_decode(state, enc["attn_emb"], enc["attn_emb_len"], 8)
        return out

    with torch.no_grad():
        ref_attn = conv_stack(state, lms, None, 6)
        ref = logits(ref_attn)
        st = ref["steps"]
        print(f"{sec} s clips, {flen.tolist()} frames, Winograd / split emulation on blocks 1..{blocks} (conv1 of block 1 exact)")
        nest = lambda mh, mw: (lambda x, w, fmt, flush: conv_wino_nest(x, w, fmt, flush, mh, mw))
        only = sys.argv[3].split(",") if len(sys.argv) > 3 else None
        for name, fn in (("direct", conv_direct), ("wino1d", conv_wino1d), ("wino2d", conv_wino2d), ("wino43", nest(4, 1)),
                         ("nest43x23", nest(4, 2)), ("nest43x43", nest(4, 4))):
            if only and name not in only:
                continue
            for fmt, flush in (("bf16", False), ("f16", False), ("f16", True)):
                if name.startswith(("wino43", "nest")) and fmt != "bf16":
                    continue
                attn = conv_stack(state, lms, lambda x, w: fn(x, w, fmt, flush), blocks)
                out = logits(attn)
                da = float((attn - ref_attn).abs().max())
                dl = float((out["logit"][:, :st] - ref["logit"][:, :st]).abs().max())
                eq = bool(torch.equal(out["seq"][:, :st], ref["seq"][:, :st]))
                print(f"  {name:7s} {fmt}{' flush-denormals' if flush else '':16s} attn_emb max|diff| {da:.2e} "
                      f"(max {float(ref_attn.abs().max()):.1f})  logits max|diff| {dl:.2e}  ids {'same' if eq else 'DIFFER'}",
                      flush=True)


if __name__ == "__main__":
    main()
