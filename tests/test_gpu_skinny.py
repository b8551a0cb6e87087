"""GPU parity tests of the weight-streaming conv kernel for few pixels (csrc/conv3x3_skinny.hip, through the C ABI): against
F.conv2d in fp32 on the CPU (reference ConvBlock.forward, cnn_encoder.py:59-75, eval BatchNorm folded to scale / shift; the
pooling / mean over mel of Cnn14Encoder.forward, cnn_encoder.py:431-444), the dropout epilogue bit for bit against the
two-pass form, and the single-clip / batch-4 encoder against the oracle."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _report(name, got, want):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    d = (got - want).abs()
    print(f"[{name}] shape {tuple(want.shape)} max|diff| {float(d.max()):.3e} mean|diff| {float(d.mean()):.3e}; "
          f"|want| max {float(want.abs().max()):.3e}")
    return float(d.max())


@pytest.fixture(scope="module")
def K():
    from audiocaption_amd import build
    build.build()
    from audiocaption_amd import kernels
    return kernels


def _to_rows(x_nchw, Hp):
    B, C, H, W = x_nchw.shape
    out = torch.zeros(B, Hp, W, C)
    out[:, :H] = x_nchw.permute(0, 2, 3, 1)
    return out.reshape(B * Hp, W, C).contiguous()


def _case(B, H, W, Cin, Cout, mode, seed):
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * math.sqrt(2.0 / (9 * Cin))
    sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
    y = F.relu(F.conv2d(x, w, padding=1) * sc[None, :, None, None] + sh[None, :, None, None])
    Hp = (H + 4) & ~3
    if mode == 2:
        want, shape = y.mean(dim=3).transpose(1, 2).contiguous(), (B, H, Cout)
    elif mode == 1:
        want, shape = _to_rows(F.avg_pool2d(y, 2), Hp // 2), (B * Hp // 2, W // 2, Cout)
    else:
        want, shape = _to_rows(y, Hp), (B * Hp, W, Cout)
    return x, w, sc, sh, Hp, want, shape


@pytest.mark.parametrize("B,H,W,Cin,Cout,mode", [
    (1, 31, 2, 1024, 2048, 0), (1, 31, 2, 2048, 2048, 2),      # block 6 of one 10 s clip: 2 pixel tiles, 32 K slices
    (4, 31, 2, 2048, 2048, 0), (4, 31, 2, 2048, 2048, 2),      # ... of four: 8 tiles
    (2, 29, 2, 256, 256, 2), (3, 13, 2, 64, 128, 0),           # 4 tiles; few chunks per slice
    (1, 62, 4, 512, 1024, 0), (1, 62, 4, 1024, 1024, 1),       # block 5 of one clip
    (4, 62, 4, 1024, 1024, 1), (3, 7, 4, 64, 128, 1), (7, 9, 4, 96, 256, 0),   # several m tiles, partial last one, odd chunk count
    (1, 93, 2, 128, 128, 2)])                                  # a 30 s clip
def test_skinny_conv_vs_conv2d(K, B, H, W, Cin, Cout, mode):
    """conv3x3 + BN + ReLU (+ pool / mean over mel) vs F.conv2d on the CPU.  Bar: the split-bf16 tiers' (2^-16 relative operand
    error on O(1..10) outputs): 1e-3 * sqrt(K / 576)."""
    x, w, sc, sh, Hp, want, shape = _case(B, H, W, Cin, Cout, mode, B * 1000 + H * 10 + W + Cin)
    n = K.skinny_workspace_floats(B, Hp, W, Cin, Cout)
    assert n > 0
    ws = torch.empty(n, device="cuda")
    out = torch.full(shape, 7.0).cuda()
    K.conv3x3_bn_relu_skinny(_to_rows(x, Hp).cuda(), K.pack_conv_weight_bf16x3_frag(w.cuda()), sc.cuda(), sh.cuda(), out, B, Hp, H, W,
                             Cin, Cout, mode, ws)
    tol = 1e-3 * max(1.0, math.sqrt(9 * Cin / 576))
    assert _report(f"conv[skinny] {B}x{H}x{W} {Cin}->{Cout} mode{mode}", out.reshape(want.shape), want) < tol
    # deterministic: the slices are added in order
    out2 = torch.full(shape, 3.0).cuda()
    K.conv3x3_bn_relu_skinny(_to_rows(x, Hp).cuda(), K.pack_conv_weight_bf16x3_frag(w.cuda()), sc.cuda(), sh.cuda(), out2, B, Hp, H, W,
                             Cin, Cout, mode, ws)
    assert torch.equal(out, out2)


@pytest.mark.parametrize("B,H,W,Cin,Cout,mode", [(4, 31, 2, 1024, 2048, 0), (2, 62, 4, 512, 1024, 1)])
def test_skinny_conv_dropout_epilogue(K, B, H, W, Cin, Cout, mode):
    """F.dropout in the finishing kernel = the layer followed by ac_dropout over the output buffer, bit for bit (the train-mode
    forward of the frozen network, cnn_encoder.py:431-442)."""
    x, w, sc, sh, Hp, want, shape = _case(B, H, W, Cin, Cout, mode, 5)
    ws = torch.empty(K.skinny_workspace_floats(B, Hp, W, Cin, Cout), device="cuda")
    xr, wp = _to_rows(x, Hp).cuda(), K.pack_conv_weight_bf16x3_frag(w.cuda())
    plain = torch.empty(shape, device="cuda")
    K.conv3x3_bn_relu_skinny(xr, wp, sc.cuda(), sh.cuda(), plain, B, Hp, H, W, Cin, Cout, mode, ws)
    seed_dev = torch.tensor([11], dtype=torch.int64, device="cuda")
    fused = torch.empty(shape, device="cuda")
    K.conv3x3_bn_relu_skinny(xr, wp, sc.cuda(), sh.cuda(), fused, B, Hp, H, W, Cin, Cout, mode, ws,
                             dropout=(0.2, 77, seed_dev.data_ptr()))
    K.dropout_(plain, plain.numel(), 0.2, 77, seed_dev.data_ptr())
    assert torch.equal(fused, plain)
    assert 0.1 < float((fused == 0).float().mean()) < 0.9


@pytest.mark.parametrize("B,seconds", [(1, 10.0), (4, 10.0), (2, 4.0)])
def test_training_forward_routes_blocks_5_6_through_the_skinny_kernel(hip_model, state4981, B, seconds, monkeypatch):
    """The frozen Cnn14 inside the training step (dropout after every block, cnn_encoder.py:431-442) at small batches: conv
    blocks 5-6 take the weight-streaming kernel (launch hook) - another arithmetic (direct instead of F(2,3), both on
    split-bf16 operands) with the SAME dropout masks: against the F(2,3) route, zeros in the same places and values within the
    tier's bar; inference keeps the F(2,3) route."""
    from audiocaption_amd import cnn_encoder as CE, kernels as Kn, procedural as P
    cnn = hip_model.encoder.cnn
    L = int(seconds * 32000)
    wav = torch.from_numpy(P.synthetic_wav(B, L, varied=True, seed=B)).cuda()
    seed_dev = torch.tensor([5], dtype=torch.int64, device="cuda")

    def run(skinny, train):
        seen = []

        def hook(phase, info):
            if phase == "pre":
                seen.append((info["algo"], info["W"]))

        monkeypatch.setattr(CE, "SKINNY", skinny)
        Kn.CONV_LAUNCH_HOOK = hook
        try:
            if train:
                out = cnn.encode(wav, dropout=(0.2, 40, seed_dev.data_ptr()), train=True)
            else:
                out = cnn({"wav": wav, "wav_len": [L] * B}, skip_fc=True)["attn_emb"]
        finally:
            Kn.CONV_LAUNCH_HOOK = None
        torch.cuda.synchronize()
        return out.clone(), [a for a, w_ in seen if w_ in (4, 2)]

    got, route = run(True, True)
    want, route0 = run(False, True)
    assert route == ["skinny"] * 4 and route0 == ["wino1d"] * 4, (route, route0)
    d = float((got - want).abs().max())
    print(f"B={B} {seconds} s train forward: max|diff| {d:.2e} (|want| max {float(want.abs().max()):.2e})")
    assert d < 5e-4
    _, route_inf = run(True, False)
    assert route_inf == ["wino1d"] * 4, route_inf
