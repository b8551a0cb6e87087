"""GPU parity tests of the F(4,3) conv kernel (csrc/conv3x3_wino43.hip, through the C ABI): against F.conv2d in fp32 on the
CPU (the arithmetic of reference ConvBlock.forward, cnn_encoder.py:59-75, with eval BatchNorm folded to scale / shift),
against the F(2,3) kernel on the same inputs, dead-row skipping and the dropout epilogue bit for bit against their
two-pass forms.  Tolerances are absolute, fp32, stated per test."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _report(name, got, want):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    d = (got - want).abs()
    i = int(d.argmax())
    print(f"[{name}] shape {tuple(want.shape)} max|diff| {float(d.max()):.3e} mean|diff| {float(d.mean()):.3e} "
          f"at flat {i}: got {float(got.flatten()[i]):.6f} want {float(want.flatten()[i]):.6f}; "
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


def _hp4(H):
    return (H + 4) & ~3   # multiple of 4, at least one zero row


def _case(B, H, W, Cin, Cout, mode, seed):
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * math.sqrt(2.0 / (9 * Cin))
    sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
    y = F.relu(F.conv2d(x, w, padding=1) * sc[None, :, None, None] + sh[None, :, None, None])
    Hp = _hp4(H)
    want = _to_rows(F.avg_pool2d(y, 2) if mode == 1 else y, Hp // 2 if mode == 1 else Hp)
    shape = (B * Hp // 2, W // 2, Cout) if mode == 1 else (B * Hp, W, Cout)
    return x, w, sc, sh, Hp, want, shape


@pytest.mark.parametrize("B,H,W,Cin,Cout,mode", [
    (3, 21, 32, 64, 128, 0), (2, 20, 32, 128, 128, 1), (2, 11, 16, 128, 256, 0), (5, 30, 16, 32, 128, 1),
    (2, 9, 8, 256, 512, 1), (3, 50, 8, 256, 512, 0), (3, 7, 4, 512, 1024, 0), (3, 6, 4, 1024, 1024, 1), (1, 95, 4, 512, 1024, 1)])
@pytest.mark.parametrize("map_mode", [-1, 0])
@pytest.mark.parametrize("tiles", [1, 2])
def test_conv3x3_wino43_vs_conv2d(K, B, H, W, Cin, Cout, mode, map_mode, tiles):
    """conv3x3 + BN + ReLU (+ 2x2 average pool) vs F.conv2d on the CPU, both tile counts per wave (half-size workgroups, two per CU, and full-size ones) and both block maps.
    Bar: the split-bf16 tiers' (2^-16 relative operand error on O(1..10) outputs): 1e-3 * sqrt(K / 576)."""
    x, w, sc, sh, Hp, want, shape = _case(B, H, W, Cin, Cout, mode, B * 1000 + H * 10 + W + Cin)
    out = torch.full(shape, 7.0).cuda()
    K.conv3x3_bn_relu_wino43(_to_rows(x, Hp).cuda(), K.pack_conv_weight_wino43_frag(w.cuda()), sc.cuda(), sh.cuda(), out,
                             B, Hp, H, W, Cin, Cout, mode, map_mode, tiles_per_wave=tiles)
    tol = 1e-3 * max(1.0, math.sqrt(9 * Cin / 576))
    assert _report(f"conv[wino43 x{tiles}] {B}x{H}x{W} {Cin}->{Cout} mode{mode}", out.reshape(want.shape), want) < tol


@pytest.mark.parametrize("B,H,Cin,Cout,mode", [(5, 3, 1024, 2048, 0), (5, 3, 2048, 2048, 2), (1, 31, 64, 128, 2), (3, 31, 512, 256, 0),
                                               (37, 31, 256, 256, 2), (2, 93, 128, 128, 2)])
def test_conv3x3_wino43_two_column_form(K, B, H, Cin, Cout, mode):
    """Conv block 6 (W = 2): column tiles, the taps on the zero padding beside the image skipped; mode 0 and the mean over the
    two mel columns (cnn_encoder.py:443) against F.conv2d on the CPU."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(B * 100 + H + Cin)
    x = torch.randn(B, Cin, H, 2, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * math.sqrt(2.0 / (9 * Cin))
    sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
    y = F.relu(F.conv2d(x, w, padding=1) * sc[None, :, None, None] + sh[None, :, None, None])
    Hp = _hp4(H)
    if mode == 2:
        want = y.mean(dim=3).transpose(1, 2).contiguous()
        out = torch.full((B, H, Cout), 7.0).cuda()
    else:
        want = _to_rows(y, Hp)
        out = torch.full((B * Hp, 2, Cout), 7.0).cuda()
    K.conv3x3_bn_relu_wino43(_to_rows(x, Hp).cuda(), K.pack_conv_weight_wino43_frag(w.cuda()), sc.cuda(), sh.cuda(), out, B, Hp, H,
                             2, Cin, Cout, mode)
    tol = 1e-3 * max(1.0, math.sqrt(9 * Cin / 576))
    assert _report(f"conv[wino43, W = 2] {B}x{H} {Cin}->{Cout} mode{mode}", out.reshape(want.shape), want) < tol


@pytest.mark.parametrize("B,H,W,Cin,Cout,mode", [(16, 250, 16, 128, 256, 0), (16, 250, 16, 256, 256, 1), (34, 125, 8, 256, 512, 1),
                                                  (9, 500, 32, 64, 128, 0), (40, 62, 4, 512, 1024, 1)])
def test_conv3x3_wino43_many_workgroups(K, B, H, W, Cin, Cout, mode):
    """Launches of several hundred workgroups (tiles spanning clips, partial last row block, every block map the dispatcher
    picks on its own): against the CPU reference, and bit for bit against the first clips convolved alone."""
    x, w, sc, sh, Hp, want, shape = _case(B, H, W, Cin, Cout, mode, B + W + Cin)
    xr, wp = _to_rows(x, Hp).cuda(), K.pack_conv_weight_wino43_frag(w.cuda())
    out = torch.full(shape, 7.0).cuda()
    K.conv3x3_bn_relu_wino43(xr, wp, sc.cuda(), sh.cuda(), out, B, Hp, H, W, Cin, Cout, mode)
    tol = 1e-3 * max(1.0, math.sqrt(9 * Cin / 576))
    assert _report(f"conv[wino43] {B}x{H}x{W} {Cin}->{Cout} mode{mode}", out.reshape(want.shape), want) < tol
    nb = 3   # quads of clip 3 never share a workgroup's OUTPUT with clip 2's, and a clip's result does not depend on its neighbours
    few = torch.full((nb * shape[0] // B,) + shape[1:], 7.0).cuda()
    K.conv3x3_bn_relu_wino43(xr[:nb * Hp].contiguous(), wp, sc.cuda(), sh.cuda(), few, nb, Hp, H, W, Cin, Cout, mode)
    rows = few.shape[0] // nb * (nb - 1)   # the last clip's final quad sees "beyond the batch" instead of the next clip: equal anyway
    assert torch.equal(few[:rows], out[:rows])
    assert torch.equal(few, out[:few.shape[0]])


@pytest.mark.parametrize("B,H,W,Cin,Cout,mode", [(4, 125, 8, 256, 512, 1), (4, 250, 16, 128, 256, 0), (6, 62, 4, 512, 1024, 1)])
def test_conv3x3_wino43_vs_wino1d(K, B, H, W, Cin, Cout, mode):
    """F(4,3) and F(2,3) on the same inputs: both carry the 2^-16 operand error, so they agree to the same bar they are
    held to against fp32 (and not better: different transforms)."""
    x, w, sc, sh, Hp, want, shape = _case(B, H, W, Cin, Cout, mode, 5 * B + W)
    xr = _to_rows(x, Hp).cuda()
    a, b = torch.full(shape, 7.0).cuda(), torch.full(shape, 7.0).cuda()
    K.conv3x3_bn_relu_wino43(xr, K.pack_conv_weight_wino43_frag(w.cuda()), sc.cuda(), sh.cuda(), a, B, Hp, H, W, Cin, Cout, mode)
    K.conv3x3_bn_relu_wino1d(xr, K.pack_conv_weight_wino1d_frag(w.cuda()), sc.cuda(), sh.cuda(), b, B, Hp, H, W, Cin, Cout, mode)
    assert _report(f"wino43 vs wino1d {B}x{H}x{W} {Cin}->{Cout} mode{mode}", a, b) < 1e-3 * max(1.0, math.sqrt(9 * Cin / 576))


@pytest.mark.parametrize("W,Cin,Cout,mode,block,conv", [(32, 64, 128, 0, 2, 1), (16, 128, 256, 0, 3, 1), (8, 512, 512, 1, 4, 2),
                                                       (4, 512, 1024, 0, 5, 1)])
def test_conv3x3_wino43_dead_row_skipping(K, W, Cin, Cout, mode, block, conv):
    """Per-clip dead rows (ragged batches): output rows below ``mul * frames[b] + add`` are bit-identical to the full
    convolution, workgroups wholly beyond it store zeros, and at least one workgroup is skipped on this ragged set."""
    from audiocaption_amd.cnn_encoder import rows_needed
    g = torch.Generator().manual_seed(W * 7 + Cin)
    frames = torch.tensor([90, 4, 50, 2, 30], dtype=torch.int32)
    B = len(frames)
    H = 90 << (6 - block)
    Hp = _hp4(H)
    x = torch.zeros(B, Hp, W, Cin)
    x[:, :H] = torch.randn(B, H, W, Cin, generator=g)
    x = x.reshape(B * Hp, W, Cin).cuda()
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * math.sqrt(2.0 / (9 * Cin))
    wp = K.pack_conv_weight_wino43_frag(w.cuda())
    sc, sh = (torch.rand(Cout, generator=g) + 0.5).cuda(), (torch.randn(Cout, generator=g) * 0.1).cuda()
    shape = (B * Hp, W, Cout) if mode == 0 else (B * Hp // 2, W // 2, Cout)
    full, skip = torch.full(shape, 7.0).cuda(), torch.full(shape, 7.0).cuda()
    K.conv3x3_bn_relu_wino43(x, wp, sc, sh, full, B, Hp, H, W, Cin, Cout, mode)
    mul, add = rows_needed(block, conv)
    K.conv3x3_bn_relu_wino43(x, wp, sc, sh, skip, B, Hp, H, W, Cin, Cout, mode, need=(frames.cuda(), mul, add))
    rows = Hp if mode == 0 else Hp // 2
    f, s_ = full.reshape(B, rows, -1).cpu(), skip.reshape(B, rows, -1).cpu()
    zeroed = 0
    for b in range(B):
        need = min(int(mul * frames[b] + add), H)
        n = need if mode != 1 else need // 2
        assert torch.equal(f[b, :n], s_[b, :n]), (b, n)
        tail = s_[b, n:]
        same = (tail == f[b, n:]).all(dim=1)
        zero = (tail == 0).all(dim=1)
        assert bool((same | zero).all())
        zeroed += int((zero & ~same).sum())
    assert zeroed > 0


@pytest.mark.parametrize("B,H,W,Cin,Cout,mode", [(3, 21, 16, 128, 256, 1), (2, 13, 4, 512, 1024, 1), (2, 40, 32, 64, 128, 0)])
def test_conv3x3_wino43_dropout_in_the_epilogue(K, B, H, W, Cin, Cout, mode):
    """Train-mode forward of the frozen network: F.dropout inside the epilogue equals, bit for bit, the layer followed by the
    counter-hash dropout pass over its output buffer - with and without the device-side step seed."""
    g = torch.Generator().manual_seed(B * 31 + W)
    Hp = _hp4(H)
    x = torch.zeros(B, Hp, W, Cin)
    x[:, :H] = torch.randn(B, H, W, Cin, generator=g)
    x = x.reshape(B * Hp, W, Cin).cuda()
    wp = K.pack_conv_weight_wino43_frag((torch.randn(Cout, Cin, 3, 3, generator=g) * math.sqrt(2.0 / (9 * Cin))).cuda())
    sc, sh = (torch.rand(Cout, generator=g) + 0.5).cuda(), (torch.randn(Cout, generator=g) * 0.1).cuda()
    shape = (B * Hp, W, Cout) if mode == 0 else (B * Hp // 2, W // 2, Cout)
    step = torch.tensor([5], dtype=torch.int64, device="cuda")
    for seed_dev in (None, step.data_ptr()):
        two, one = torch.full(shape, 7.0).cuda(), torch.full(shape, 7.0).cuda()
        K.conv3x3_bn_relu_wino43(x, wp, sc, sh, two, B, Hp, H, W, Cin, Cout, mode)
        K.dropout_(two, two.numel(), 0.2, 1234, seed_dev)
        K.conv3x3_bn_relu_wino43(x, wp, sc, sh, one, B, Hp, H, W, Cin, Cout, mode, dropout=(0.2, 1234, seed_dev))
        assert torch.equal(one, two)
        assert 0.1 < float((one == 0).float().mean()) < 0.9


def test_conv3x3_wino43_rejects_what_it_does_not_cover(K):
    from audiocaption_amd._lib import HipLibraryError
    x = torch.zeros(2 * 8, 2, 64).cuda()
    out = torch.zeros(2 * 8, 2, 128).cuda()
    wp = torch.zeros(4, 18, 4, 2, 64, 8, dtype=torch.bfloat16).cuda()
    sc = torch.ones(128).cuda()
    with pytest.raises(HipLibraryError):
        K.conv3x3_bn_relu_wino43(x, wp, sc, sc, out, 2, 8, 5, 2, 64, 128, 1)      # W = 2 is never pooled
    x = torch.zeros(2 * 6, 4, 64).cuda()
    with pytest.raises(HipLibraryError):
        K.conv3x3_bn_relu_wino43(x, wp, sc, sc, out, 2, 6, 5, 4, 64, 128, 0)      # Hp % 4 != 0
    assert K.wino43_workgroups(64, 256, 16, 256) == (64 * 64 + 3) // 4 * 2 and K.wino43_workgroups(64, 32, 2, 2048) == 16 * 16


# ---- conv block 1 in one kernel (csrc/conv3x3_block1_w4.hip) ----------------------------------------------------------
def _block1_case(B, H, seed):
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(seed)
    Hp = (H + 8) & ~7
    x = torch.randn(B, 1, H, 64, generator=g)
    w1 = torch.randn(64, 1, 3, 3, generator=g) * 0.5
    s1, t1 = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1
    w2 = torch.randn(64, 64, 3, 3, generator=g) * math.sqrt(2.0 / (9 * 64))
    s2, t2 = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1
    y1 = F.relu(F.conv2d(x, w1, padding=1) * s1[None, :, None, None] + t1[None, :, None, None])
    y2 = F.relu(F.conv2d(y1, w2, padding=1) * s2[None, :, None, None] + t2[None, :, None, None])
    want = _to_rows(F.avg_pool2d(y2, 2), Hp // 2)
    x0 = torch.zeros(B, Hp, 64)
    x0[:, :H] = x[:, 0]
    return x0.reshape(B * Hp, 64), w1.reshape(64, 9), s1, t1, w2, s2, t2, Hp, want


CONV1_FORMS = ["mfma", "valu"]   # conv1 of the one-kernel block: on the matrix cores (default) | the f32 chain of ac_conv3x3_first


@pytest.mark.parametrize("conv1", CONV1_FORMS)
@pytest.mark.parametrize("B,H", [(1, 13), (3, 37), (2, 250), (5, 1001)])
def test_block1_wino43_vs_conv2d_and_vs_the_two_kernel_form(K, B, H, conv1):
    """conv_block1 + avg_pool2d of the reference (cnn_encoder.py:59-75, :431-432) in ONE kernel against F.conv2d in fp32 on the
    CPU (bar of the split-bf16 tiers), and against conv1 (ac_conv3x3_first) followed by the unfused form of the same kernel
    reading the 64-channel intermediate from HBM: bit for bit with conv1 as the same f32 chain ("valu"), within the
    split-bf16 grade (2e-5 of the largest output) with conv1 as a split-bf16 product ("mfma")."""
    x0, w1, s1, t1, w2, s2, t2, Hp, want = _block1_case(B, H, 100 * B + H)
    dev = "cuda"
    x0, w1, s1, t1, s2, t2 = (t.to(dev) for t in (x0, w1, s1, t1, s2, t2))
    wp = K.pack_conv_weight_wino43_frag(w2.to(dev))
    fused = torch.full((B * Hp // 2, 32, 64), 7.0, device=dev)
    K.conv3x3_block1_wino43(x0, w1, s1, t1, wp, s2, t2, fused, B, Hp, H, conv1=conv1)
    assert _report(f"block1 fused [{conv1}] {B}x{H}", fused.reshape(want.shape), want) < 1e-3
    mid = torch.full((B * Hp, 64, 64), 7.0, device=dev)
    K.conv3x3_first(x0, w1, s1, t1, mid, B, Hp, H)
    two = torch.full((B * Hp // 2, 32, 64), 7.0, device=dev)
    K.conv3x3_block1_conv2_wino43(mid, wp, s2, t2, two, B, Hp, H)
    if conv1 == "valu":
        assert torch.equal(fused, two)
    else:
        d = float((fused - two).abs().max())
        print(f"block1 [mfma] vs the two-kernel form: max|diff| {d:.3e} (|out| max {float(two.abs().max()):.3e})")
        assert d < 2e-5 * float(two.abs().max())
    again = torch.full_like(fused, 7.0)
    K.conv3x3_block1_wino43(x0, w1, s1, t1, wp, s2, t2, again, B, Hp, H, conv1=conv1)
    assert torch.equal(fused, again)


@pytest.mark.parametrize("conv1", CONV1_FORMS)
def test_block1_wino43_many_tiles_per_workgroup(K, conv1):
    """More tiles than CUs (persistent workgroups walk several tiles each; the staging of a tile's first K step runs under
    the previous tile's last): 40 clips x 500 rows = 2560 tiles, every clip equal to the same clip convolved alone."""
    B, H = 40, 500
    x0, w1, s1, t1, w2, s2, t2, Hp, _ = _block1_case(1, H, 7)
    dev = "cuda"
    g = torch.Generator().manual_seed(3)
    xs = torch.zeros(B, Hp, 64)
    xs[:, :H] = torch.randn(B, H, 64, generator=g)
    xs = xs.reshape(B * Hp, 64).to(dev)
    w1, s1, t1, s2, t2 = (t.to(dev) for t in (w1, s1, t1, s2, t2))
    wp = K.pack_conv_weight_wino43_frag(w2.to(dev))
    out = torch.full((B * Hp // 2, 32, 64), 7.0, device=dev)
    K.conv3x3_block1_wino43(xs, w1, s1, t1, wp, s2, t2, out, B, Hp, H, conv1=conv1)
    for b in (0, 17, 39):
        one = torch.full((Hp // 2, 32, 64), 7.0, device=dev)
        K.conv3x3_block1_wino43(xs[b * Hp:(b + 1) * Hp].contiguous(), w1, s1, t1, wp, s2, t2, one, 1, Hp, H, conv1=conv1)
        assert torch.equal(one, out[b * Hp // 2:(b + 1) * Hp // 2])


@pytest.mark.parametrize("conv1", CONV1_FORMS)
def test_block1_wino43_dead_rows_and_dropout(K, conv1):
    """Ragged batches: rows below a clip's need are bit-identical to the full convolution, tiles beyond it are zeros (and
    some are skipped); F.dropout in the epilogue equals the separate counter-hash pass bit for bit."""
    from audiocaption_amd.cnn_encoder import rows_needed
    B, H = 5, 2900
    x0, w1, s1, t1, w2, s2, t2, Hp, _ = _block1_case(1, 16, 11)
    dev = "cuda"
    g = torch.Generator().manual_seed(5)
    xs = torch.zeros(B, (H + 8) & ~7, 64)
    Hp = xs.shape[1]
    xs[:, :H] = torch.randn(B, H, 64, generator=g)
    xs = xs.reshape(B * Hp, 64).to(dev)
    w1, s1, t1, s2, t2 = (t.to(dev) for t in (w1, s1, t1, s2, t2))
    wp = K.pack_conv_weight_wino43_frag(w2.to(dev))
    frames = torch.tensor([80, 4, 50, 2, 30], dtype=torch.int32)
    mul, add = rows_needed(1, 2, quads=True)
    full, skip = (torch.full((B * Hp // 2, 32, 64), 7.0, device=dev) for _ in range(2))
    K.conv3x3_block1_wino43(xs, w1, s1, t1, wp, s2, t2, full, B, Hp, H, conv1=conv1)
    K.conv3x3_block1_wino43(xs, w1, s1, t1, wp, s2, t2, skip, B, Hp, H, need=(frames.to(dev), mul, add), conv1=conv1)
    f, s_ = full.reshape(B, Hp // 2, -1).cpu(), skip.reshape(B, Hp // 2, -1).cpu()
    zeroed = 0
    for b in range(B):
        n = min(int(mul * frames[b] + add), H) // 2
        assert torch.equal(f[b, :n], s_[b, :n]), (b, n)
        tail = s_[b, n:]
        same, zero = (tail == f[b, n:]).all(dim=1), (tail == 0).all(dim=1)
        assert bool((same | zero).all())
        zeroed += int((zero & ~same).sum())
    assert zeroed > 0
    step = torch.tensor([5], dtype=torch.int64, device=dev)
    for seed_dev in (None, step.data_ptr()):
        two = full.clone()
        K.dropout_(two, two.numel(), 0.2, 77, seed_dev)
        one = torch.full_like(full, 7.0)
        K.conv3x3_block1_wino43(xs, w1, s1, t1, wp, s2, t2, one, B, Hp, H, dropout=(0.2, 77, seed_dev), conv1=conv1)
        assert torch.equal(one, two)
