"""GPU parity tests, kernel by kernel: the HIP path (through the C ABI) against the CPU oracle on the
same seeded inputs.  Tolerances are absolute, fp32, stated per test."""
import math
import os

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


def test_library_loaded_is_in_tree():
    from audiocaption_amd import _lib, build
    build.build()
    _lib.load()
    assert os.path.exists(_lib.LIB_PATH)
    with open("/proc/self/maps") as f:
        assert "libaudiocaption_hip.so" in f.read()


@pytest.mark.parametrize("M,N,K_", [(64, 256, 256), (64, 768, 256), (64, 4981, 256), (64, 256, 1024),
                                    (4, 768, 256), (192, 4981, 256), (1984, 1536, 2048), (837, 512, 256),
                                    (130, 70, 96)])
@pytest.mark.parametrize("relu", [False, True])
def test_linear(K, M, N, K_, relu):
    g = torch.Generator().manual_seed(M * 131 + N)
    x = torch.randn(M, K_, generator=g)
    w = torch.randn(N, K_, generator=g) / math.sqrt(K_)
    b = torch.randn(N, generator=g)
    want = torch.nn.functional.linear(x, w, b)
    if relu:
        want = want.relu()
    got = K.linear(x.cuda(), w.cuda(), b.cuda(), relu=relu)
    assert _report(f"linear {M}x{N}x{K_}", got, want) < 2e-5 * max(1.0, math.sqrt(K_ / 256))


@pytest.mark.parametrize("M,N,K_,relu,bias", [(1984, 1536, 2048, False, True), (1984, 1536, 512, False, True),
                                               (31, 1536, 512, True, True), (1985, 1536, 2048, False, False),
                                               (130, 832, 1024, True, False)])
def test_linear_split_bf16_path(K, M, N, K_, relu, bias):
    """The large linear layers run as the one-tap instance of the split-bf16 conv kernel (ac_linear_bf16x3): against
    float64, with the bar of the exact-f32 GEMM plus the 2^-16 operand error."""
    g = torch.Generator().manual_seed(M + 7 * N + K_)
    x = torch.randn(M, K_, generator=g)
    w = torch.randn(N, K_, generator=g) / math.sqrt(K_)
    b = torch.randn(N, generator=g) if bias else None
    want = torch.nn.functional.linear(x.double(), w.double(), b.double() if bias else None)
    if relu:
        want = want.relu()
    assert K.LINEAR_ALGO == "bf16x3"
    xd, wd, bd = x.cuda(), w.cuda(), (b.cuda() if bias else None)
    got = K.linear(xd, wd, bd, relu=relu)
    assert _report(f"linear bf16x3 {M}x{N}x{K_}", got, want.float()) < 4e-5 * max(1.0, math.sqrt(K_ / 256))
    # the exact-f32 GEMM on the same inputs shows which path ran above (it differs in the last bits)
    saved, K.LINEAR_ALGO = K.LINEAR_ALGO, "f32"
    try:
        f32 = K.linear(xd, wd, bd, relu=relu)
    finally:
        K.LINEAR_ALGO = saved
    assert _report("f32 GEMM", f32, want.float()) < 2e-5 * max(1.0, math.sqrt(K_ / 256))
    assert not torch.equal(f32, got)
    # a second weight that reuses the first one's storage address must not hit the packed-weight cache
    del wd
    w2 = (torch.randn(N, K_, generator=g) / math.sqrt(K_)).cuda()
    got2 = K.linear(xd, w2, bd, relu=relu)
    want2 = torch.nn.functional.linear(x.double(), w2.cpu().double(), b.double() if bias else None)
    assert _report("second weight", got2, (want2.relu() if relu else want2).float()) < 4e-5 * max(1.0, math.sqrt(K_ / 256))


def test_linear_asymmetric_identity(K):
    """transpose-detecting check: X = I gives Y = W^T rows, with an asymmetric W."""
    n = 64
    w = torch.arange(n * n, dtype=torch.float32).reshape(n, n) / 100.0
    got = K.linear(torch.eye(n).cuda(), w.cuda(), None)
    assert torch.equal(got.cpu(), w.t().contiguous())


def test_linear_strided_rows(K):
    g = torch.Generator().manual_seed(5)
    big = torch.randn(64, 20 * 256, generator=g)
    w = torch.randn(300, 256, generator=g)
    x = big[:, 3 * 256:4 * 256]  # row stride 5120, like embed[b, t]
    got = K.linear(x.cuda() if False else big.cuda()[:, 3 * 256:4 * 256], w.cuda(), None)
    assert _report("linear strided", got, x @ w.t()) < 5e-5


def test_add_layernorm(K):
    g = torch.Generator().manual_seed(1)
    x, y = torch.randn(70, 256, generator=g), torch.randn(70, 256, generator=g)
    w, b = torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g)
    want = torch.nn.functional.layer_norm(x + y, (256,), w, b)
    got = K.add_layernorm(x.cuda(), y.cuda(), w.cuda(), b.cuda())
    assert _report("add_layernorm", got, want) < 5e-6
    got = K.add_layernorm(x.cuda(), None, w.cuda(), b.cuda())
    assert _report("layernorm", got, torch.nn.functional.layer_norm(x, (256,), w, b)) < 5e-6


@pytest.mark.parametrize("L,B", [(320000, 2), (64000, 3), (35231, 2)])
def test_logmel_matches_oracle(K, L, B):
    """fp32 log-mel in dB vs the oracle (torch.stft path).  Tolerance 2e-3 dB absolute: both are fp32
    transforms of 1024 points followed by 10*log10; the noise inputs have no near-empty bins."""
    from audiocaption_amd import procedural as P
    from audiocaption_amd.mel import MelTables
    from oracle import cpu_path as O
    wav = torch.from_numpy(P.synthetic_wav(B, L, varied=True))
    want = O.logmel(wav, 32000)
    tables = MelTables(32000, 1024, 320, 50.0, 14000.0, 64, "slaney", "slaney", "cuda:0")
    got = K.logmel(wav.cuda(), tables, channels_last=False)
    assert got.shape == want.shape
    assert _report(f"logmel L={L}", got, want) < 2e-3
    # rows layout + folded affine + zero pad rows
    T = L // 320 + 1
    Hp = T + 5
    sc, sh = torch.rand(64) + 0.5, torch.randn(64)
    rows = K.logmel(wav.cuda(), tables, sc.cuda(), sh.cuda(), rows_per_clip=Hp).cpu().reshape(B, Hp, 64)
    want_rows = want.transpose(1, 2) * sc + sh
    assert _report("logmel rows", rows[:, :T], want_rows) < 4e-3
    assert float(rows[:, T:].abs().max()) == 0.0


def test_logmel_vs_independent_witness(K, golden_dir):
    """Row A1 / A2 against vectors this repository did not compute: ``tests/golden/g10_logmel.npz`` holds the float64
    ``transformers.audio_utils`` log-mel (slaney scale + slaney norm, 50-14000 Hz, n_fft 1024, hop 320, 10 log10, no
    top_db: torchaudio's MelSpectrogram + AmplitudeToDB as called at cnn_encoder.py:338-350,418-419) of seeded inputs
    (tests/golden/make_witness.py).  Bars in dB: 3e-3 on the maximum (an f32 transform against an f64 one; the largest
    differences sit ~60 dB below the clip's peak), 3e-4 on the 99th percentile."""
    import os
    import sys
    import numpy as np
    from audiocaption_amd.mel import MelTables
    sys.path.insert(0, golden_dir)
    import make_witness as W
    g = np.load(os.path.join(golden_dir, "g10_logmel.npz"))
    wav32, _ = W.logmel_inputs()
    tables = MelTables(32000, 1024, 320, 50.0, 14000.0, 64, "slaney", "slaney", "cuda:0")
    assert float((tables.melfb.cpu() - torch.from_numpy(g["fb_slaney"])).abs().max()) < 1e-6
    got = K.logmel(torch.from_numpy(wav32).cuda(), tables, channels_last=False).cpu().double().numpy()
    d = np.abs(got - g["cnn14_db"].astype(np.float64))
    print(f"[log-mel vs transformers.audio_utils witness] max|diff| {d.max():.3e} dB, p99 {np.percentile(d, 99):.3e} dB")
    assert d.max() < 3e-3 and np.percentile(d, 99) < 3e-4


def test_logmel_known_answers(K):
    """zeros -> -100 dB everywhere (clamp at 1e-10); a bin-centred sinusoid peaks in the right mel band."""
    from audiocaption_amd.mel import MelTables
    tables = MelTables(32000, 1024, 320, 50.0, 14000.0, 64, "slaney", "slaney", "cuda:0")
    z = K.logmel(torch.zeros(1, 32000).cuda(), tables, channels_last=False).cpu()
    assert float((z + 100.0).abs().max()) < 1e-4  # 10*log10f(1e-10f) is -100 to within an ulp
    t = torch.arange(32000) / 32000.0
    f0 = 32000.0 / 1024 * 100  # bin 100 = 3125 Hz
    s = K.logmel(torch.sin(2 * math.pi * f0 * t)[None].cuda(), tables, channels_last=False).cpu()[0]
    fb = tables.melfb.cpu()
    assert int(s[:, 50].argmax()) == int(fb[100].argmax())


def _ref_conv_block_input(B, H, W, C, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B, C, H, W, generator=g)


def _to_rows(x_nchw, Hp):
    """(B, C, H, W) -> [B*Hp][W][C] with zero pad rows."""
    B, C, H, W = x_nchw.shape
    out = torch.zeros(B, Hp, W, C)
    out[:, :H] = x_nchw.permute(0, 2, 3, 1)
    return out.reshape(B * Hp, W, C).contiguous()


@pytest.mark.parametrize("B,H,W,Cin,Cout,mode", [
    (2, 13, 64, 64, 64, 0), (2, 13, 64, 64, 64, 1), (3, 21, 32, 64, 128, 0), (2, 20, 32, 128, 128, 1),
    (2, 11, 16, 128, 256, 0), (2, 9, 8, 256, 512, 1), (3, 7, 4, 512, 1024, 0), (3, 6, 4, 1024, 1024, 1),
    (5, 3, 2, 1024, 2048, 0), (5, 3, 2, 2048, 2048, 2), (1, 31, 2, 64, 128, 2), (2, 30, 16, 32, 64, 1)])
@pytest.mark.parametrize("map_mode", [-1, 0])
@pytest.mark.parametrize("algo", ["direct", "winograd", "bf16x3", "bf16x3_gw", "f16x2", "wino1d"])
def test_conv3x3_bn_relu(K, B, H, W, Cin, Cout, mode, map_mode, algo):
    """conv3x3+BN+ReLU(+pool / +mean over W) vs F.conv2d on the CPU.  Tolerance 1e-4 * sqrt(K/576) abs on
    O(1) activations (fp32 accumulation-order differences only)."""
    import torch.nn.functional as F
    if map_mode == 0 and Cin > 512:
        pytest.skip("linear mapping only exercised on the small shapes")
    if algo == "wino1d" and ((Cout % 128 and not (Cout == 64 and W % 16 == 0)) or (mode == 2 and W != 2)):
        pytest.skip("the F(2,3) kernel covers 128-channel column tiles and the 64-channel / 16-column form")
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + W + Cin)
    x = torch.randn(B, Cin, H, W, generator=g)
    if algo == "f16x2":
        # the tier's input format IS fp16: the reference convolves the same fp16-representable activations, so the
        # test isolates the kernel (column tiles, tap skipping, fp16 hi+lo weights, packed fp16 stores)
        x = x.half().float()
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * math.sqrt(2.0 / (9 * Cin))
    sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
    y = F.relu(F.conv2d(x, w, padding=1) * sc[None, :, None, None] + sh[None, :, None, None])
    Hp = H + 1 + ((H + 1) % 2)  # even, >= H+1
    if mode == 1:
        want = F.avg_pool2d(y, 2)
        Ho, Wo, Hpo = H // 2, W // 2, Hp // 2
        want_rows = _to_rows(want, Hpo)
        out = torch.full((B * Hpo, Wo, Cout), 7.0).cuda()
    elif mode == 0:
        want_rows = _to_rows(y, Hp)
        out = torch.full((B * Hp, W, Cout), 7.0).cuda()
    else:
        want_rows = y.mean(dim=3).transpose(1, 2).contiguous()  # (B, H, Cout)
        out = torch.full((B, H, Cout), 7.0).cuda()
    if algo == "direct":
        K.conv3x3_bn_relu(_to_rows(x, Hp).cuda(), K.pack_conv_weight(w.cuda()), sc.cuda(), sh.cuda(), out, B, Hp, H,
                          W, Cin, Cout, mode, map_mode)
    elif algo == "winograd":
        K.conv3x3_bn_relu_winograd(_to_rows(x, Hp).cuda(), K.pack_conv_weight_winograd(w.cuda()), sc.cuda(),
                                   sh.cuda(), out, B, Hp, H, W, Cin, Cout, mode, map_mode)
    elif algo == "bf16x3":
        K.conv3x3_bn_relu_bf16x3(_to_rows(x, Hp).cuda(), K.pack_conv_weight_bf16x3(w.cuda()), sc.cuda(),
                                 sh.cuda(), out, B, Hp, H, W, Cin, Cout, mode, map_mode)
    elif algo == "wino1d":
        K.conv3x3_bn_relu_wino1d(_to_rows(x, Hp).cuda(), K.pack_conv_weight_wino1d_frag(w.cuda()), sc.cuda(),
                                 sh.cuda(), out, B, Hp, H, W, Cin, Cout, mode, map_mode)
    elif algo == "f16x2":
        wfrag, inv = K.pack_conv_weight_f16x2_frag(w.cuda())
        out16 = out if mode == 2 else torch.full(out.shape, 7.0, dtype=torch.float16, device="cuda")
        K.conv3x3_bn_relu_f16x2_gw(_to_rows(x, Hp).cuda().half(), wfrag, (sc.cuda() * inv).contiguous(), sh.cuda(), out16,
                                   B, Hp, H, W, Cin, Cout, mode, map_mode)
        out = out16.float()
        if mode != 2:
            want_rows = want_rows.half().float()   # fp16 storage of the output
    else:
        K.conv3x3_bn_relu_bf16x3_gw(_to_rows(x, Hp).cuda(), K.pack_conv_weight_bf16x3_frag(w.cuda()), sc.cuda(),
                                    sh.cuda(), out, B, Hp, H, W, Cin, Cout, mode, map_mode)
    tol = 1e-4 * max(1.0, math.sqrt(9 * Cin / 576))
    if algo.startswith("bf16x3") or algo == "wino1d":
        tol *= 10  # split-bf16 tier: 2^-16 relative operand error (f32: 2^-24) on O(1..10) outputs
    if algo == "f16x2" and mode != 2:
        tol += 4e-3  # one fp16 ulp of an O(1..8) output (the reference value may round the other way)
    assert _report(f"conv[{algo}] {B}x{H}x{W} {Cin}->{Cout} mode{mode}", out.reshape(want_rows.shape), want_rows) < tol


@pytest.mark.parametrize("B,H,W,Cin,Cout,mode", [(16, 250, 16, 128, 256, 0), (16, 250, 16, 256, 256, 1), (34, 125, 8, 256, 512, 1)])
def test_conv3x3_wino1d_256_channel_workgroups(K, B, H, W, Cin, Cout, mode):
    """Launches with >= 1024 workgroups of the 128-channel wide form run as 256-channel workgroups (512 threads, one staging
    of a pixel tile for eight channel groups): same per-channel arithmetic, so the result must equal the CPU reference like
    the small shapes of test_conv3x3_bn_relu do - and, bit for bit, a launch too small to take that form (the first clips
    alone)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(B + W + Cin)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * math.sqrt(2.0 / (9 * Cin))
    sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
    y = F.relu(F.conv2d(x, w, padding=1) * sc[None, :, None, None] + sh[None, :, None, None])
    Hp = H + 1 + ((H + 1) % 2)
    want = _to_rows(F.avg_pool2d(y, 2) if mode == 1 else y, Hp // 2 if mode == 1 else Hp)
    shape = (B * Hp // 2, W // 2, Cout) if mode == 1 else (B * Hp, W, Cout)
    xr, wp = _to_rows(x, Hp).cuda(), K.pack_conv_weight_wino1d_frag(w.cuda())
    out = torch.full(shape, 7.0).cuda()
    K.conv3x3_bn_relu_wino1d(xr, wp, sc.cuda(), sh.cuda(), out, B, Hp, H, W, Cin, Cout, mode)
    tol = 1e-3 * max(1.0, math.sqrt(9 * Cin / 576))
    assert _report(f"conv[wino1d, 256-channel workgroups] {B}x{H}x{W} {Cin}->{Cout} mode{mode}", out.reshape(want.shape), want) < tol
    nb = 2                                                     # two clips: the 128-channel form
    few = torch.full((nb * shape[0] // B,) + shape[1:], 7.0).cuda()
    K.conv3x3_bn_relu_wino1d(xr[:nb * Hp].contiguous(), wp, sc.cuda(), sh.cuda(), few, nb, Hp, H, W, Cin, Cout, mode)
    assert torch.equal(few, out[:few.shape[0]])


@pytest.mark.parametrize("W,Cin,Cout,mode,block,conv", [(64, 64, 64, 1, 1, 2), (16, 128, 256, 0, 3, 1), (8, 512, 512, 1, 4, 2), (4, 512, 1024, 0, 5, 1),
                                                       (2, 1024, 2048, 0, 6, 1), (2, 2048, 2048, 2, 6, 2)])
def test_conv3x3_wino1d_dead_row_skipping(K, W, Cin, Cout, mode, block, conv):
    """Per-clip dead rows: output rows below ``mul * frames[b] + add`` are bit-identical to the full convolution, workgroups
    wholly beyond it store zeros, and at least one workgroup is skipped on this ragged set."""
    from audiocaption_amd.cnn_encoder import rows_needed
    g = torch.Generator().manual_seed(W * 7 + Cin)
    # clips long enough for whole workgroups (up to 128 rows) to fall between a short clip's need and the next clip
    frames = torch.tensor([90, 4, 50, 2, 30], dtype=torch.int32) * (2 if block == 6 else 1)
    B = len(frames)
    H = 90 << (6 - block) if block < 6 else 218
    Hp = H + 1 + ((H + 1) % 2)
    x = torch.zeros(B, Hp, W, Cin)
    x[:, :H] = torch.randn(B, H, W, Cin, generator=g)
    x = x.reshape(B * Hp, W, Cin).cuda()
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * math.sqrt(2.0 / (9 * Cin))
    wp = K.pack_conv_weight_wino1d_frag(w.cuda())
    sc, sh = (torch.rand(Cout, generator=g) + 0.5).cuda(), (torch.randn(Cout, generator=g) * 0.1).cuda()
    shape = (B * Hp, W, Cout) if mode == 0 else ((B * Hp // 2, W // 2, Cout) if mode == 1 else (B, H, Cout))
    full, skip = torch.full(shape, 7.0).cuda(), torch.full(shape, 7.0).cuda()
    K.conv3x3_bn_relu_wino1d(x, wp, sc, sh, full, B, Hp, H, W, Cin, Cout, mode)
    mul, add = rows_needed(block, conv)
    K.conv3x3_bn_relu_wino1d(x, wp, sc, sh, skip, B, Hp, H, W, Cin, Cout, mode, need=(frames.cuda(), mul, add))
    rows = Hp if mode == 0 else (Hp // 2 if mode == 1 else H)
    f, s_ = full.reshape(B, rows, -1).cpu(), skip.reshape(B, rows, -1).cpu()
    zeroed = 0
    for b in range(B):
        need = min(int(mul * frames[b] + add), H)
        n = need if mode != 1 else need // 2
        assert torch.equal(f[b, :n], s_[b, :n]), (b, n)
        tail = s_[b, n:]
        same = (tail == f[b, n:]).all(dim=1)
        zero = (tail == 0).all(dim=1)
        assert bool((same | zero).all())          # a row beyond the need is either computed as usual or a skipped block's zeros
        zeroed += int((zero & ~same).sum())
    assert zeroed > 0


@pytest.mark.parametrize("B,H,W,Cin,Cout,mode", [(1, 31, 2, 2048, 2048, 2), (2, 31, 2, 1024, 2048, 0), (1, 62, 4, 1024, 1024, 1),
                                                  (1, 62, 4, 512, 1024, 0), (1, 125, 8, 512, 512, 1), (3, 9, 8, 256, 512, 1)])
def test_conv3x3_wino1d_k_sliced_launch(K, B, H, W, Cin, Cout, mode):
    """Single clips: a layer of a few workgroups runs K-sliced (slices of the channel loop on separate workgroups, a second
    kernel adds them in order and applies the epilogue).  Same result as the one-slice launch up to the order of the
    channel sum, identical from run to run, and the geometry that fills the chip on its own is not split."""
    g = torch.Generator().manual_seed(B * 100 + W)
    Hp = H + 1 + ((H + 1) % 2)
    x = torch.zeros(B, Hp, W, Cin)
    x[:, :H] = torch.randn(B, H, W, Cin, generator=g)
    x = x.reshape(B * Hp, W, Cin).cuda()
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * math.sqrt(2.0 / (9 * Cin))
    wp = K.pack_conv_weight_wino1d_frag(w.cuda())
    sc, sh = (torch.rand(Cout, generator=g) + 0.5).cuda(), (torch.randn(Cout, generator=g) * 0.1).cuda()
    shape = (B * Hp, W, Cout) if mode == 0 else ((B * Hp // 2, W // 2, Cout) if mode == 1 else (B, H, Cout))
    one = torch.full(shape, 7.0).cuda()
    K.conv3x3_bn_relu_wino1d(x, wp, sc, sh, one, B, Hp, H, W, Cin, Cout, mode)
    floats = K.wino1d_splitk_floats(B, Hp, W, Cin, Cout)
    assert floats > 0 and floats % (B * Hp * W * Cout) == 0 and floats // (B * Hp * W * Cout) >= 2
    ws = torch.full((floats,), float("nan")).cuda()
    a, b = torch.full(shape, 7.0).cuda(), torch.full(shape, 7.0).cuda()
    K.conv3x3_bn_relu_wino1d(x, wp, sc, sh, a, B, Hp, H, W, Cin, Cout, mode, workspace=ws)
    ws.fill_(float("nan"))
    K.conv3x3_bn_relu_wino1d(x, wp, sc, sh, b, B, Hp, H, W, Cin, Cout, mode, workspace=ws)
    assert torch.equal(a, b)
    assert _report(f"k-sliced conv {B}x{H}x{W} {Cin}->{Cout} mode{mode}", a, one) < 2e-5
    assert K.wino1d_splitk_floats(64, 32, 2, 2048, 2048) == 0      # 16 row blocks x 16 channel tiles: not split


@pytest.mark.parametrize("B,H,W,Cin,Cout,mode", [(2, 29, 64, 64, 64, 1), (3, 21, 16, 128, 256, 1), (4, 7, 2, 1024, 2048, 0),
                                                  (2, 13, 4, 512, 1024, 1)])
def test_conv3x3_wino1d_dropout_in_the_epilogue(K, B, H, W, Cin, Cout, mode):
    """Train-mode forward of the frozen network: F.dropout on a block's output applied inside the conv kernel's epilogue
    equals, bit for bit, the layer followed by the counter-hash dropout pass over its output buffer (same seed, same
    element indices) - with and without the device-side step seed."""
    g = torch.Generator().manual_seed(B * 31 + W)
    Hp = H + 1 + ((H + 1) % 2)
    x = torch.zeros(B, Hp, W, Cin)
    x[:, :H] = torch.randn(B, H, W, Cin, generator=g)
    x = x.reshape(B * Hp, W, Cin).cuda()
    wp = K.pack_conv_weight_wino1d_frag((torch.randn(Cout, Cin, 3, 3, generator=g) * math.sqrt(2.0 / (9 * Cin))).cuda())
    sc, sh = (torch.rand(Cout, generator=g) + 0.5).cuda(), (torch.randn(Cout, generator=g) * 0.1).cuda()
    shape = (B * Hp, W, Cout) if mode == 0 else (B * Hp // 2, W // 2, Cout)
    step = torch.tensor([5], dtype=torch.int64, device="cuda")
    for seed_dev in (None, step.data_ptr()):
        two, one = torch.full(shape, 7.0).cuda(), torch.full(shape, 7.0).cuda()
        K.conv3x3_bn_relu_wino1d(x, wp, sc, sh, two, B, Hp, H, W, Cin, Cout, mode)
        K.dropout_(two, two.numel(), 0.2, 1234, seed_dev)
        K.conv3x3_bn_relu_wino1d(x, wp, sc, sh, one, B, Hp, H, W, Cin, Cout, mode, dropout=(0.2, 1234, seed_dev))
        assert torch.equal(one, two)
        kept = float((one != 0).float().mean()) / max(float((two != 0).float().mean()), 1e-9)
        assert kept == 1.0 and 0.1 < float((one == 0).float().mean()) < 0.9


def test_conv3x3_first(K):
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(3)
    B, H, W = 3, 37, 64
    x = torch.randn(B, 1, H, W, generator=g)
    w = torch.randn(64, 1, 3, 3, generator=g) * 0.5
    sc, sh = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1
    y = F.relu(F.conv2d(x, w, padding=1) * sc[None, :, None, None] + sh[None, :, None, None])
    Hp = 40
    out = torch.full((B * Hp, W, 64), 7.0).cuda()
    K.conv3x3_first(_to_rows(x, Hp).reshape(B * Hp, W).cuda(), w.reshape(64, 9).cuda(), sc.cuda(), sh.cuda(), out,
                    B, Hp, H, W)
    assert _report("conv_first", out, _to_rows(y, Hp)) < 1e-5


@pytest.mark.parametrize("gru_algo", ["split", "single"])
def test_gru_layer_and_pooling_vs_oracle(K, state4981, gru_algo):
    """Both recurrence kernels: "split" (default: a (clip, direction) on four 256-thread workgroups, W_hh register resident,
    hidden quarters traded through L2 every step) and "single" (one workgroup, 44 % of W_hh re-streamed per step)."""
    from oracle import cpu_path as O
    import audiocaption_amd as A
    g = torch.Generator().manual_seed(11)
    x = torch.randn(5, 31, 2048, generator=g)
    lens = [31, 20, 7, 1, 25]
    want = O.gru_forward(state4981, x, lens)
    rnn = A.RnnEncoder(-1, 2048, 2048, bidirectional=True, hidden_size=256, dropout=0.5, num_layers=3)
    rnn.load_state_dict({k[len("encoder.rnn."):]: v for k, v in state4981.items() if k.startswith("encoder.rnn.")})
    rnn = rnn.eval().cuda()
    rnn.gru_algo = gru_algo
    got = rnn({"attn": x.cuda(), "attn_len": torch.tensor(lens)})
    if gru_algo == "split":
        assert int(got["gru_error"].item()) == 0
    assert _report("gru attn_emb", got["attn_emb"], want["attn_emb"]) < 2e-5
    assert _report("gru fc_emb", got["fc_emb"], want["fc_emb"]) < 2e-5
    assert torch.equal(got["attn_emb_len"], torch.tensor(lens))
    # truncation to max(len) (pad_packed_sequence) and zeros at padded steps
    got2 = rnn({"attn": x.cuda(), "attn_len": torch.tensor([9, 20, 7, 1, 12])})
    assert got2["attn_emb"].shape == (5, 20, 512)
    assert float(got2["attn_emb"][0, 9:].abs().max()) == 0.0


def test_gru_split_kernel_under_uneven_load(K):
    """The split recurrence trades data between workgroups inside one launch (8-byte {tag, value} granules, agent-scope
    relaxed atomics).  Such hand-offs have to be tested with the consumer's L1 warm and the chip unevenly loaded: a
    second stream streams memory while 130 clips x 93 steps (1040 workgroups: more than the 256 CUs hold, so partners
    start at different times) run, five times over the same workspace; every word against the single-workgroup kernel."""
    g = torch.Generator().manual_seed(5)
    B, T, Hh = 130, 93, 256
    gx = torch.randn(B * T, 6 * Hh, generator=g).cuda()
    whh = (torch.randn(2, 3 * Hh, Hh, generator=g) / 16).cuda()
    bhh = torch.randn(2, 3 * Hh, generator=g).cuda()
    lens = torch.randint(1, T + 1, (B,), generator=g).to(torch.int32).cuda()
    want = K.gru_layer(gx, K.gru_pack_whh(whh, Hh), bhh, lens, B, T, Hh)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    junk = torch.randn(64 << 20, device="cuda")
    ws = None
    for rep in range(5):
        with torch.cuda.stream(side):
            for _ in range(6):
                junk.mul_(1.0001)
        got, ws = K.gru_layer_split(gx, whh, bhh, lens, B, T, Hh, ws)
        torch.cuda.synchronize()
        assert int(K.gru_split_error(ws, B).item()) == 0
        assert float((got - want).abs().max()) < 1e-5, rep
    for b in range(B):   # zeros at the padded steps
        assert float(got[b, int(lens[b]):].abs().max() if int(lens[b]) < T else 0.0) == 0.0


def test_conv3x3_wino1d_beyond_2gib_runs_in_clip_chunks(K):
    """The F(2,3) kernel addresses its input with 32-bit byte offsets (one buffer descriptor): an input of 2 GiB or more -
    conv2 of block 1 from 128 ten-second clips - is REJECTED by the C ABI, and the tier's launcher convolves the batch in
    clip chunks instead.  160 clips: the clips on both sides of the chunk boundary equal the same clips convolved alone."""
    from audiocaption_amd import cnn_encoder as CE
    from audiocaption_amd._lib import HipLibraryError
    B, H, Hp, W, C = 160, 1001, 1024, 64, 64
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(B, Hp, W, C, device="cuda", generator=g)
    x[:, H:] = 0
    x = x.reshape(B * Hp, W, C)
    wp = K.pack_conv_weight_wino1d_frag(torch.randn(C, C, 3, 3, device="cuda", generator=g) * math.sqrt(2.0 / (9 * C)))
    sc, sh = torch.rand(C, device="cuda", generator=g) + 0.5, torch.randn(C, device="cuda", generator=g) * 0.1
    out = torch.full((B * Hp // 2, W // 2, C), 7.0, device="cuda")
    with pytest.raises(HipLibraryError):
        K.conv3x3_bn_relu_wino1d(x, wp, sc, sh, out, B, Hp, H, W, C, C, 1)
    chunk = CE._wino1d_clip_chunk(B, Hp, W, C)
    assert 100 < chunk < 128
    CE._conv_wino1d(x, wp, sc, sh, out, B, Hp, H, W, C, C, 1)
    one = torch.empty(Hp // 2, W // 2, C, device="cuda")
    for b in (0, chunk - 1, chunk, B - 1):
        K.conv3x3_bn_relu_wino1d(x[b * Hp:(b + 1) * Hp], wp, sc, sh, one, 1, Hp, H, W, C, C, 1)
        assert torch.equal(one, out[b * Hp // 2:(b + 1) * Hp // 2]), b
    assert float(out.abs().max()) < 50 and float(out[:Hp // 2].abs().mean()) > 0.05
