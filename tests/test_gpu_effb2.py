"""GPU tests of the EfficientNet-B2 encoder path through the C ABI against
oracle/effb2_path.py and against the independent-witness fixtures tests/golden/g10_logmel.npz / g11_effb2.npz
(transformers.audio_utils / transformers.EfficientNetModel, tests/golden/make_witness.py): the oracle restates the
published efficientnet_pytorch / torchaudio algorithms, which the reference does not vendor."""
import ctypes
import os
import subprocess
import sys

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from audiocaption_amd import _lib, build
    build.build()
    return _lib.load()


@pytest.fixture(scope="module")
def effb2_model(state_effb2):
    import audiocaption_amd as A
    from audiocaption_amd import build
    build.build()
    model = A.init_model_from_config(A.effb2_trm_config(4981), print_fn=lambda s: None)
    model.load_state_dict(state_effb2, strict=True)
    return model.eval().to("cuda:0")


def S():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_KEEP = []   # P() only takes an address: temporaries such as ``x.cuda()`` are kept alive until the next test starts


@pytest.fixture(autouse=True)
def _drop_kept_tensors():
    _KEEP.clear()
    yield
    _KEEP.clear()


def P(t):
    if t is None:
        return None
    _KEEP.append(t)
    return ctypes.c_void_p(t.data_ptr())


def rel(name, got, want):
    got, want = torch.as_tensor(got).detach().double().cpu(), torch.as_tensor(want).detach().double().cpu()
    d = float((got - want).abs().max()) / (float(want.abs().max()) + 1e-30)
    print(f"[{name}] max|diff| / max|want| = {d:.3e}")
    return d


def test_stem_kernel(lib):
    g = torch.Generator().manual_seed(0)
    B, T, Fm, C = 2, 37, 64, 32
    x = torch.randn(B, T, Fm, generator=g)
    w = torch.randn(C, 1, 3, 3, generator=g)
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    ref = F.conv2d(F.pad(x.transpose(1, 2).unsqueeze(1), (0, 1, 0, 1)), w, stride=2)       # (B, C, F', T')
    ref = ref * sc[None, :, None, None] + sh[None, :, None, None]
    ref = (ref * torch.sigmoid(ref)).permute(0, 3, 2, 1)                                    # [B][T'][F'][C]
    y = torch.empty(ref.shape, device="cuda")
    assert lib.ac_effnet_stem(P(x.cuda()), P(w.reshape(C, 9).contiguous().cuda()), P(sc.cuda()), P(sh.cuda()), P(y), B, T,
                              Fm, C, 0, 1, S()) == 0
    assert rel("stem", y, ref) < 1e-5


@pytest.mark.parametrize("k,stride,pad,C", [(3, 1, (1, 1), 48), (3, 2, (0, 1), 48), (5, 2, (2, 2), 48), (5, 1, (2, 2), 48),
                                            (3, 2, (1, 1), 48), (5, 1, (2, 2), 1248), (3, 1, (1, 1), 2112)])
def test_depthwise_kernel_and_squeeze_sums(lib, k, stride, pad, C):
    """C = 1248 / 2112: more than 256 channel quads per position (a thread then walks several channel groups)."""
    g = torch.Generator().manual_seed(k * 10 + stride)
    B, T, Fm = 3, 21, 10
    x = torch.randn(B, T, Fm, C, generator=g)
    w = torch.randn(C, 1, k, k, generator=g) * 0.3          # [C][1][k mel][k time]
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    xn = x.permute(0, 3, 2, 1)                               # (B, C, F, T)
    ref = F.conv2d(F.pad(xn, (pad[0], pad[1], pad[0], pad[1])), w, stride=stride, groups=C)
    ref = ref * sc[None, :, None, None] + sh[None, :, None, None]
    ref = ref * torch.sigmoid(ref)
    ref_cl = ref.permute(0, 3, 2, 1).contiguous()
    y = torch.empty(ref_cl.shape, device="cuda")
    pool = torch.zeros(B, C, device="cuda")
    wp = w[:, 0].permute(2, 1, 0).contiguous().cuda()        # [k time][k mel][C]
    assert lib.ac_effnet_depthwise(P(x.cuda()), P(wp), P(sc.cuda()), P(sh.cuda()), P(y), P(pool), 1.0, B, T, Fm, C, k, stride,
                                   pad[0], pad[1], S()) == 0
    assert rel(f"depthwise k{k} s{stride}", y, ref_cl) < 1e-5
    assert rel("squeeze sums", pool, ref.sum(dim=(2, 3))) < 1e-5


@pytest.mark.parametrize("k,Fm,T,C,B", [(5, 4, 63, 720, 3), (5, 2, 32, 1248, 2), (3, 4, 63, 528, 2), (3, 2, 32, 2112, 2),
                                        (3, 8, 126, 288, 2), (5, 4, 7, 48, 3), (3, 2, 1, 8, 1), (5, 2, 94, 1248, 1), (5, 4, 188, 132, 1),
                                        (5, 8, 126, 288, 2), (3, 16, 251, 144, 2), (5, 8, 5, 16, 1), (3, 16, 17, 260, 1),
                                        (3, 32, 501, 32, 2), (3, 32, 501, 16, 2), (3, 32, 1, 32, 1), (3, 32, 70, 16, 3)])
def test_depthwise_rows_in_registers_form(lib, k, Fm, T, C, B, monkeypatch):
    """Stride 1 on a narrow mel axis (the 63 x 4 and 32 x 2 stages; 188 / 94 rows: 30 s clips): the form that keeps the K x F
    window in registers and slides along time (csrc/effnet.hip depthwise_rows_kernel) against F.conv2d, chunk borders
    (partial last chunk, a single row), channel counts that leave lanes idle, and the squeeze sums.  F = 32 with 32 / 16
    channels (the first stage): the row-segment form, a wave per row (depthwise_rowseg_kernel)."""
    g = torch.Generator().manual_seed(k * 100 + Fm * 10 + T)
    x = torch.randn(B, T, Fm, C, generator=g)
    w = torch.randn(C, 1, k, k, generator=g) * 0.3
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    pad = (k - 1) // 2
    ref = F.conv2d(F.pad(x.permute(0, 3, 2, 1), (pad, pad, pad, pad)), w, stride=1, groups=C)
    ref = ref * sc[None, :, None, None] + sh[None, :, None, None]
    ref = ref * torch.sigmoid(ref)
    ref_cl = ref.permute(0, 3, 2, 1).contiguous()
    y = torch.full(ref_cl.shape, 7.0, device="cuda")
    pool = torch.zeros(B, C, device="cuda")
    wp = w[:, 0].permute(2, 1, 0).contiguous().cuda()
    assert lib.ac_effnet_depthwise(P(x.cuda()), P(wp), P(sc.cuda()), P(sh.cuda()), P(y), P(pool), 0.5, B, T, Fm, C, k, 1, pad, pad,
                                   S()) == 0
    assert rel(f"depthwise rows form k{k} F{Fm} T{T} C{C}", y, ref_cl) < 1e-5
    assert rel("squeeze sums", pool, 0.5 * ref.sum(dim=(2, 3))) < 1e-5


@pytest.mark.parametrize("k,stride,pad,Fm,cin,mid", [(3, 1, (1, 1), 16, 24, 144), (3, 2, (0, 1), 32, 16, 96),
                                                     (5, 2, (2, 2), 10, 24, 144), (5, 1, (2, 2), 4, 88, 528),
                                                     (3, 2, (1, 1), 8, 48, 288), (5, 1, (2, 2), 2, 208, 1248)])
@pytest.mark.parametrize("lds_kb", [None, "40"])
def test_fused_expand_depthwise_vs_torch(lib, k, stride, pad, Fm, cin, mid, lds_kb):
    """ac_effnet_expand_depthwise (expand 1x1 + BN + swish -> depthwise + BN + swish -> squeeze sums in one kernel, the
    expanded tensor in LDS) against F.conv2d on the CPU: band seams (several bands per clip), partial last band, partial
    32-channel chunk (144, 528), the zero padding of the EXPANDED tensor (swish(bias) != 0 must not leak into it)."""
    g = torch.Generator().manual_seed(k * 100 + stride * 10 + Fm)
    B, T = 3, 45
    x = torch.randn(B, T, Fm, cin, generator=g)
    we = torch.randn(mid, cin, generator=g) / cin ** 0.5
    be = torch.randn(mid, generator=g) * 0.5 + 0.3
    w = torch.randn(mid, 1, k, k, generator=g) * 0.3          # [C][1][k mel][k time]
    sc, sh = torch.rand(mid, generator=g) + 0.5, torch.randn(mid, generator=g) * 0.1
    e = F.linear(x, we, be)
    e = e * torch.sigmoid(e)                                  # [B][T][F][mid]
    en = e.permute(0, 3, 2, 1)                                # (B, C, F, T)
    ref = F.conv2d(F.pad(en, (pad[0], pad[1], pad[0], pad[1])), w, stride=stride, groups=mid)
    ref = ref * sc[None, :, None, None] + sh[None, :, None, None]
    ref = ref * torch.sigmoid(ref)
    ref_cl = ref.permute(0, 3, 2, 1).contiguous()
    wp = w[:, 0].permute(2, 1, 0).contiguous().cuda()        # [k time][k mel][C]
    dev_in = [t.cuda() for t in (x, we, be)] + [wp, sc.cuda(), sh.cuda()]     # kept alive: P() only takes addresses
    args = tuple(P(t) for t in dev_in)
    if lds_kb is not None:
        # the LDS budget is read once per process: the small-band geometry (many band seams) runs in a child process
        child = subprocess.run([sys.executable, "-c", _FUSED_CHILD, str(k), str(stride), str(pad[0]), str(pad[1]), str(Fm),
                                str(cin), str(mid)], env=dict(os.environ, AUDIOCAPTION_EDW_LDS_KB=lds_kb),
                               capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        print(child.stdout[-2000:], child.stderr[-2000:])
        assert child.returncode == 0
        return
    y = torch.full(ref_cl.shape, float("nan"), device="cuda")
    pool = torch.zeros(B, mid, device="cuda")
    assert lib.ac_effnet_expand_depthwise(*args, P(y), P(pool), 1.0, B, T, Fm, cin, mid, k, stride, pad[0], pad[1], S()) == 0
    assert rel(f"fused expand+depthwise k{k} s{stride} F{Fm}", y, ref_cl) < 1e-5
    assert rel("squeeze sums", pool, ref.sum(dim=(2, 3))) < 1e-5
    pool2 = torch.zeros(B, mid, device="cuda")               # sums only
    assert lib.ac_effnet_expand_depthwise(*args, None, P(pool2), 1.0, B, T, Fm, cin, mid, k, stride, pad[0], pad[1], S()) == 0
    assert rel("squeeze sums (y = NULL)", pool2, ref.sum(dim=(2, 3))) < 1e-5


_FUSED_CHILD = r"""
import ctypes, sys, torch, torch.nn.functional as F
from audiocaption_amd import _lib, build
build.build(); lib = _lib.load()
k, stride, p0, p1, Fm, cin, mid = map(int, sys.argv[1:8])
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
g = torch.Generator().manual_seed(7)
B, T = 2, 45
x = torch.randn(B, T, Fm, cin, generator=g)
we = torch.randn(mid, cin, generator=g) / cin ** 0.5
be = torch.randn(mid, generator=g) * 0.5 + 0.3
w = torch.randn(mid, 1, k, k, generator=g) * 0.3
sc, sh = torch.rand(mid, generator=g) + 0.5, torch.randn(mid, generator=g) * 0.1
e = F.linear(x, we, be); e = e * torch.sigmoid(e)
ref = F.conv2d(F.pad(e.permute(0, 3, 2, 1), (p0, p1, p0, p1)), w, stride=stride, groups=mid)
ref = ref * sc[None, :, None, None] + sh[None, :, None, None]
ref = ref * torch.sigmoid(ref)
ref_cl = ref.permute(0, 3, 2, 1).contiguous()
y = torch.full(ref_cl.shape, float("nan"), device="cuda")
pool = torch.zeros(B, mid, device="cuda")
dev_in = [t.cuda() for t in (x, we, be, w[:, 0].permute(2, 1, 0).contiguous(), sc, sh)]   # kept alive: P() takes addresses
rc = lib.ac_effnet_expand_depthwise(*[P(t) for t in dev_in], P(y), P(pool), 1.0, B, T, Fm, cin, mid, k, stride, p0, p1,
                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
if rc == -1:
    print("band does not fit this LDS budget: AC_ERR_ARG (the caller keeps the two-kernel chain)"); sys.exit(0)
assert rc == 0, rc
d = float((y.cpu() - ref_cl).abs().max()) / float(ref_cl.abs().max())
dp = float((pool.cpu() - ref.sum(dim=(2, 3))).abs().max()) / float(ref.sum(dim=(2, 3)).abs().max())
print("small-band geometry: y", d, "pool", dp)
assert d < 1e-5 and dp < 1e-5
"""


def test_se_gate_and_gated_projection(lib):
    g = torch.Generator().manual_seed(3)
    B, HW, C, Sq, Co = 3, 35, 96, 4, 24
    x = torch.randn(B * HW, C, generator=g)
    pool = x.view(B, HW, C).sum(1)
    w1, b1 = torch.randn(Sq, C, generator=g) * 0.1, torch.randn(Sq, generator=g) * 0.1
    w2, b2 = torch.randn(C, Sq, generator=g) * 0.5, torch.randn(C, generator=g) * 0.1
    sq = F.linear(pool / HW, w1, b1)
    gate_ref = torch.sigmoid(F.linear(sq * torch.sigmoid(sq), w2, b2))
    gate = torch.empty(B, C, device="cuda")
    assert lib.ac_effnet_se_gate(P(pool.cuda()), 1.0 / HW, P(w1.cuda()), P(b1.cuda()), P(w2.cuda()), P(b2.cuda()), P(gate),
                                 B, C, Sq, S()) == 0
    assert rel("se gate", gate, gate_ref) < 1e-5
    # the transposed-weight kernel the encoder uses (one launch per block), also at the widths of the late blocks
    gate_t = torch.empty(B, C, device="cuda")
    dev_in = [t.cuda() for t in (pool, w1, b1, w2.t().contiguous(), b2)]
    assert lib.ac_effnet_se_gate_t(P(dev_in[0]), 1.0 / HW, P(dev_in[1]), P(dev_in[2]), P(dev_in[3]), P(dev_in[4]), P(gate_t),
                                   B, C, Sq, S()) == 0
    assert rel("se gate (w2 transposed)", gate_t, gate_ref) < 1e-5
    for Cb, Sb in ((1248, 52), (2112, 88), (16, 4), (528, 22)):
        pb_ = torch.randn(5, Cb, generator=g)
        w1b, b1b = torch.randn(Sb, Cb, generator=g) / Cb ** 0.5, torch.randn(Sb, generator=g) * 0.1
        w2b, b2b = torch.randn(Cb, Sb, generator=g) * 0.3, torch.randn(Cb, generator=g) * 0.1
        sqb = F.linear(pb_ * 0.25, w1b, b1b)
        want = torch.sigmoid(F.linear(sqb * torch.sigmoid(sqb), w2b, b2b))
        got = torch.empty(5, Cb, device="cuda")
        dev_b = [t.cuda() for t in (pb_, w1b, b1b, w2b.t().contiguous(), b2b)]
        assert lib.ac_effnet_se_gate_t(P(dev_b[0]), 0.25, P(dev_b[1]), P(dev_b[2]), P(dev_b[3]), P(dev_b[4]), P(got), 5, Cb,
                                       Sb, S()) == 0
        assert rel(f"se gate C={Cb} S={Sb}", got, want) < 1e-5
    # 1x1 projection of the gated tensor with a residual: y = res + (x * gate) W^T + b
    w, b = torch.randn(Co, C, generator=g) * 0.1, torch.randn(Co, generator=g)
    res = torch.randn(B * HW, Co, generator=g)
    ref = res + F.linear((x.view(B, HW, C) * gate_ref[:, None]).reshape(-1, C), w, b)
    y = res.clone().cuda()
    assert lib.ac_gemm(P(x.cuda()), C, 1, P(w.cuda()), 1, C, P(y), Co, B * HW, Co, C, P(b.cuda()), 0, 1.0, 1, 0.0, 0, None,
                       0, P(gate), HW, S()) == 0
    assert rel("gated projection + residual", y, ref) < 1e-5
    y = res.clone().cuda()
    assert lib.ac_pointwise_conv(P(x.cuda()), P(w.cuda()), P(b.cuda()), P(y), B * HW, Co, C, 0, 1.0, P(gate), HW, S()) == 0
    assert rel("pointwise kernel: gated projection + residual", y, ref) < 1e-5
    for n_out in (16, 48, 88, 200):   # 1..4 accumulator tiles per pass, ragged last tile, several passes
        wn, bn = torch.randn(n_out, 88, generator=g) * 0.1, torch.randn(n_out, generator=g)
        r3 = F.linear(x[:, :88], wn, bn)
        r3 = r3 * torch.sigmoid(r3)
        y3 = torch.empty(B * HW, n_out, device="cuda")
        assert lib.ac_pointwise_conv(P(x[:, :88].contiguous().cuda()), P(wn.cuda()), P(bn.cuda()), P(y3), B * HW, n_out, 88,
                                     2, 0.0, None, 0, S()) == 0
        assert rel(f"pointwise kernel N={n_out} swish", y3, r3) < 1e-5
    # swish epilogue, K not a multiple of 16
    ref2 = F.linear(x[:, :88], w[:, :88].contiguous(), b)
    ref2 = ref2 * torch.sigmoid(ref2)
    y2 = torch.empty(B * HW, Co, device="cuda")
    xs = x[:, :88].contiguous().cuda()
    assert lib.ac_gemm(P(xs), 88, 1, P(w[:, :88].contiguous().cuda()), 1, 88, P(y2), Co, B * HW, Co, 88, P(b.cuda()), 2, 0.0,
                       1, 0.0, 0, None, 0, None, 0, S()) == 0
    assert rel("swish epilogue", y2, ref2) < 1e-5


@pytest.mark.parametrize("M,N,K", [(8192, 1248, 208), (8200, 208, 1248), (4097, 352, 2112), (1344, 768, 256), (300, 88, 528),
                                   (32256, 120, 720), (70, 1408, 352), (1344, 256, 1024), (2000, 120, 720)])
def test_pw_gemm_bf16x3_vs_float64(lib, M, N, K):
    """ac_pw_gemm_bf16x3 (activation-stationary split-bf16 1x1 convolution over weights pre-split in MFMA fragment order,
    csrc/pw_gemm.hip) against float64 on EfficientNet-B2's matrix-bound shapes: the three tile heights, K and N that are
    not multiples of 32 (zero-padded fragments), column groups with one and two tiles per wave, row tails, the 128-k
    variant for long K with few workgroups (K = 528 and 720: partial last chunk); plain, swish,
    and the squeeze-excite form y = res + (x .* gate[clip]) w^T + b.  2^-16 relative operand error: 3e-5 of the
    largest output (the bar of ac_gemm_bf16x3)."""
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    xd, wd, bd = x.cuda(), w.cuda(), b.cuda()
    wfrag = torch.empty(lib.ac_pw_gemm_packed_bytes(N, K), device="cuda", dtype=torch.uint8)
    assert lib.ac_pw_gemm_pack(P(wd), P(wfrag), N, K, S()) == 0
    y = torch.full((M, N), float("nan"), device="cuda")
    assert lib.ac_pw_gemm_bf16x3(P(xd), P(wfrag), P(bd), P(y), M, N, K, 0, 0.0, None, 0, S()) == 0
    ref = x.double() @ w.double().t() + b.double()
    assert rel("pw gemm: x w^T + b", y, ref) < 3e-5
    assert lib.ac_pw_gemm_bf16x3(P(xd), P(wfrag), P(bd), P(y), M, N, K, 2, 0.0, None, 0, S()) == 0
    assert rel("pw gemm: swish", y, ref * torch.sigmoid(ref)) < 3e-5
    rows_per = 64
    gate = torch.rand((M + rows_per - 1) // rows_per, K, generator=g)
    res = torch.randn(M, N, generator=g)
    yg, gd = res.clone().cuda(), gate.cuda()
    assert lib.ac_pw_gemm_bf16x3(P(xd), P(wfrag), P(bd), P(yg), M, N, K, 0, 1.0, P(gd), rows_per, S()) == 0
    xg = x.double() * gate.double().repeat_interleave(rows_per, 0)[:M]
    assert rel("pw gemm: gated x w^T + b + res", yg, res.double() + xg @ w.double().t() + b.double()) < 3e-5


def test_logmel_htk_top_db_vs_oracle(effb2_model):
    from audiocaption_amd import procedural as Pr
    from oracle import effb2_path as E
    wav = torch.from_numpy(Pr.synthetic_wav(3, 48000, sample_rate=16000, varied=True))
    wav[2] *= 1e-4                                            # a nearly silent clip: its floor is set by the batch maximum
    want = E.logmel_effb2(wav)                                # (B, 64, T)
    enc = effb2_model.encoder
    got = enc.logmel(wav.cuda()).view(3, -1, 64).transpose(1, 2).cpu()
    d = float((got - want).abs().max())
    print(f"log-mel (HTK, top_db 120) max|diff| {d:.3e} dB; floor {float(want.min()):.2f} dB")
    assert d < 5e-3
    assert float(got.min()) == pytest.approx(float(want.max()) - 120.0, abs=1e-3)


def test_backbone_from_logmel_vs_oracle(effb2_model, state_effb2):
    from audiocaption_amd import procedural as Pr
    from oracle import effb2_path as E
    wav = torch.from_numpy(Pr.synthetic_wav(2, 100000, sample_rate=16000, varied=True))
    lms = E.logmel_effb2(wav)                                 # (B, 64, T) from the oracle: the backbone in isolation
    want = E.effb2_from_logmel(state_effb2, lms)
    x = lms.transpose(1, 2).contiguous().cuda()               # [B][T][64]
    got = effb2_model.encoder.features(x, 2, lms.shape[2])
    assert got.shape == want.shape
    assert rel("EffB2 attn_emb", got, want) < 2e-4


def test_logmel_htk_top_db_vs_independent_witness(effb2_model, golden_dir):
    """Rows A1 / A2 (EffB2 front-end, hf_wrapper.py:270-279,292-293) against ``g10_logmel.npz``: the float64
    ``transformers.audio_utils`` log-mel (htk scale, no norm, 0-8000 Hz, n_fft 512, hop 160) clamped at the maximum of the
    WHOLE batch - 120 dB; clip 2 is nearly silent and sits on that floor."""
    import numpy as np
    sys.path.insert(0, golden_dir)
    import make_witness as W
    g = np.load(os.path.join(golden_dir, "g10_logmel.npz"))
    _, wav16 = W.logmel_inputs()
    got = effb2_model.encoder.logmel(torch.from_numpy(wav16).cuda()).view(3, -1, 64).transpose(1, 2).cpu().double().numpy()
    want = g["effb2_db"].astype(np.float64)
    d = np.abs(got - want)
    print(f"[log-mel (HTK, top_db 120) vs witness] max|diff| {d.max():.3e} dB, p99 {np.percentile(d, 99):.3e} dB")
    assert d.max() < 8e-3 and np.percentile(d, 99) < 3e-4
    assert float(got.min()) == pytest.approx(float(want.max()) - 120.0, abs=1e-3)


@pytest.mark.parametrize("name", ["lms10", "lms30", "sq260"])
def test_backbone_vs_independent_witness(effb2_model, golden_dir, name):
    """Row A8 against ``g11_effb2.npz``: ``transformers.EfficientNetModel`` (B2: width 1.1, depth 1.2, the 260-px static
    padding chain, one input channel) carrying the SAME procedural weights mapped by name, mean over mel
    (hf_wrapper.py:229-232) - for a 10 s and a 30 s log-mel and a square 260 x 260 input."""
    import numpy as np
    sys.path.insert(0, golden_dir)
    import make_witness as W
    want = torch.from_numpy(np.load(os.path.join(golden_dir, "g11_effb2.npz"))[name])
    lms = torch.from_numpy(W.effb2_inputs()[name])            # (B, F, T)
    B, Fm, T = lms.shape
    x = lms.transpose(1, 2).contiguous().cuda()               # [B][T][F]
    got = effb2_model.encoder.features(x, B, T, Fm)
    assert got.shape == want.shape
    assert rel(f"EffB2 attn_emb vs transformers.EfficientNetModel ({name})", got, want) < 2e-4


@pytest.mark.parametrize("seconds", [10, 4])
def test_effb2_trm_tokens_vs_oracle(effb2_model, state_effb2, seconds):
    """wav -> tokens: greedy and beam-3 through ``model(input_dict)`` and the HF call surface ``model(audio, len)``."""
    from audiocaption_amd import procedural as Pr
    from audiocaption_amd.hf_wrapper import CaptioningConfig, CaptioningModel
    from oracle import effb2_path as E
    L = seconds * 16000
    wav = torch.from_numpy(Pr.synthetic_wav(3, L, sample_rate=16000, varied=True))
    wav_len = [L, int(0.8 * L), int(0.55 * L)]
    want_g = E.caption_forward(state_effb2, wav, wav_len, "greedy", max_length=12)
    inp = {"mode": "inference", "wav": wav.cuda(), "wav_len": wav_len, "specaug": False}
    got_g = effb2_model(dict(inp, sample_method="greedy", max_length=12))
    assert torch.equal(got_g["attn_emb_len"], want_g["attn_emb_len"])
    assert rel("attn_emb", got_g["attn_emb"], want_g["attn_emb"]) < 5e-4
    assert rel("fc_emb", got_g["fc_emb"], want_g["fc_emb"]) < 5e-4
    st = want_g["steps"]
    assert rel("greedy logit", got_g["logit"][:, :st], want_g["logit"][:, :st]) < 5e-4
    top2 = want_g["logit"][:, :st].topk(2, -1).values
    if float((top2[..., 0] - top2[..., 1]).min()) > 1e-3:
        assert torch.equal(got_g["seq"], want_g["seq"])
    want_b = E.caption_forward(state_effb2, wav, wav_len, "beam", beam_size=3, max_length=12)
    hf = CaptioningModel(effb2_model, CaptioningConfig(sample_rate=16000, vocab_size=4981))
    seq = hf(wav, wav_len, max_length=12)                     # defaults: beam search, beam_size 3
    assert seq.device.type == "cpu" and seq.dtype == torch.int64 and seq.shape == (3, 12)
    assert torch.equal(seq, want_b["seq"])


def test_effb2_30s_beam4_vs_oracle(effb2_model, state_effb2):
    """BASELINE configs[4]: EffB2-Trm on 30 s clips (T = 3001 frames -> 94 encoder frames, the last one masked), ragged
    lengths, beam search with beam_size 4 - token ids identical to the oracle's restatement of
    ``EfficientNetB2.forward`` (hf_wrapper.py:287-315) + ``beam_search`` (base.py:254-325), encoder outputs to 5e-4."""
    from audiocaption_amd import procedural as Pr
    from oracle import effb2_path as E
    L = 30 * 16000
    wav = Pr.synthetic_wav(3, L, sample_rate=16000, seed=41, varied=True)
    wav_len = [L, 400000, 250000]
    for i, n in enumerate(wav_len):
        wav[i, n:] = 0.0
    wav = torch.from_numpy(wav)
    want = E.caption_forward(state_effb2, wav, wav_len, "beam", beam_size=4, max_length=20)
    got = effb2_model({"mode": "inference", "wav": wav.cuda(), "wav_len": wav_len, "specaug": False,
                       "sample_method": "beam", "beam_size": 4, "max_length": 20})
    assert got["attn_emb"].shape == (3, 94, 1408) and got["attn_emb_len"].tolist() == [93, 78, 48]
    assert torch.equal(got["attn_emb_len"], want["attn_emb_len"])
    assert rel("30 s attn_emb", got["attn_emb"], want["attn_emb"]) < 5e-4
    assert rel("30 s fc_emb", got["fc_emb"], want["fc_emb"]) < 5e-4
    print(got["seq"].tolist(), want["seq"].tolist())
    assert got["seq"].shape == (3, 20) and got["seq"].dtype == torch.int64 and got["seq"].device.type == "cpu"
    assert torch.equal(got["seq"], want["seq"])
    # the throughput-mode entry (encoder submitted up front, search at result()) returns the same ids
    pend = effb2_model.forward_async({"mode": "inference", "wav": wav.cuda(), "wav_len": wav_len, "specaug": False,
                                      "sample_method": "beam", "beam_size": 4, "max_length": 20})
    assert torch.equal(pend.result()["seq"], want["seq"])


def test_encoder_graph_survives_a_buffer_regrow(effb2_model):
    """The EffB2 encoder replays a HIP graph per input shape over SHARED activation buffers.  10 s, 10 s (capture),
    30 s (the buffers are re-allocated), 10 s: the 10 s graph must be re-captured, not replayed into freed memory."""
    from audiocaption_amd import procedural as Pr
    enc = effb2_model.encoder
    enc._graphs.clear()
    enc._bufs.clear()
    short = torch.from_numpy(Pr.synthetic_wav(2, 160000, sample_rate=16000, seed=3, varied=True)).cuda()
    long_ = torch.from_numpy(Pr.synthetic_wav(2, 480000, sample_rate=16000, seed=4, varied=True)).cuda()
    want_s, want_l = enc._encode(short).clone(), enc._encode(long_).clone()
    enc._bufs.clear()                      # start again from small buffers
    a = enc._encode_graph(short)           # eager
    b = enc._encode_graph(short)           # captured + replayed
    gen = enc._buf_gen
    c = enc._encode_graph(long_)           # larger shape: shared buffers re-allocated
    assert enc._buf_gen > gen
    filler = torch.full((int(2e8),), 7.0, device="cuda")   # recycle whatever the old buffers occupied
    d = enc._encode_graph(short)           # must not replay the stale graph
    e = enc._encode_graph(long_)           # second use of the long shape: captured
    f = enc._encode_graph(short)
    del filler
    for name, got, want in (("a", a, want_s), ("b", b, want_s), ("c", c, want_l), ("d", d, want_s), ("e", e, want_l),
                            ("f", f, want_s)):
        # (the squeeze-excite sums are float atomics, so two runs agree to rounding, not bit for bit)
        assert rel(f"graph vs eager {name}", got, want) < 1e-5, name


def test_effb2_hf_class_on_the_published_layout(state_effb2):
    """``Effb2TrmCaptioningModel`` (hf_wrapper.py:1144-1181) built from ``Effb2TrmConfig`` defaults, weights loaded from a
    state dict in the published layout (``model.model.*`` + the distillation heads), called the way README.md:30-40 does:
    ``model(audio, audio_length)`` -> CPU LongTensor, beam 3 by default - ids identical to the oracle."""
    from audiocaption_amd import procedural as Pr
    from audiocaption_amd.hf_wrapper import Effb2TrmCaptioningModel, Effb2TrmConfig
    from oracle import effb2_path as E
    ck = {"model.model." + k: v for k, v in state_effb2.items()}
    g = torch.Generator().manual_seed(0)
    ck.update({"model.stdnt_proj.weight": torch.randn(1024, 1408, generator=g), "model.stdnt_proj.bias": torch.zeros(1024),
               "model.tchr_proj.weight": torch.randn(1024, 768, generator=g), "model.tchr_proj.bias": torch.zeros(1024),
               "model.logit_scale": torch.tensor(2.66)})
    model = Effb2TrmCaptioningModel(Effb2TrmConfig())
    model.load_checkpoint(ck, strict=True)
    model = model.to("cuda:0").eval()
    assert model.device.type == "cuda" and model.config.sample_rate == 16000
    L = 5 * 16000
    wav = torch.from_numpy(Pr.synthetic_wav(3, L, sample_rate=16000, seed=8, varied=True))
    lens = [L, 60000, 50000]
    want = E.caption_forward(state_effb2, wav, lens, "beam", beam_size=3, max_length=20)
    seq = model(wav, lens)                                   # CPU tensor in, moved by the wrapper (hf_wrapper.py:1170)
    assert seq.device.type == "cpu" and seq.dtype == torch.int64 and seq.shape == (3, 20)
    assert torch.equal(seq, want["seq"])
    want_g = E.caption_forward(state_effb2, wav, lens, "greedy", max_length=10)
    assert torch.equal(model(audio=wav, audio_length=torch.tensor(lens), sample_method="greedy", max_length=10), want_g["seq"])
