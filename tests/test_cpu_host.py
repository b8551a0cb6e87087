"""CPU tests of the host side: the C-ABI library builds, loads and exports every declared symbol (no
compute calls), plugin construction mirrors the reference protocol, and the product path refuses to
run without the GPU instead of falling back."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import audiocaption_amd as A
from audiocaption_amd import _lib, build, config, procedural as P

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib_path():
    return build.build()


def test_library_exports_every_declared_symbol(lib_path):
    header = open(os.path.join(REPO, "include", "audiocaption_hip.h")).read()
    declared = set(re.findall(r"\b(ac_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    lib = ctypes.CDLL(lib_path)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/audiocaption_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), "ctypes signature table out of sync with the header"
    assert lib.ac_abi_version() == _lib.ABI_VERSION == 2


def test_ctypes_signatures_match_the_header_prototypes():
    """Argument count and coarse type (pointer / integer / float) of every prototype vs the ctypes table."""
    header = open(os.path.join(REPO, "include", "audiocaption_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    protos = re.findall(r"\b(?:int|long)\s+(ac_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", header)
    assert len(protos) == len(_lib.SIGNATURES)
    for name, args in protos:
        args = [a.strip() for a in args.split(",") if a.strip() and a.strip() != "void"]
        want = _lib.SIGNATURES[name][1]
        assert len(args) == len(want), f"{name}: header has {len(args)} parameters, ctypes table {len(want)}"
        for a, w in zip(args, want):
            if "*" in a:
                kind = "ptr"
            elif a.startswith("float"):
                kind = "float"
            else:
                kind = "int"
            got = ("ptr" if w in (ctypes.c_void_p, _lib._WP) else
                   "float" if w is ctypes.c_float else "int")
            assert kind == got, f"{name}: parameter '{a}' is declared {kind} but bound as {got}"
            if kind == "int":
                size = 8 if ("long" in a.split() or "int64_t" in a) else 4
                assert ctypes.sizeof(w) == size, f"{name}: parameter '{a}' width mismatch"


def test_argument_validation_without_gpu(lib_path):
    lib = _lib.load()
    # rejected before any HIP call: null pointers / bad sizes -> AC_ERR_ARG
    assert lib.ac_linear(None, None, None, None, 4, 4, 32, 32, 32, 4, 0, None) == -1
    assert lib.ac_conv3x3_bn_relu(None, None, None, None, None, 1, 8, 4, 16, 32, 64, 0, -1, None) == -1
    assert lib.ac_gru_layer(None, None, None, None, None, 1, 1, 256, None) == -1
    w = _lib.AcTrmWeights()
    assert lib.ac_trm_workspace_floats(ctypes.byref(w), 4, 20) == -1  # zeroed config is invalid


def test_wide_decode_projection_sizes_and_argument_validation(lib_path):
    """csrc/decoder_wide.hip through the C ABI without a GPU: the pack size formula (32-row tiles x 16-k steps x three 1 KiB
    planes), and what the projection refuses before it would launch anything."""
    lib = _lib.load()
    assert lib.ac_dec_wide_packed_floats(100, 64) == 4 * 4 * 768
    assert lib.ac_dec_wide_packed_floats(4368, 256) == 137 * 16 * 768
    assert lib.ac_dec_wide_packed_floats(64, 24) == -1                       # K in 16-element steps
    nul = None
    call = lambda pro, K, ntb, split: lib.ac_dec_wide_gemm(pro, nul, 0, nul, 0, nul, nul, nul, 0, 0, nul, nul, 0.0, nul, 0,
                                                           nul, nul, nul, 64, 64, 64, K, 0, ntb, split, nul)
    assert call(0, 256, 1, 0) == -1 and call(2, 256, 1, 0) == -1 and call(3, 256, 1, 0) == -1   # null operands, unknown producer
    assert lib.ac_dec_wide_pack(nul, 256, 64, 256, nul, nul) == -1


def test_workspace_size_formula(lib_path):
    lib = _lib.load()
    w = _lib.AcTrmWeights()
    w.d_model, w.nhead, w.nlayers, w.dim_ff, w.vocab, w.max_pos, w.attn_emb_dim = 256, 4, 2, 1024, 4368, 100, 512
    n = lib.ac_trm_workspace_floats(ctypes.byref(w), 64, 20)
    cache = 2 * 2 * 2 * 64 * 20 * 256  # 2 sets x (K,V) x layers x rows x len x d
    assert n > cache + 64 * 4368


def test_state_dict_keys_and_param_count_match_reference_layout():
    model = A.init_model_from_config(A.cnn14rnn_trm_config(4981), print_fn=lambda s: None)
    sd = model.state_dict()
    want = P.cnn14rnn_trm_state(4981)
    # the reference's state dict also carries the two torchaudio MelSpectrogram buffers (cnn_encoder.py:338-348): present
    # here under the same names, optional on load (the procedural state - and the golden script's stand-in torchaudio -
    # have none)
    mel = {"encoder.cnn.melspec_extractor.spectrogram.window": (1024,), "encoder.cnn.melspec_extractor.mel_scale.fb": (513, 64)}
    assert set(sd) == set(want) | set(mel)
    for k in sd:
        assert tuple(sd[k].shape) == (mel[k] if k in mel else tuple(want[k].shape)), k
    model.load_state_dict(P.to_torch(want), strict=True)                    # without the mel buffers
    model.load_state_dict(dict(P.to_torch(want), **{k: sd[k] for k in mel}), strict=True)   # and with them
    assert sum(p.numel() for p in model.parameters()) == 90_395_840  # SURVEY.md §2.4 (AudioCaps vocab)
    trainable = sum(p.numel() for p in model.parameters() if p.requires_grad)
    assert trainable == 10_696_448  # GRU + decoder minus the frozen pe table: the DDP gradient payload
    assert not any(p.requires_grad for p in model.encoder.cnn.parameters())  # freeze_cnn


def test_stale_clotho_encoder_name_is_accepted():
    cfg = A.cnn14rnn_trm_config(4368, encoder_name="Cnn14RnnEncoder")
    model = A.init_model_from_config(cfg, print_fn=lambda s: None)
    assert isinstance(model.encoder, A.CrnnEncoder)


def test_decoder_compatibility_assertion():
    enc = torch.nn.Identity()

    class NotADecoder(torch.nn.Module):
        vocab_size = 10

    with pytest.raises(AssertionError):
        A.TransformerModel(enc, NotADecoder())


def test_set_index_is_class_level():
    try:
        A.TransformerModel.set_index(5, 6, 7)
        assert (A.TransformerModel.start_idx, A.TransformerModel.end_idx, A.TransformerModel.pad_idx) == (5, 6, 7)
    finally:
        A.TransformerModel.set_index(1, 2, 0)


def test_no_cpu_fallback():
    model = A.init_model_from_config(A.cnn14rnn_trm_config(64), print_fn=lambda s: None).eval()
    with pytest.raises(_lib.HipLibraryError):
        model({"mode": "inference", "wav": torch.zeros(1, 32000), "wav_len": [32000], "specaug": False})
    with pytest.raises(Exception):
        model({"mode": "bogus", "wav": torch.zeros(1, 32000), "wav_len": [32000], "specaug": False})
    model.train()
    with pytest.raises(_lib.HipLibraryError):
        model({"mode": "train", "wav": torch.zeros(1, 32000), "wav_len": [32000], "specaug": False,
               "cap": torch.ones(1, 5, dtype=torch.long), "cap_len": [5], "ss_ratio": 1.0})


def test_feat_len_and_geometry():
    from audiocaption_amd.cnn_encoder import cnn14_feat_len
    lens = cnn14_feat_len([320000, 280000, 160000, 300000], 320)
    assert lens.tolist() == [31, 27, 15, 29] and lens.dtype == torch.int64 and lens.device.type == "cpu"
    cnn = A.Cnn14Encoder(32000)
    T, H, Hp = cnn.geometry(320000)
    assert T == 1001 and H == [1001, 500, 250, 125, 62, 31] and Hp == [1024, 512, 256, 128, 64, 32]
    T, H, Hp = cnn.geometry(960000)
    assert H[5] == 93 and Hp[5] == 96 and all(Hp[k] == 2 * Hp[k + 1] and Hp[k] > H[k] for k in range(5))
    assert all(h % 4 == 0 for h in Hp)   # row quads of the F(4,3) kernels never straddle two clips
    with pytest.raises(ValueError):
        cnn.geometry(3000)


def test_pack_conv_weight_layout():
    from audiocaption_amd.kernels import pack_conv_weight
    w = torch.arange(128 * 64 * 9, dtype=torch.float32).reshape(128, 64, 3, 3)
    p = pack_conv_weight(w)
    assert p.shape == (2, 9, 128, 32)
    assert p[1, 5, 77, 3] == w[77, 35, 1, 2]  # chunk 1, tap ky=1,kx=2, cout 77, cin 32+3


def test_pack_conv_weight_fragment_layouts_and_fp16_split():
    """Host side of the two half-precision conv tiers: the MFMA fragment order (lane = cout % 32 + 32 * ((cin % 16)
    // 8), element = cin % 8) and the fp16 hi + lo split with its per-channel power-of-two scale."""
    from audiocaption_amd.kernels import pack_conv_weight_bf16x3_frag, pack_conv_weight_f16x2_frag
    g = torch.Generator().manual_seed(3)
    w = torch.randn(128, 64, 3, 3, generator=g) * torch.logspace(-4, 0, 128).view(-1, 1, 1, 1)   # channel scales 1e-4..1
    pb = pack_conv_weight_bf16x3_frag(w)
    pf, inv = pack_conv_weight_f16x2_frag(w)
    assert pb.shape == pf.shape == (2, 9, 2, 4, 2, 64, 8) and pb.dtype == torch.bfloat16 and pf.dtype == torch.float16
    for (c, tap, ks, nt, lane, e) in [(0, 0, 0, 0, 0, 0), (1, 5, 1, 3, 45, 6), (1, 8, 0, 2, 63, 7), (0, 4, 1, 1, 31, 3)]:
        cout, cin = nt * 32 + lane % 32, c * 32 + ks * 16 + (lane // 32) * 8 + e
        ref = w[cout, cin, tap // 3, tap % 3]
        assert float(pb[c, tap, ks, nt, 0, lane, e].float() + pb[c, tap, ks, nt, 1, lane, e].float()) == \
            pytest.approx(float(ref), rel=2 ** -15)
        got = (pf[c, tap, ks, nt, 0, lane, e].double() + pf[c, tap, ks, nt, 1, lane, e].double()) * inv[cout].double()
        assert float(got) == pytest.approx(float(ref), rel=0, abs=float(w[cout].abs().max()) * 2 ** -21)
    # the scales are exact powers of two that bring every channel's largest weight into [2^13, 2^14)
    m, ex = torch.frexp(inv)
    assert torch.all(m == 0.5)
    top = (w.abs().amax(dim=(1, 2, 3)) / inv)
    assert torch.all(top >= 2 ** 13) and torch.all(top < 2 ** 14)
    # no lo part is flushed: every weight above 2^-9 of its channel maximum keeps >= 20 significant bits
    full = (pf[..., 0, :, :].double() + pf[..., 1, :, :].double())
    assert torch.isfinite(full).all() and float(pf[..., 0, :, :].abs().max()) < 65504


def test_f16x2_argument_validation_without_gpu(lib_path):
    lib = _lib.load()
    assert lib.ac_conv3x3_bn_relu_f16x2_gw(None, None, None, None, None, 1, 8, 4, 16, 32, 64, 0, -1, 0, None, None) == -1
    assert lib.ac_conv3x3_first_f16(None, None, None, None, None, 1, 8, 4, 64, None, None) == -1
    assert lib.ac_conv3x3_block1_f16x2(None, None, None, None, None, None, None, None, 1, 8, 4, 64, None, None) == -1
    # 32-bit staging offsets: B*Hp*W*Cin >= 2^32 elements is refused, not wrapped (512 x 30 s clips in block 2)
    one = ctypes.c_void_p(16)   # never dereferenced: the size check comes before any launch
    assert lib.ac_conv3x3_bn_relu_f16x2_gw(one, one, one, one, one, 512, 3008, 3001, 32, 128, 128, 0, -1, 0, None, None) == -1
    assert lib.ac_conv3x3_bn_relu_bf16x3_gw(one, one, one, one, one, 512, 3008, 3001, 32, 128, 128, 0, -1, None) == -1
    # the f32 pooled output exists for mode 1 only
    assert lib.ac_conv3x3_bn_relu_f16x2_gw(one, one, one, one, one, 1, 8, 4, 16, 32, 64, 0, -1, 1, None, None) == -1


def test_compat_install_resolves_reference_dotted_paths():
    import importlib
    import sys
    from audiocaption_amd import compat
    saved = {k: v for k, v in sys.modules.items() if k.startswith("captioning")}
    try:
        compat.install()
        mod = importlib.import_module("captioning.models.crnn_trm_encoder")
        assert mod.CrnnEncoder is A.CrnnEncoder
        assert importlib.import_module("captioning.models.transformer_model").TransformerModel is A.TransformerModel
    finally:
        for k in [k for k in sys.modules if k.startswith("captioning")]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_procedural_weights_are_deterministic():
    a = P.decoder_state("decoder.", 100)
    b = P.decoder_state("decoder.", 100)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    w = P.synthetic_wav(2, 1000)
    assert w.dtype == np.float32 and np.abs(w).max() <= 1.0 and np.array_equal(w, P.synthetic_wav(2, 1000))


def test_mel_filterbank_properties():
    from audiocaption_amd.mel import melscale_fbanks
    fb = melscale_fbanks(513, 50.0, 14000.0, 64, 32000, "slaney", "slaney")
    assert fb.shape == (513, 64) and float(fb.min()) >= 0
    nz = (fb > 0)
    assert nz.any(0).all()                      # every filter has support
    first = nz.float().argmax(0)
    assert (first[1:] >= first[:-1]).all()      # supports move up monotonically
    assert int(nz.sum(1).max()) <= 2            # triangular: a bin feeds at most 2 filters
    freqs = torch.linspace(0, 16000, 513)
    assert float(fb[freqs < 50.0].abs().max()) == 0 and float(fb[freqs > 14000.0].abs().max()) == 0


def test_product_code_never_imports_the_oracle():
    """oracle/ is test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it."""
    pat = re.compile(r"^\s*(from\s+oracle\b|import\s+oracle\b|from\s+\.+oracle\b)", re.M)
    offenders = []
    for root in ("audiocaption_amd", "tools"):
        for dirpath, _, files in os.walk(os.path.join(REPO, root)):
            for f in files:
                if f.endswith(".py") and pat.search(open(os.path.join(dirpath, f)).read()):
                    offenders.append(os.path.join(dirpath, f))
    assert not offenders, offenders
    bench = open(os.path.join(REPO, "bench.py")).read()
    assert len(pat.findall(bench)) == 1 and "cpu_baseline" in bench[bench.index("from oracle"):][:3000]


def test_contrastive_kd_wrapper_loss():
    """ContraEncoderKdWrapper (hf_wrapper.py:1071-1112): with ``tchr_output`` the symmetric contrastive loss between the
    projected clip embedding and the projected teacher embedding comes back as ``enc_kd_loss`` - checked against the
    formula in float64 numpy; ``unsup`` routes through the encoder only.  (The captioner is a stub: no GPU here.)"""
    from audiocaption_amd.hf_wrapper import ContraEncoderKdWrapper

    class Enc(torch.nn.Module):
        fc_emb_size = 24

        def forward(self, d):
            return {"fc_emb": d["feat"], "from": "encoder"}

    class Cap(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.encoder = Enc()

        def forward(self, d):
            return {"fc_emb": d["feat"] * 2.0, "from": "model"}

    torch.manual_seed(3)
    w = ContraEncoderKdWrapper(Cap(), shared_dim=16, tchr_dim=12)
    assert abs(float(w.logit_scale) - np.log(1 / 0.07)) < 1e-6
    feat, tchr = torch.randn(5, 24), torch.randn(5, 12)
    for unsup, scale in ((False, 2.0), (True, 1.0)):
        out = w({"feat": feat, "unsup": unsup, "tchr_output": {"embedding": tchr}})
        assert out["from"] == ("encoder" if unsup else "model")
        s = (feat.double().numpy() * scale) @ w.stdnt_proj.weight.double().detach().numpy().T + w.stdnt_proj.bias.double().detach().numpy()
        t = tchr.double().numpy() @ w.tchr_proj.weight.double().detach().numpy().T + w.tchr_proj.bias.double().detach().numpy()
        s /= np.linalg.norm(s, axis=1, keepdims=True)
        t /= np.linalg.norm(t, axis=1, keepdims=True)
        z = float(w.logit_scale) * s @ t.T

        def ce(z):
            lse = np.log(np.exp(z - z.max(1, keepdims=True)).sum(1)) + z.max(1)
            return float((lse - np.diag(z)).mean())
        want = 0.5 * (ce(z) + ce(z.T))
        assert abs(float(out["enc_kd_loss"]) - want) < 1e-5
    assert "enc_kd_loss" not in w({"feat": feat})
    out["enc_kd_loss"].backward()
    assert w.stdnt_proj.weight.grad is not None and w.logit_scale.grad is not None
    # distilling INTO the encoder (mode "train") needs the encoder's backward, which the accelerated path does not have: a
    # clear refusal instead of a KeyError on the training output or a loss that trains only the heads
    with pytest.raises(NotImplementedError, match="head-only"):
        w({"feat": feat, "mode": "train", "tchr_output": {"embedding": tchr}})


def test_split_gru_publishes_before_it_polls(tmp_path):
    """The split GRU kernel's hand-off (csrc/gru.hip) lives on PROGRAM ORDER inside a wave: the lanes kq = 0 publish their
    hidden value as a tagged granule, the other lanes of the same wave then poll their partners' granules.  Relaxed atomics
    on different addresses may be reordered by the compiler (it did once, in a sibling variant: every part then waits for
    every other).  The source pins the order with a signal fence + a compiler barrier; this test checks the ISA hipcc emits
    for the shipped flags: inside the recurrence loop the write-through (sc1) granule store comes before the first sc1
    granule load, and the sleeping re-poll loop comes after both."""
    import subprocess
    try:
        hipcc = build._hipcc()
    except RuntimeError:
        pytest.skip("hipcc not found: the ISA of the shipped object cannot be inspected here")
    asm = tmp_path / "gru.s"
    cmd = [hipcc, "-x", "hip", "-S", "--cuda-device-only", os.path.join(build.CSRC, "gru.hip"), "-o", str(asm)] \
        + [f for f in build.FLAGS if f != "-fPIC"] + build.NO_PACKED_F32
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()
    text = asm.read_text().split("\n")
    start = next(i for i, l in enumerate(text) if re.match(r"^_ZN.*gru_layer_split_kernel.*:", l))
    end = next(i for i in range(start, len(text)) if text[i].startswith(".Lfunc_end"))
    body = [l.strip() for l in text[start:end]]
    stores = [i for i, l in enumerate(body) if l.startswith("global_store_dwordx2") and " sc1" in l]
    loads = [i for i, l in enumerate(body) if l.startswith("global_load_dwordx2") and " sc1" in l]
    sleeps = [i for i, l in enumerate(body) if l.startswith("s_sleep")]
    # ORDER only (instruction counts and exact mnemonics are the compiler's business and move with ROCm releases): a granule
    # store exists, the first granule load follows it, and no barrier / sleeping re-poll sits between the two
    assert stores and loads and sleeps, (stores, loads, sleeps)
    assert stores[0] < loads[0] < sleeps[0], (stores, loads, sleeps)
    between = body[stores[0] + 1:loads[0]]
    assert not any(l.startswith("s_barrier") or l.startswith("s_sleep") for l in between), between


def test_ingest_crop_offsets_cover_the_inclusive_range_for_every_rng_kind():
    """caption_dataset.py:124 draws ``random.randint(0, excess)`` - both ends included.  ``WaveformIngest`` accepts the
    ``random`` module, a ``random.Random``, a numpy ``Generator`` and a numpy ``RandomState`` (whose own upper bounds are
    exclusive): every one must reach offset ``excess`` and none may exceed it."""
    import random
    from audiocaption_amd.ingest import WaveformIngest
    for rng in (random, random.Random(1), np.random.default_rng(1), np.random.RandomState(1)):
        ing = WaveformIngest(32000, 32000, audio_duration=1.0, rng=rng)
        seen = {ing._draw_offset(3) for _ in range(400)}
        assert seen == {0, 1, 2, 3}, (type(rng), seen)
    r1, r2 = random.Random(7), random.Random(7)
    ing = WaveformIngest(32000, 32000, audio_duration=1.0, rng=r1)
    assert [ing._draw_offset(20000) for _ in range(5)] == [r2.randint(0, 20000) for _ in range(5)]   # the dataset's own draws
    with pytest.raises(TypeError):
        WaveformIngest(32000, 32000, audio_duration=1.0, rng=object())._draw_offset(3)
