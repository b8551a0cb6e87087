"""GPU tests of the wide decode route (csrc/decoder_wide.hip: 32-row workgroups, three-plane bf16 operands, six products per
k on the bf16 matrix cores) - the projections of a decode step over >= AUDIOCAPTION_DEC_WIDE_MIN rows (opt-in: a greedy chain
shared by several submissions, a beam search over grouped batches).  Every call goes through the C ABI.

Reference arithmetic: F.linear of nn.TransformerDecoderLayer / the classifier (transformer_decoder.py:92-101) in float64;
the searches against the launch chain's narrow route (exact f32) and against oracle/cpu_path.py."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from audiocaption_amd import _lib, build
    build.build()
    return _lib.load()


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _pack(lib, W):
    N, K = W.shape
    n = lib.ac_dec_wide_packed_floats(N, K)
    assert n == ((N + 31) // 32) * (K // 16) * 768
    out = torch.empty(n, device="cuda", dtype=torch.float32)
    assert lib.ac_dec_wide_pack(_p(W), K, N, K, _p(out), _stream()) == 0
    return out


def test_pack_is_an_exact_three_plane_split():
    """The three bf16 planes of a packed matrix add up to the f32 weights bit for bit, in the documented fragment order."""
    lib = _lib()
    g = torch.Generator().manual_seed(3)
    N, K = 100, 64
    W = (torch.randn(N, K, generator=g) * torch.logspace(-6, 3, N).unsqueeze(1)).cuda()
    pk = _pack(lib, W)
    raw = pk.view(torch.int16).cpu().view((N + 31) // 32, K // 16, 3, 64, 8)
    planes = (raw.to(torch.int32) << 16).view(torch.float32)          # bf16 -> f32
    lane = torch.arange(64)
    for nt in range((N + 31) // 32):
        for ks in range(K // 16):
            n = nt * 32 + (lane & 31)
            k = ks * 16 + (lane >> 5) * 8
            want = torch.zeros(64, 8)
            ok = n < N
            idx = k.unsqueeze(1) + torch.arange(8).unsqueeze(0)
            want[ok] = W.cpu()[n[ok]].gather(1, idx[ok])
            got = planes[nt, ks].double().sum(0)
            assert torch.equal(got.float(), want), (nt, ks)


def _unpack(pk, rows, K):
    """Fragment pack -> the f32 matrix it holds (the three planes added in float64)."""
    raw = pk.view(torch.int16).cpu().view((rows + 31) // 32, K // 16, 3, 64, 8)
    planes = (raw.to(torch.int32) << 16).view(torch.float32).double().sum(2)      # [tile][k step][lane][8]
    out = torch.zeros(((rows + 31) // 32) * 32, K, dtype=torch.float64)
    lane = torch.arange(64)
    for tile in range(planes.shape[0]):
        for ks in range(K // 16):
            r = tile * 32 + (lane & 31)
            k = ks * 16 + (lane >> 5) * 8
            out[r.unsqueeze(1), k.unsqueeze(1) + torch.arange(8).unsqueeze(0)] = planes[tile, ks]
    return out[:rows]


@pytest.mark.parametrize("M,N,K,relu", [(256, 256, 256, 0), (200, 768, 256, 1), (70, 1024, 256, 1), (96, 256, 1024, 0),
                                        (33, 4368, 256, 0), (64, 4981, 256, 0), (40, 192, 512, 1), (768, 64, 1024, 0)])
def test_wide_gemm_packed_rows_vs_float64(M, N, K, relu):
    """Producer 0: both operands as fragment packs (K = 256: column halves x K halves; 512 / 1024: K quarters)."""
    lib = _lib()
    g = torch.Generator().manual_seed(M * 7 + N)
    X = torch.randn(M, K, generator=g).cuda()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    Y = torch.full((M, N), float("nan"), device="cuda")
    pk, xs = _pack(lib, W), _pack(lib, X)
    rc = lib.ac_dec_wide_gemm(0, _p(xs), 0, None, 0, None, None, None, 0, 0, None, None, 0.0, None, 0, _p(pk), _p(b), _p(Y), N,
                              M, N, K, relu, 1, 0, _stream())
    assert rc == 0
    want = X.double().cpu() @ W.double().cpu().T + b.double().cpu()
    if relu:
        want = want.clamp_min(0)
    err = float((Y.cpu().double() - want).abs().max() / want.abs().max())
    f32 = X.cpu() @ W.cpu().T + b.cpu()
    f32 = f32.clamp_min(0) if relu else f32
    err32 = float((f32.double() - want).abs().max() / want.abs().max())
    print(f"M={M} N={N} K={K}: error {err:.2e} of the largest output (a CPU f32 matmul: {err32:.2e})")
    assert err < 1e-6          # f32 grade: the two-plane split of the conv tiers sits at 5e-6


@pytest.mark.parametrize("M,N,ntb", [(256, 768, 1), (77, 768, 1), (256, 4368, 2), (100, 4981, 4), (40, 4368, 8)])
def test_wide_gemm_fused_producers(M, N, ntb):
    """Producer 1 (embedding * sqrt(d) + positional row) and 2 (LayerNorm(X + Y2)): the produced rows (xout) against torch,
    the product against float64 of the produced rows; several column groups per workgroup; unaligned logits rows (N = 4981)."""
    lib = _lib()
    g = torch.Generator().manual_seed(M + N)
    d, V, T = 256, 500, 21
    W = (torch.randn(N, d, generator=g) / 16).cuda()
    b = torch.randn(N, generator=g).cuda()
    pk = _pack(lib, W)
    emb = torch.randn(V, d, generator=g).cuda()
    pe = torch.randn(T, d, generator=g).cuda()
    tok = torch.randint(0, V, (M, T), generator=g, dtype=torch.int32).cuda()
    t = 5
    xout = torch.empty(M, d, device="cuda")
    Y = torch.full((M, N), float("nan"), device="cuda")
    assert lib.ac_dec_wide_gemm(1, None, 0, None, 0, None, None, _p(tok), T, t, _p(emb), _p(pe), 16.0, _p(xout), d, _p(pk),
                                _p(b), _p(Y), N, M, N, d, 0, ntb, 0, _stream()) == 0
    a = emb[tok[:, t].long()] * 16.0 + pe[t]
    assert float((xout - a).abs().max()) <= 1e-6 * float(a.abs().max())
    want = xout.double().cpu() @ W.double().cpu().T + b.double().cpu()
    assert float((Y.cpu().double() - want).abs().max() / want.abs().max()) < 1e-6
    X = torch.randn(M, d, generator=g).cuda()
    Y2 = torch.randn(M, d, generator=g).cuda()
    lw, lb = (1 + 0.1 * torch.randn(d, generator=g)).cuda(), (0.1 * torch.randn(d, generator=g)).cuda()
    Y.fill_(float("nan"))
    assert lib.ac_dec_wide_gemm(2, _p(X), d, _p(Y2), d, _p(lw), _p(lb), None, 0, 0, None, None, 0.0, _p(xout), d, _p(pk),
                                _p(b), _p(Y), N, M, N, d, 1, ntb, 0, _stream()) == 0
    a = torch.nn.functional.layer_norm((X + Y2).double().cpu(), (d,), lw.double().cpu(), lb.double().cpu(), 1e-5)
    assert float((xout.cpu().double() - a).abs().max()) < 5e-6
    want = (xout.double().cpu() @ W.double().cpu().T + b.double().cpu()).clamp_min(0)
    assert float((Y.cpu().double() - want).abs().max() / want.abs().max()) < 1e-6


@pytest.mark.parametrize("M", [256, 50])
def test_wide_gemm_split_output_feeds_the_next_product(M):
    """split_out: the result of a fused launch leaves as a fragment pack - unpacked it equals the f32 result exactly (a
    three-plane split of an f32 value is exact), and a producer-0 launch consumes it (the feed-forward pair of a layer)."""
    lib = _lib()
    g = torch.Generator().manual_seed(M)
    d, ff = 256, 1024
    X, Y2 = torch.randn(M, d, generator=g).cuda(), torch.randn(M, d, generator=g).cuda()
    lw, lb = torch.ones(d).cuda(), torch.zeros(d).cuda()
    W1, b1 = (torch.randn(ff, d, generator=g) / 16).cuda(), torch.randn(ff, generator=g).cuda()
    W2, b2 = (torch.randn(d, ff, generator=g) / 32).cuda(), torch.randn(d, generator=g).cuda()
    p1, p2 = _pack(lib, W1), _pack(lib, W2)
    H = torch.empty(M, ff, device="cuda")
    Hp = torch.zeros(lib.ac_dec_wide_packed_floats(M, ff), device="cuda")
    for split, dst in ((0, H), (1, Hp)):
        assert lib.ac_dec_wide_gemm(2, _p(X), d, _p(Y2), d, _p(lw), _p(lb), None, 0, 0, None, None, 0.0, None, 0, _p(p1), _p(b1),
                                    _p(dst), ff, M, ff, d, 1, 1, split, _stream()) == 0
    assert torch.equal(_unpack(Hp, M, ff).float(), H.cpu())
    out = torch.empty(M, d, device="cuda")
    assert lib.ac_dec_wide_gemm(0, _p(Hp), 0, None, 0, None, None, None, 0, 0, None, None, 0.0, None, 0, _p(p2), _p(b2), _p(out), d,
                                M, d, ff, 0, 1, 0, _stream()) == 0
    want = H.double().cpu() @ W2.double().cpu().T + b2.double().cpu()
    assert float((out.cpu().double() - want).abs().max() / want.abs().max()) < 1e-6


def test_wide_gemm_rejects_what_it_does_not_cover():
    lib = _lib()
    X = torch.zeros(64, 512, device="cuda")
    pk = torch.zeros(lib.ac_dec_wide_packed_floats(64, 1024), device="cuda")
    Y = torch.zeros(64, 64, device="cuda")
    call = lambda pro, K, ntb, split: lib.ac_dec_wide_gemm(pro, _p(X), 512, _p(X), 512, _p(X), _p(X), None, 0, 0, None, None, 0.0,
                                                           None, 0, _p(pk), None, _p(Y), 64, 64, 64, K, 0, ntb, split, _stream())
    assert call(0, 320, 1, 0) == -1       # packed rows: K = 256, 512 or 1024
    assert call(2, 512, 1, 0) == -1       # a fused producer stages a 256-wide row
    assert call(0, 256, 2, 0) == -1       # several column groups only with a fused producer
    assert call(0, 256, 1, 1) == -1       # split output only from a fused producer
    assert call(0, 256, 1, 0) == 0


def _enc(B, Tm, seed, ragged=True):
    g = torch.Generator().manual_seed(seed)
    attn = torch.randn(B, Tm, 512, generator=g) * 0.5
    lens = torch.randint(max(1, Tm // 3), Tm + 1, (B,), generator=g) if ragged else torch.full((B,), Tm)
    lens[0] = Tm
    for b in range(B):
        attn[b, int(lens[b]):] = 0.0
    return attn, lens


@pytest.mark.parametrize("B,Tm", [(256, 31), (200, 31), (192, 94), (530, 15)])
def test_greedy_wide_route_equals_narrow_route(hip_model, monkeypatch, B, Tm):
    """The same greedy chain by both routes (eager launches, so that the switch is read per call): ids, stop bookkeeping equal,
    logits / embeddings / log-probabilities within 2e-5 (different summation orders of f32-grade products)."""
    dec = hip_model.decoder
    attn, lens = _enc(B, Tm, seed=B + Tm)
    args = (attn.cuda(), lens, 20, hip_model.start_idx, hip_model.end_idx, hip_model.pad_idx)
    monkeypatch.setenv("AUDIOCAPTION_DECODE_GRAPH", "0")
    monkeypatch.setenv("AUDIOCAPTION_DEC_WIDE_MIN", "0")
    want = dec.greedy(*args, mode="chain")
    torch.cuda.synchronize()
    monkeypatch.setenv("AUDIOCAPTION_DEC_WIDE_MIN", "128")
    got = dec.greedy(*args, mode="chain")
    torch.cuda.synchronize()
    cnt_w, cnt_g = want["unfinished_cnt"].cpu().numpy(), got["unfinished_cnt"].cpu().numpy()
    np.testing.assert_array_equal(cnt_g, cnt_w)
    np.testing.assert_array_equal(got["seq"].cpu().numpy(), want["seq"].cpu().numpy())
    steps = int(np.argmax(cnt_w == 0)) + 1 if (cnt_w == 0).any() else 20
    d = float((got["logit"][:, :steps] - want["logit"][:, :steps]).abs().max())
    e = float((got["embed"][:, :steps] - want["embed"][:, :steps]).abs().max())
    lp = float((got["sampled_logprob"] - want["sampled_logprob"]).abs().max())
    print(f"B={B} Tm={Tm}: steps {steps}, max|dlogit| {d:.2e} |dembed| {e:.2e} |dlogprob| {lp:.2e}")
    assert d < 2e-5 and e < 2e-5 and lp < 2e-5
    assert not torch.equal(got["logit"], want["logit"]), "both runs took the same route"


def test_greedy_wide_route_vs_oracle(hip_model, state4981, monkeypatch):
    """A 256-row chain (what four 64-clip submissions of forward_async share) on the wide route against the oracle's decoder
    (base.py:152-218 restated in oracle/cpu_path.py): ids identical, logits within 1e-4."""
    from oracle import cpu_path as O
    monkeypatch.setenv("AUDIOCAPTION_DEC_WIDE_MIN", "128")
    monkeypatch.setenv("AUDIOCAPTION_DECODE_GRAPH", "0")
    B, Tm = 256, 31
    attn, lens = _enc(B, Tm, seed=77)
    got = hip_model.decoder.greedy(attn.cuda(), lens, 20, hip_model.start_idx, hip_model.end_idx, hip_model.pad_idx, mode="chain")
    torch.cuda.synchronize()
    want = O.greedy_decode(state4981, attn, lens, 20)
    st = want["steps"]
    assert torch.equal(got["seq"].cpu()[:, :st], want["seq"][:, :st])
    d = float((got["logit"].cpu()[:, :st] - want["logit"][:, :st]).abs().max())
    print(f"256 rows, {st} steps: max|dlogit| vs the oracle {d:.2e}")
    assert d < 1e-4


def test_general_launch_sequence_at_512_rows_and_more(hip_model, monkeypatch):
    """From 512 rows on the projections take two column tiles per block - which only a single-chunk K allows.  The second
    feed-forward product (K = dim_ff = 1024: two chunks) of the GENERAL 18-launch sequence (AUDIOCAPTION_DEC_ROW=gemm; the
    route of every decoder shape other than d_model 256 / 4 heads) used to be launched with two tiles and rejected
    (AC_ERR_ARG): the launcher now takes one tile per block there.  Same ids, logits within 2e-5 of the fused row kernels."""
    dec = hip_model.decoder
    attn, lens = _enc(528, 15, seed=5)
    args = (attn.cuda(), lens, 6, hip_model.start_idx, hip_model.end_idx, hip_model.pad_idx)
    monkeypatch.setenv("AUDIOCAPTION_DECODE_GRAPH", "0")
    monkeypatch.setenv("AUDIOCAPTION_DEC_WIDE_MIN", "0")
    want = dec.greedy(*args, mode="chain")
    monkeypatch.setenv("AUDIOCAPTION_DEC_ROW", "gemm")
    got = dec.greedy(*args, mode="chain")
    torch.cuda.synchronize()
    np.testing.assert_array_equal(got["seq"].cpu().numpy(), want["seq"].cpu().numpy())
    assert float((got["logit"] - want["logit"]).abs().max()) < 2e-5
    assert not torch.equal(got["logit"], want["logit"]), "both runs took the same launch sequence"


def test_hybrid_route_from_512_rows_equals_the_narrow_kernels(hip_model, monkeypatch):
    """AUDIOCAPTION_DEC_HYBRID=1 (opt-in): from 512 rows on the joined QKV projection and the classifier run on their wide twins -
    same ids and stop bookkeeping as the all-narrow chain, logits within 2e-5; below 512 rows nothing changes (bit-equal)."""
    dec = hip_model.decoder
    monkeypatch.setenv("AUDIOCAPTION_DECODE_GRAPH", "0")
    monkeypatch.setenv("AUDIOCAPTION_DEC_WIDE_MIN", "0")
    for B, differs in ((530, True), (300, False)):
        attn, lens = _enc(B, 15, seed=B)
        args = (attn.cuda(), lens, 10, hip_model.start_idx, hip_model.end_idx, hip_model.pad_idx)
        monkeypatch.delenv("AUDIOCAPTION_DEC_HYBRID", raising=False)
        want = dec.greedy(*args, mode="chain")
        monkeypatch.setenv("AUDIOCAPTION_DEC_HYBRID", "1")
        got = dec.greedy(*args, mode="chain")
        torch.cuda.synchronize()
        np.testing.assert_array_equal(got["seq"].cpu().numpy(), want["seq"].cpu().numpy())
        np.testing.assert_array_equal(got["unfinished_cnt"].cpu().numpy(), want["unfinished_cnt"].cpu().numpy())
        assert float((got["logit"] - want["logit"]).abs().max()) < 2e-5
        assert torch.equal(got["logit"], want["logit"]) != differs
