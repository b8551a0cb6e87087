"""Thin Python entry points over the C ABI (include/audiocaption_hip.h).

Each function takes/returns torch tensors that live on the ROCm device, passes raw pointers and the
current stream to libaudiocaption_hip.so and raises on any failure.  No function here computes
anything on the host and none has a fallback.
"""
import os

import torch

from . import _lib
from ._lib import check, f32c, ptr, stream


def upload(t, device, dtype=None):
    """Host tensor / list -> device WITHOUT blocking the host.  A plain ``.to(device)`` of pageable memory makes the host
    wait until the stream has executed everything submitted before it (the whole conv stack, when the lengths are
    uploaded between the encoder's kernels), so nothing can be queued ahead; a pinned staging tensor (kept alive by
    torch's caching host allocator until the copy has run) does not."""
    t = torch.as_tensor(t)
    if t.is_cuda:
        return t.to(device=device, dtype=dtype)
    if dtype is not None:
        t = t.to(dtype)
    pin = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    pin.copy_(t)
    return pin.to(device, non_blocking=True)


def _dev(t):
    if not t.is_cuda:
        raise _lib.HipLibraryError("the HIP path needs tensors on a ROCm device; there is no CPU fallback")
    return t


def logmel(wav, tables, scale=None, shift=None, rows_per_clip=None, channels_last=True):
    """wav (B, L) -> log-mel.  channels_last: (B*rows_per_clip, 64) rows layout for the conv stack
    (frames >= T zero); else (B, 64, T) like the reference's MelSpectrogram+AmplitudeToDB output."""
    lib = _lib.load()
    wav = f32c(_dev(wav))
    B, L = wav.shape
    T = L // tables.hop + 1
    if channels_last:
        Hp = rows_per_clip if rows_per_clip is not None else T
        out = torch.empty(B * Hp, 64, device=wav.device, dtype=torch.float32)
        sb, st, sm = Hp * 64, 64, 1
    else:
        Hp = T
        out = torch.empty(B, 64, T, device=wav.device, dtype=torch.float32)
        sb, st, sm = 64 * T, 1, T
    check(lib.ac_logmel(ptr(wav), B, L, tables.n_fft, tables.hop, ptr(tables.window), ptr(tables.twiddle),
                        ptr(tables.melfb), ptr(tables.mel_lo), ptr(tables.mel_hi), ptr(scale), ptr(shift),
                        ptr(out), Hp, sb, st, sm, stream()), "ac_logmel")
    return out


def conv3x3_first(x, w, scale, shift, out, B, Hp, H, W=64, overflow=None):
    """``out`` float32, or float16 for the "f16x2" conv tier (``overflow``: its fp16 range flag, see
    ``conv3x3_bn_relu_f16x2_gw``)."""
    lib = _lib.load()
    if out.dtype == torch.float16:
        check(lib.ac_conv3x3_first_f16(ptr(x), ptr(w), ptr(scale), ptr(shift), ptr(out), B, Hp, H, W, ptr(overflow),
                                       stream()), "ac_conv3x3_first_f16")
        return out
    check(lib.ac_conv3x3_first(ptr(x), ptr(w), ptr(scale), ptr(shift), ptr(out), B, Hp, H, W, stream()),
          "ac_conv3x3_first")
    return out


def conv3x3_block1_f16x2(x0, w1, scale1, shift1, wfrag2, scale2, shift2, out, B, Hp, H, W=64, overflow=None):
    """conv_block1 of the "f16x2" tier in one launch (conv1 computed into conv2's patch); out fp16."""
    lib = _lib.load()
    if out.dtype != torch.float16:
        raise ValueError("conv3x3_block1_f16x2 writes fp16")
    check(lib.ac_conv3x3_block1_f16x2(ptr(x0), ptr(w1), ptr(scale1), ptr(shift1), ptr(wfrag2), ptr(scale2),
                                      ptr(shift2), ptr(out), B, Hp, H, W, ptr(overflow), stream()),
          "ac_conv3x3_block1_f16x2")
    return out


# Optional per-launch observer used by bench.py to time the dominant kernel with HIP events on the
# launch stream: called as hook(phase, info) with phase "pre"/"post" around every MFMA conv launch.
CONV_LAUNCH_HOOK = None


def conv3x3_bn_relu(x, wpk, scale, shift, out, B, Hp, H, W, Cin, Cout, mode, map_mode=-1):
    lib = _lib.load()
    hook = CONV_LAUNCH_HOOK
    if hook is not None:
        info = {"B": B, "H": H, "Hp": Hp, "W": W, "Cin": Cin, "Cout": Cout, "mode": mode, "algo": "direct"}
        hook("pre", info)
    check(lib.ac_conv3x3_bn_relu(ptr(x), ptr(wpk), ptr(scale), ptr(shift), ptr(out), B, Hp, H, W, Cin, Cout,
                                 mode, map_mode, stream()), "ac_conv3x3_bn_relu")
    if hook is not None:
        hook("post", info)
    return out


def conv3x3_bn_relu_winograd(x, upk, scale, shift, out, B, Hp, H, W, Cin, Cout, mode, map_mode=-1):
    lib = _lib.load()
    hook = CONV_LAUNCH_HOOK
    if hook is not None:
        info = {"B": B, "H": H, "Hp": Hp, "W": W, "Cin": Cin, "Cout": Cout, "mode": mode, "algo": "winograd"}
        hook("pre", info)
    check(lib.ac_conv3x3_bn_relu_winograd(ptr(x), ptr(upk), ptr(scale), ptr(shift), ptr(out), B, Hp, H, W, Cin,
                                          Cout, mode, map_mode, stream()), "ac_conv3x3_bn_relu_winograd")
    if hook is not None:
        hook("post", info)
    return out


def conv3x3_bn_relu_bf16x3(x, wpk, scale, shift, out, B, Hp, H, W, Cin, Cout, mode, map_mode=-1):
    lib = _lib.load()
    hook = CONV_LAUNCH_HOOK
    if hook is not None:
        info = {"B": B, "H": H, "Hp": Hp, "W": W, "Cin": Cin, "Cout": Cout, "mode": mode, "algo": "bf16x3"}
        hook("pre", info)
    check(lib.ac_conv3x3_bn_relu_bf16x3(ptr(x), ptr(wpk), ptr(scale), ptr(shift), ptr(out), B, Hp, H, W, Cin, Cout,
                                        mode, map_mode, stream()), "ac_conv3x3_bn_relu_bf16x3")
    if hook is not None:
        hook("post", info)
    return out


def pack_conv_weight_bf16x3(w):
    """OIHW f32 -> split bf16 planes [Cin/32][9][2 (hi, lo)][Cout][32] (csrc/conv3x3.hip, bf16x3 kernel).
    hi = RNE_bf16(w), lo = RNE_bf16(w - hi)."""
    cout, cin = w.shape[0], w.shape[1]
    hi = w.to(torch.bfloat16)
    lo = (w - hi.float()).to(torch.bfloat16)

    def lay(t):
        return t.permute(1, 2, 3, 0).reshape(cin // 32, 32, 9, cout).permute(0, 2, 3, 1)

    return torch.stack([lay(hi), lay(lo)], dim=2).contiguous()


def conv3x3_bn_relu_bf16x3_gw(x, wfrag, scale, shift, out, B, Hp, H, W, Cin, Cout, mode, map_mode=-1):
    lib = _lib.load()
    hook = CONV_LAUNCH_HOOK
    if hook is not None:
        info = {"B": B, "H": H, "Hp": Hp, "W": W, "Cin": Cin, "Cout": Cout, "mode": mode, "algo": "bf16x3"}
        hook("pre", info)
    check(lib.ac_conv3x3_bn_relu_bf16x3_gw(ptr(x), ptr(wfrag), ptr(scale), ptr(shift), ptr(out), B, Hp, H, W, Cin,
                                           Cout, mode, map_mode, stream()), "ac_conv3x3_bn_relu_bf16x3_gw")
    if hook is not None:
        hook("post", info)
    return out


def wino1d_splitk_floats(B, Hp, W, Cin, Cout):
    """Floats of workspace ``conv3x3_bn_relu_wino1d(..., workspace=...)`` needs to run this geometry K-sliced (launches of
    a few workgroups: single clips); 0 = the launch is not split."""
    return int(_lib.load().ac_conv3x3_wino1d_splitk_floats(B, Hp, W, Cin, Cout))


def conv3x3_bn_relu_wino1d(x, wfrag, scale, shift, out, B, Hp, H, W, Cin, Cout, mode, map_mode=-1, need=None,
                           workspace=None, dropout=None):
    """F(2,3) Winograd along time on split-bf16 operands (csrc/conv3x3_wino1d.hip); ``wfrag`` from
    ``pack_conv_weight_wino1d_frag``.  Covers Cout % 128 == 0 and Cout == 64 with W % 16 == 0 (conv2 of block 1); other
    layers must be routed to ``conv3x3_bn_relu_bf16x3_gw`` by the caller.  ``need = (clip_frames int32 device tensor, mul, add)``: ragged
    batches - output rows at or beyond ``mul * clip_frames[b] + add`` of clip b are not computed (stored as zeros).
    ``workspace`` (f32 tensor of at least ``wino1d_splitk_floats`` elements): few-workgroup launches run K-sliced over it.
    ``dropout = (p, seed, seed_dev)``: F.dropout on the layer's output inside the epilogue - the mask ``dropout_`` over
    ``out`` with the same seed would apply (train-mode forward of the frozen network; modes 0 and 1, uniform batches)."""
    lib = _lib.load()
    hook = CONV_LAUNCH_HOOK
    if hook is not None:
        info = {"B": B, "H": H, "Hp": Hp, "W": W, "Cin": Cin, "Cout": Cout, "mode": mode, "algo": "wino1d"}
        hook("pre", info)
    cf, mul, add = need if need is not None else (None, 0, 0)
    if dropout is not None:
        if cf is not None:
            raise ValueError("conv3x3_bn_relu_wino1d: dropout and dead-row skipping are not combined")
        check(lib.ac_conv3x3_bn_relu_wino1d_drop(ptr(x), ptr(wfrag), ptr(scale), ptr(shift), ptr(out), B, Hp, H, W, Cin, Cout,
                                                 mode, map_mode, float(dropout[0]), int(dropout[1]), dropout[2], stream()),
              "ac_conv3x3_bn_relu_wino1d_drop")
    elif workspace is not None:
        check(lib.ac_conv3x3_bn_relu_wino1d_splitk(ptr(x), ptr(wfrag), ptr(scale), ptr(shift), ptr(out), B, Hp, H, W, Cin,
                                                   Cout, mode, map_mode, ptr(cf), int(mul), int(add), ptr(workspace),
                                                   workspace.numel(), stream()),
              "ac_conv3x3_bn_relu_wino1d_splitk")
    else:
        check(lib.ac_conv3x3_bn_relu_wino1d(ptr(x), ptr(wfrag), ptr(scale), ptr(shift), ptr(out), B, Hp, H, W, Cin,
                                            Cout, mode, map_mode, ptr(cf), int(mul), int(add), stream()),
              "ac_conv3x3_bn_relu_wino1d")
    if hook is not None:
        hook("post", info)
    return out


def conv3x3_block1_wino43(x0, w1, scale1, shift1, wfrag2, scale2, shift2, out, B, Hp, H, need=None, dropout=None,
                          conv1="mfma"):
    """Conv block 1 in one kernel at f32 grade: conv1 computed into conv2's staging, conv2 + pool as F(4,3) on split-bf16
    operands (csrc/conv3x3_block1_w4.hip).  x0 [B*Hp][64] f32, out [B*Hp/2][32][64] f32; ``wfrag2`` from
    ``pack_conv_weight_wino43_frag``; ``need`` / ``dropout`` as for ``conv3x3_bn_relu_wino1d``.  ``conv1``: "mfma" (a
    split-bf16 product on the matrix cores) or "valu" (the f32 chain of ``conv3x3_first``, bit-identical to it)."""
    if conv1 not in ("mfma", "valu"):
        raise ValueError(f"conv1 = {conv1!r}")
    cf, mul, add = need if need is not None else (None, 0, 0)
    dp, dseed, ddev = dropout if dropout is not None else (0.0, 0, None)
    hook = CONV_LAUNCH_HOOK
    if hook is not None:
        info = {"B": B, "H": H, "Hp": Hp, "W": 64, "Cin": 1, "Cout": 64, "mode": 1, "algo": "block1_w4", "conv1": conv1}
        hook("pre", info)
        try:
            return _block1_wino43(x0, w1, scale1, shift1, wfrag2, scale2, shift2, out, B, Hp, H, cf, mul, add, dp, dseed, ddev,
                                  conv1)
        finally:
            hook("post", info)
    return _block1_wino43(x0, w1, scale1, shift1, wfrag2, scale2, shift2, out, B, Hp, H, cf, mul, add, dp, dseed, ddev, conv1)


def _block1_wino43(x0, w1, scale1, shift1, wfrag2, scale2, shift2, out, B, Hp, H, cf, mul, add, dp, dseed, ddev, conv1):
    name = "ac_conv3x3_block1_wino43_mfma" if conv1 == "mfma" else "ac_conv3x3_block1_wino43"
    check(getattr(_lib.load(), name)(ptr(x0), ptr(w1), ptr(scale1), ptr(shift1), ptr(wfrag2), ptr(scale2), ptr(shift2),
                                     ptr(out), B, Hp, H, ptr(cf), int(mul), int(add), float(dp), int(dseed), ddev,
                                     stream()), name)
    return out


def conv3x3_block1_conv2_wino43(x64, wfrag2, scale2, shift2, out, B, Hp, H):
    """The conv2 + pool half of ``conv3x3_block1_wino43`` on a 64-channel input in HBM (tests: the unfused form)."""
    check(_lib.load().ac_conv3x3_block1_conv2_wino43(ptr(x64), ptr(wfrag2), ptr(scale2), ptr(shift2), ptr(out), B, Hp, H,
                                                     stream()), "ac_conv3x3_block1_conv2_wino43")
    return out


def wino43_workgroups(B, Hp, W, Cout):
    """Workgroups of ``conv3x3_bn_relu_wino43`` on this geometry (0: the F(4,3) kernel does not cover it)."""
    return int(_lib.load().ac_conv3x3_wino43_workgroups(B, Hp, W, Cout))


def conv3x3_bn_relu_wino43(x, wfrag, scale, shift, out, B, Hp, H, W, Cin, Cout, mode, map_mode=-1, need=None, dropout=None,
                           tiles_per_wave=0):
    """F(4,3) Winograd along time on split-bf16 operands, one wave per SIMD (csrc/conv3x3_wino43.hip); ``wfrag`` from
    ``pack_conv_weight_wino43_frag``.  Covers W in (32, 16, 8, 4) with modes 0 / 1 and W = 2 with modes 0 / 2 (mean over
    mel), Cout % 128 == 0, Hp % 4 == 0; ``need`` and ``dropout`` as for ``conv3x3_bn_relu_wino1d``."""
    lib = _lib.load()
    hook = CONV_LAUNCH_HOOK
    if hook is not None:
        info = {"B": B, "H": H, "Hp": Hp, "W": W, "Cin": Cin, "Cout": Cout, "mode": mode, "algo": "wino43"}
        hook("pre", info)
    cf, mul, add = need if need is not None else (None, 0, 0)
    if dropout is not None:
        if cf is not None:
            raise ValueError("conv3x3_bn_relu_wino43: dropout and dead-row skipping are not combined")
        check(lib.ac_conv3x3_bn_relu_wino43_drop(ptr(x), ptr(wfrag), ptr(scale), ptr(shift), ptr(out), B, Hp, H, W, Cin, Cout,
                                                 mode, map_mode, float(dropout[0]), int(dropout[1]), dropout[2], stream()),
              "ac_conv3x3_bn_relu_wino43_drop")
    else:
        check(lib.ac_conv3x3_bn_relu_wino43(ptr(x), ptr(wfrag), ptr(scale), ptr(shift), ptr(out), B, Hp, H, W, Cin, Cout,
                                            mode, map_mode, int(tiles_per_wave), ptr(cf), int(mul), int(add), stream()),
              "ac_conv3x3_bn_relu_wino43")
    if hook is not None:
        hook("post", info)
    return out


def skinny_workspace_floats(B, Hp, W, Cin, Cout):
    """Floats of workspace ``conv3x3_bn_relu_skinny`` needs for this geometry; 0 = not a launch for that kernel (more than
    2048 pixels, W other than 2 / 4)."""
    return int(_lib.load().ac_conv3x3_skinny_workspace_floats(B, Hp, W, Cin, Cout))


def conv3x3_bn_relu_skinny(x, wfrag, scale, shift, out, B, Hp, H, W, Cin, Cout, mode, workspace, dropout=None):
    """Few pixels, heavy weights (conv blocks 5-6 of single clips / small training batches): the K-sliced direct convolution
    on split-bf16 operands that streams its weights once (csrc/conv3x3_skinny.hip); ``wfrag`` from
    ``pack_conv_weight_bf16x3_frag``; ``workspace``: f32 tensor of at least ``skinny_workspace_floats`` elements;
    ``dropout = (p, seed, seed_dev)`` as for ``conv3x3_bn_relu_wino1d`` (modes 0 and 1)."""
    hook = CONV_LAUNCH_HOOK
    if hook is not None:
        info = {"B": B, "H": H, "Hp": Hp, "W": W, "Cin": Cin, "Cout": Cout, "mode": mode, "algo": "skinny"}
        hook("pre", info)
    dp, dseed, ddev = dropout if dropout is not None else (0.0, 0, None)
    try:
        check(_lib.load().ac_conv3x3_bn_relu_skinny(ptr(x), ptr(wfrag), ptr(scale), ptr(shift), ptr(out), B, Hp, H, W, Cin, Cout,
                                                   mode, ptr(workspace), workspace.numel(), float(dp), int(dseed), ddev, stream()),
              "ac_conv3x3_bn_relu_skinny")
    finally:   # a rejected launch must not leave the observer's "pre" event unpaired
        if hook is not None:
            hook("post", info)
    return out


def conv3x3_bn_relu_f16x2_gw(x, wfrag, scale, shift, out, B, Hp, H, W, Cin, Cout, mode, map_mode=-1, overflow=None):
    """``out``: fp16 for modes 0 / 1 (an f32 ``out`` with mode 1 selects the f32 pooled output that feeds a split-bf16
    block), f32 for mode 2.  ``overflow``: a uint32 / int32 device word OR-ed with 1 when a value stored as fp16
    exceeded the fp16 range."""
    lib = _lib.load()
    hook = CONV_LAUNCH_HOOK
    if hook is not None:
        info = {"B": B, "H": H, "Hp": Hp, "W": W, "Cin": Cin, "Cout": Cout, "mode": mode, "algo": "f16x2"}
        hook("pre", info)
    out_f32 = 1 if (mode == 1 and out.dtype == torch.float32) else 0
    if x.dtype != torch.float16 or out.dtype != (torch.float32 if (mode == 2 or out_f32) else torch.float16):
        raise ValueError("f16x2 conv: fp16 activations in, fp16 out (f32 for mode 2)")
    check(lib.ac_conv3x3_bn_relu_f16x2_gw(ptr(x), ptr(wfrag), ptr(scale), ptr(shift), ptr(out), B, Hp, H, W,
                                          Cin, Cout, mode, map_mode, out_f32, ptr(overflow), stream()),
          "ac_conv3x3_bn_relu_f16x2_gw")
    if hook is not None:
        hook("post", info)
    return out


def pack_conv_weight_f16x2_frag(w):
    """OIHW f32 -> (fp16 hi + lo in the fragment order of pack_conv_weight_bf16x3_frag, inv_scale[Cout]).
    Every output channel is first multiplied by the power of two that brings its largest |w| into [2^13, 2^14): the
    lo parts (~2^-12 of the hi parts) are then normal fp16 numbers for every weight above 2^-15 of the channel maximum;
    ``inv_scale`` (exact powers of two) goes into the BN scale of the epilogue."""
    cout, cin = w.shape[0], w.shape[1]
    amax = w.abs().amax(dim=(1, 2, 3)).clamp_min(1e-30)
    e = 13 - torch.floor(torch.log2(amax))
    ws = w * torch.exp2(e).view(-1, 1, 1, 1)
    hi = ws.to(torch.float16)
    lo = (ws - hi.float()).to(torch.float16)

    def lay(t):
        t = t.permute(1, 2, 3, 0).reshape(cin // 32, 2, 2, 8, 9, cout // 32, 32)
        return t.permute(0, 4, 1, 5, 2, 6, 3).reshape(cin // 32, 9, 2, cout // 32, 64, 8)

    return torch.stack([lay(hi), lay(lo)], dim=4).contiguous(), torch.exp2(-e).contiguous()


def pack_conv_weight_bf16x3_frag(w):
    """OIHW f32 -> split bf16 in MFMA fragment order [Cin/32][9][2 ks][Cout/32][2 (hi, lo)][64 lanes][8]:
    lane = (cout % 32) + 32 * ((cin % 16) // 8), element = cin % 8 (csrc/conv3x3.hip, "gw" kernel)."""
    cout, cin = w.shape[0], w.shape[1]
    hi = w.to(torch.bfloat16)
    lo = (w - hi.float()).to(torch.bfloat16)

    taps = w.shape[2] * w.shape[3]   # 9, or 1 for the linear layers

    def lay(t):
        # (cout, cin, 3, 3) -> (cin, tap, cout) -> [c][ks][h][e][tap][nt][r] -> [c][tap][ks][nt][h][r][e]
        t = t.permute(1, 2, 3, 0).reshape(cin // 32, 2, 2, 8, taps, cout // 32, 32)
        return t.permute(0, 4, 1, 5, 2, 6, 3).reshape(cin // 32, taps, 2, cout // 32, 64, 8)

    return torch.stack([lay(hi), lay(lo)], dim=4).contiguous()


def pack_conv_weight_wino1d_frag(w):
    """OIHW f32 -> U = G g (F(2,3) filter transform over the time taps ky, evaluated in float64), split into bf16
    hi + lo, in MFMA fragment order [Cin/32][12 = 3 kx x 4 positions][2 ks][Cout/32][2 (hi, lo)][64 lanes][8]
    (csrc/conv3x3_wino1d.hip).  Pure elementwise / small-matrix torch ops: no library GEMM."""
    cout, cin = w.shape[0], w.shape[1]
    g = w.double()                                                   # (o, c, ky, kx)
    u = torch.stack([g[:, :, 0], 0.5 * (g[:, :, 0] + g[:, :, 1] + g[:, :, 2]),
                     0.5 * (g[:, :, 0] - g[:, :, 1] + g[:, :, 2]), g[:, :, 2]], dim=3)   # (o, c, kx, position)
    u = u.reshape(cout, cin, 12, 1)                                  # "tap" = kx * 4 + position
    hi = u.to(torch.bfloat16)
    lo = (u - hi.double()).to(torch.bfloat16)

    def lay(t):
        t = t.permute(1, 2, 3, 0).reshape(cin // 32, 2, 2, 8, 12, cout // 32, 32)
        return t.permute(0, 4, 1, 5, 2, 6, 3).reshape(cin // 32, 12, 2, cout // 32, 64, 8)

    return torch.stack([lay(hi), lay(lo)], dim=4).contiguous()


def pack_conv_weight_wino43_frag(w):
    """OIHW f32 -> U = G g (F(4,3) filter transform over the time taps ky, evaluated in float64), split into bf16
    hi + lo, in MFMA fragment order [Cin/16][18 = 3 kx x 6 positions][Cout/32][2 (hi, lo)][64 lanes][8]: lane
    (cout % 32) + 32 * ((cin % 16) // 8), element cin % 8 (csrc/conv3x3_wino43.hip).  Elementwise torch ops only."""
    cout, cin = w.shape[0], w.shape[1]
    g = w.double()                                                   # (o, c, ky, kx)
    G = ((1 / 4, 0, 0), (-1 / 6, -1 / 6, -1 / 6), (-1 / 6, 1 / 6, -1 / 6), (1 / 24, 1 / 12, 1 / 6), (1 / 24, -1 / 12, 1 / 6),
         (0, 0, 1))
    u = torch.stack([G[p][0] * g[:, :, 0] + G[p][1] * g[:, :, 1] + G[p][2] * g[:, :, 2] for p in range(6)], dim=3)   # (o, c, kx, p)
    u = u.reshape(cout, cin, 18)                                     # tap = kx * 6 + position
    hi = u.to(torch.bfloat16)
    lo = (u - hi.double()).to(torch.bfloat16)

    def lay(t):
        t = t.permute(1, 2, 0).reshape(cin // 16, 2, 8, 18, cout // 32, 32)     # (step, k-half, j, tap, channel tile, n)
        return t.permute(0, 3, 4, 1, 5, 2).reshape(cin // 16, 18, cout // 32, 64, 8)

    return torch.stack([lay(hi), lay(lo)], dim=3).contiguous()


def pack_conv_weight_winograd(w):
    """OIHW (Cout, Cin, 3, 3) -> U = G g G^T as [Cin/32][4 j][4 i][Cout][32] (csrc/conv3x3_winograd.hip).
    The transform is evaluated in float64 and rounded once to fp32 - with elementwise adds only (G has entries 0, +-1/2,
    1), so packing launches no library GEMM."""
    cout, cin = w.shape[0], w.shape[1]

    def g_times(t, dim):   # G applied along one 3-tap axis: (t0, (t0 + t1 + t2) / 2, (t0 - t1 + t2) / 2, t2)
        t0, t1, t2 = t.select(dim, 0), t.select(dim, 1), t.select(dim, 2)
        return torch.stack([t0, 0.5 * (t0 + t1 + t2), 0.5 * (t0 - t1 + t2), t2], dim=dim)

    out = torch.empty(cin // 32, 4, 4, cout, 32, device=w.device, dtype=torch.float32)
    step = max(1, (1 << 22) // (cin * 16))  # bound the float64 temporary
    for o0 in range(0, cout, step):
        u = g_times(g_times(w[o0:o0 + step].double(), 2), 3)  # (o, c, i, j) = G g G^T
        u = u.permute(1, 3, 2, 0).reshape(cin // 32, 32, 4, 4, -1).permute(0, 2, 3, 4, 1)  # [c/32][j][i][o][32]
        out[:, :, :, o0:o0 + step] = u.float()
    return out


def pack_conv_weight(w):
    """OIHW (Cout, Cin, 3, 3) -> [Cin/32][9][Cout][32] (see csrc/conv3x3.hip)."""
    cout, cin = w.shape[0], w.shape[1]
    return (w.permute(1, 2, 3, 0).reshape(cin // 32, 32, 9, cout).permute(0, 2, 3, 1).contiguous())


def fold_bn(bn_weight, bn_bias, mean, var, eps):
    scale = bn_weight / torch.sqrt(var + eps)
    return scale.contiguous(), (bn_bias - mean * scale).contiguous()


# Large f32 linear layers (the GRU input projections, the decoder's memory projection) run on the split-bf16 matrix
# path (2^-16 relative operand error, f32 accumulate: the "bf16x3" arithmetic of the conv tier); "f32" keeps the exact
# f32 MFMA GEMM everywhere.
LINEAR_ALGO = os.environ.get("AUDIOCAPTION_LINEAR_ALGO", "bf16x3")
_LINEAR_PACKS = {}   # id(weight tensor) -> (weakref, version, generation, packed weight, ones, zeros)


def pack_linear_weight_bf16x3_frag(w):
    """(N, K) f32 -> split bf16 in the one-tap fragment order of the "gw" kernel: [K/32][1][2 ks][N/32][2][64][8]."""
    n, k = w.shape
    return pack_conv_weight_bf16x3_frag(w.reshape(n, k, 1, 1))


def _linear_pack(w):
    """Packed copy of a weight, cached per live tensor OBJECT (a weak reference guards against a freed tensor's
    address being reused) and invalidated by in-place updates (``_version``) and checkpoint loads."""
    import weakref
    hit = _LINEAR_PACKS.get(id(w))
    if hit is not None and hit[0]() is w and hit[1] == w._version and hit[2] == _lib.param_generation():
        return hit[3:]
    for k in [k for k, v in _LINEAR_PACKS.items() if v[0]() is None]:
        del _LINEAR_PACKS[k]
    with torch.no_grad():
        pack = (pack_linear_weight_bf16x3_frag(w.detach().float()), torch.ones(w.shape[0], device=w.device),
                torch.zeros(w.shape[0], device=w.device))
    _LINEAR_PACKS[id(w)] = (weakref.ref(w), w._version, _lib.param_generation()) + pack
    return pack


def linear(x, w, b=None, relu=False, out=None):
    """y = act(x @ w.T + b); x (M, K) row-major (stride(0) may exceed K), w (N, K)."""
    lib = _lib.load()
    M, K = x.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, device=x.device, dtype=torch.float32)
    # Chosen by the LAYER (its weight matrix), never by the batch: a clip gets the same numbers in a batch of 1 and of
    # 64.  From 1536 x 512 weights up the split path wins at the bench's 1984 rows (39 -> 26 us; K = 2048: 144 -> 77 us).
    if (LINEAR_ALGO == "bf16x3" and N * K >= 1536 * 512 and K % 32 == 0 and N % 64 == 0 and x.stride(0) == K
            and out.stride(0) == N and w.is_contiguous()):
        wfrag, ones, zeros = _linear_pack(w)
        check(lib.ac_linear_bf16x3(ptr(x), ptr(wfrag), ptr(ones), ptr(b if b is not None else zeros), ptr(out), M, N, K,
                                   1 if relu else 0, stream()), "ac_linear_bf16x3")
        return out
    check(lib.ac_linear(ptr(x), ptr(w), ptr(b), ptr(out), M, N, K, x.stride(0), w.stride(0), out.stride(0),
                        1 if relu else 0, stream()), "ac_linear")
    return out


def gru_pack_whh(whh, hidden):
    """nn.GRU weight_hh of both directions [2][3H][H] -> the recurrence kernel's layout [2][H/4][3H][4]."""
    lib = _lib.load()
    whh = f32c(_dev(whh))
    out = torch.empty_like(whh)
    check(lib.ac_gru_pack_whh(ptr(whh), ptr(out), hidden, stream()), "ac_gru_pack_whh")
    return out


def gru_layer(gx, whhT, bhh, lens_i32, B, T, hidden):
    lib = _lib.load()
    out = torch.empty(B, T, 2 * hidden, device=gx.device, dtype=torch.float32)
    check(lib.ac_gru_layer(ptr(gx), ptr(whhT), ptr(bhh), ptr(lens_i32), ptr(out), B, T, hidden, stream()),
          "ac_gru_layer")
    return out


def gru_layer_split(gx, whh, bhh, lens_i32, B, T, hidden, workspace=None):
    """The recurrence with every (clip, direction) split over four 256-thread workgroups (W_hh register resident).  ``whh``: the
    UNPACKED [2][3H][H] weights.  Returns (out, workspace); ``gru_split_error(workspace)`` is its sticky error word."""
    lib = _lib.load()
    need = lib.ac_gru_split_workspace_bytes(B)
    if workspace is None or workspace.numel() * 8 < need or workspace.device != gx.device:
        workspace = torch.zeros((need + 7) // 8, device=gx.device, dtype=torch.int64)
    out = torch.empty(B, T, 2 * hidden, device=gx.device, dtype=torch.float32)
    check(lib.ac_gru_layer_split(ptr(gx), ptr(whh), ptr(bhh), ptr(lens_i32), ptr(out), None, ptr(workspace), B, T, hidden,
                                 stream()), "ac_gru_layer_split")
    return out, workspace


def gru_split_error(workspace, B):
    """The error word of a split-GRU workspace as a 1-element int32 view (device tensor; reading it synchronises)."""
    return workspace.view(torch.int32)[:1]


def mean_with_lens(x, lens_i32, add_max=False):
    lib = _lib.load()
    B, T, C = x.shape
    out = torch.empty(B, C, device=x.device, dtype=torch.float32)
    check(lib.ac_mean_with_lens(ptr(x), ptr(lens_i32), ptr(out), B, T, C, 1 if add_max else 0, stream()),
          "ac_mean_with_lens")
    return out


def add_layernorm(x, y, w, b, out=None):
    lib = _lib.load()
    rows, d = x.shape
    if out is None:
        out = torch.empty(rows, d, device=x.device, dtype=torch.float32)
    check(lib.ac_add_layernorm(ptr(x), ptr(y), ptr(w), ptr(b), ptr(out), rows, d, x.stride(0),
                               y.stride(0) if y is not None else 0, out.stride(0), stream()), "ac_add_layernorm")
    return out


def dropout_(x, n, p, seed, seed_dev=None):
    """In-place counter-hash dropout over the first n floats of x (csrc/train.hip)."""
    lib = _lib.load()
    check(lib.ac_dropout(ptr(x), ptr(x), n, float(p), int(seed), seed_dev, 0, stream()), "ac_dropout")
    return x


def rows_mean_w(x, out, B, Hp, H, W, C):
    lib = _lib.load()
    check(lib.ac_rows_mean_w(ptr(x), ptr(out), B, Hp, H, W, C, stream()), "ac_rows_mean_w")
    return out


def specaug_(x, stripes, fill, B, rows_per_clip, T, n_time=2, n_freq=2):
    """In-place SpecAugment stripes on the bn0-normalised log-mel rows (csrc/train.hip)."""
    lib = _lib.load()
    check(lib.ac_specaug(ptr(x), ptr(stripes), ptr(fill), B, rows_per_clip, T, 64, n_time, n_freq, stream()), "ac_specaug")
    return x


def specaug_stripes(seed, B, T, F=64, time_width=64, time_num=2, freq_width=8, freq_num=2):
    """Stripe draws of torchlibrosa's SpecAugmentation(time_drop_width=64, time_stripes_num=2, freq_drop_width=8,
    freq_stripes_num=2) (cnn_encoder.py:352-354): per clip and stripe a width in [0, drop_width) and a start in
    [0, total - width); all time stripes of the batch first, then all mel stripes.  numpy Generator instead of torch's."""
    import numpy as np
    rng = np.random.default_rng(int(seed) & 0xFFFFFFFFFFFFFFFF)
    out = np.zeros((B, time_num + freq_num, 2), dtype=np.int32)
    for b in range(B):
        for k in range(time_num):
            d = int(rng.integers(0, time_width))
            out[b, k] = (int(rng.integers(0, T - d)), d)
    for b in range(B):
        for k in range(freq_num):
            d = int(rng.integers(0, freq_width))
            out[b, time_num + k] = (int(rng.integers(0, F - d)), d)
    return out
