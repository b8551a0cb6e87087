#!/usr/bin/env python
"""Development tool: phase timestamps inside dec_wide_gemm_kernel (csrc/decoder_wide.hip built with -DAC_WIDE_STAMPS) and the
kernel's duration by HIP events, per projection shape of a decode step.
    python tools/wide_stamps.py --build     # here (no GPU): tools/bin/libwide_stamps.so
    python tools/wide_stamps.py [ROWS]      # on the GPU box"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SO = os.path.join(ROOT, "tools", "bin", "libwide_stamps.so")
if "--build" in sys.argv:
    from audiocaption_amd import build as B
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    cmd = [B._hipcc(), "-x", "hip", os.path.join(B.CSRC, "decoder_wide.hip"), "-shared", "-o", SO, "-DAC_WIDE_STAMPS"] + B.FLAGS + B.NO_PACKED_F32
    subprocess.check_call(cmd)
    print("built", SO)
    sys.exit(0)
import torch
from audiocaption_amd import _lib

M = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lib = ctypes.CDLL(SO)
for n in ("ac_dec_wide_packed_floats", "ac_dec_wide_pack", "ac_dec_wide_gemm"):
    getattr(lib, n).restype, getattr(lib, n).argtypes = _lib.SIGNATURES[n]
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
names = ["start", "loads requested", "planes written", "barrier", "products", "-", "K parts met", "end"]
for pro, N, K, ntb, split in ((0, 256, 256, 1, 0), (2, 256, 256, 1, 0), (2, 768, 256, 1, 0), (2, 1024, 256, 1, 1), (0, 256, 1024, 1, 0),
                              (2, 4368, 256, 2, 0), (2, 4368, 256, 4, 0)):
    X = torch.randn(M, K, device="cuda")
    Y2 = torch.randn(M, K, device="cuda")
    g, b = torch.ones(K, device="cuda"), torch.zeros(K, device="cuda")
    W = torch.randn(N, K, device="cuda") / 16
    pk = torch.empty(lib.ac_dec_wide_packed_floats(N, K), device="cuda")
    assert lib.ac_dec_wide_pack(P(W), K, N, K, P(pk), S()) == 0
    xs = torch.empty(lib.ac_dec_wide_packed_floats(M, K), device="cuda")
    assert lib.ac_dec_wide_pack(P(X), K, M, K, P(xs), S()) == 0
    Y = torch.empty(max(M * N, lib.ac_dec_wide_packed_floats(M, N) if N % 16 == 0 else 0), device="cuda")
    bias = torch.zeros(N, device="cuda")
    call = lambda: lib.ac_dec_wide_gemm(pro, P(xs if pro == 0 else X), K, P(Y2), K, P(g), P(b), None, 0, 0, None, None, 0.0, None, 0,
                                        P(pk), P(bias), P(Y), N, M, N, K, 0, ntb, split, S())
    for _ in range(3):
        assert call() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n):
        call()
    e1.record()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    assert lib.ac_wide_stamps_read(buf) == 0
    t = [int(x) for x in buf]
    prev, ph = t[0], []
    for k in range(1, 8):
        if t[k] >= t[0] and t[k] - t[0] < 100000:
            ph.append(f"{names[k]} +{(t[k] - prev) / 100:.2f}")
            prev = t[k]
    print(f"pro {pro} M {M} N {N} K {K} ntb {ntb} split {split}: {1e3 * e0.elapsed_time(e1) / n:6.2f} us per back-to-back launch; "
          f"last workgroup {(t[7] - t[0]) / 100:.2f} us = " + "  ".join(ph))
    buf = (ctypes.c_ulonglong * 16)()
