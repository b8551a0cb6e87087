"""Development probe: the streaming exact-f32 1x1 kernel (ac_pointwise_conv) vs ac_pw_gemm_bf16x3 on the early,
HBM-bound EfficientNet-B2 layers."""
import ctypes

import torch

from audiocaption_amd import _lib, build

build.build()
lib = _lib.load()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, M, N, K, act in [("s3 expand", 129024, 288, 48, 2), ("s3 project", 129024, 48, 288, 0), ("s2 expand", 514048, 144, 24, 2),
                           ("s2 project", 514048, 24, 144, 0), ("s1 project", 2052096, 16, 32, 0)]:
    x = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda")
    y1, y2 = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
    gate = torch.rand((M + 63) // 64, K, device="cuda") if act == 0 else None
    wfrag = torch.empty(lib.ac_pw_gemm_packed_bytes(N, K), device="cuda", dtype=torch.uint8)
    assert lib.ac_pw_gemm_pack(P(w), P(wfrag), N, K, S()) == 0
    res = []
    for fn in (lambda: lib.ac_pointwise_conv(P(x), P(w), P(b), P(y1), M, N, K, act, 0.0, P(gate), 64, S()),
               lambda: lib.ac_pw_gemm_bf16x3(P(x), P(wfrag), P(b), P(y2), M, N, K, act, 0.0, P(gate), 64, S())):
        for _ in range(3):
            assert fn() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 100)
    gb = (M * K + M * N) * 4 / 1e9
    print(f"{name:11s} {M:8d} x {N:4d} x {K:4d}: pointwise {res[0]:7.1f} us ({gb / res[0] * 1e3:5.2f} TB/s) | pw gemm {res[1]:7.1f} us "
          f"({gb / res[1] * 1e3:5.2f} TB/s) | diff {float((y1 - y2).abs().max()) / float(y1.abs().max()):.1e}")
