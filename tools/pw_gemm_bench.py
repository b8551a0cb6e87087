"""Development probe: EfficientNet-B2's matrix-bound 1x1 convolutions at 128 clips x 10 s - ac_gemm_bf16x3 (both operands
split per tile) vs ac_pw_gemm_bf16x3 (weights pre-split in fragment order, activation-stationary)."""
import ctypes

import torch

from audiocaption_amd import _lib, build

build.build()
lib = _lib.load()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
shapes = [("s4 expand", 32256, 528, 88, 2), ("s4 project", 32256, 88, 528, 0), ("s5 expand", 32256, 720, 120, 2),
          ("s5 project", 32256, 120, 720, 0), ("s6 expand", 8192, 1248, 208, 2), ("s6 project", 8192, 208, 1248, 0),
          ("s7 expand", 8192, 2112, 352, 2), ("s7 project", 8192, 352, 2112, 0), ("head", 8192, 1408, 352, 2),
          ("s3 project", 129024, 48, 288, 0), ("s3 expand", 129024, 288, 48, 2), ("s2 project", 514048, 24, 144, 0),
          ("s2 expand", 514048, 144, 24, 2), ("tf qkv", 1344, 768, 256, 0), ("tf ffn1", 1344, 1024, 256, 1),
          ("tf ffn2", 1344, 256, 1024, 0), ("tf cls", 1344, 4368, 256, 0),
          ("tr qkv", 7392, 768, 256, 1), ("tr out", 7392, 256, 256, 1), ("tr ffn1", 7392, 1024, 256, 1),
          ("tr ffn2", 7392, 256, 1024, 1), ("tr kv", 992, 512, 256, 1), ("tr gru0", 992, 1536, 2048, 1),
          ("tr gru1", 992, 1536, 512, 1), ("tr cls", 672, 4984, 256, 1),
          ("inf gru0", 1984, 1536, 2048, 1), ("inf gru1", 1984, 1536, 512, 1), ("inf mem", 1984, 256, 2048, 1)]


def timeit(fn, reps=20):
    for _ in range(3):
        assert fn() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for name, M, N, K, act in shapes:
    x = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda")
    y1, y2 = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
    gate = torch.rand((M + 63) // 64, K, device="cuda") if act == 0 else None
    wfrag = torch.empty(lib.ac_pw_gemm_packed_bytes(N, K), device="cuda", dtype=torch.uint8)
    assert lib.ac_pw_gemm_pack(P(w), P(wfrag), N, K, S()) == 0
    t1 = timeit(lambda: lib.ac_gemm_bf16x3(P(x), K, 1, P(w), 1, K, P(y1), N, M, N, K, P(b), act, 0.0, 1, 0.0, 0, None, 0,
                                           P(gate), 64, S()))
    t2 = timeit(lambda: lib.ac_pw_gemm_bf16x3(P(x), P(wfrag), P(b), P(y2), M, N, K, act, 0.0, P(gate), 64, S()))
    d = float((y1 - y2).abs().max()) / float(y1.abs().max())
    gf = 2.0 * M * N * K / 1e9
    print(f"{name:11s} {M:6d} x {N:5d} x {K:5d}: gemm_bf16x3 {t1:7.1f} us {gf / t1 * 1e3:6.1f} TF | pw {t2:7.1f} us "
          f"{gf / t2 * 1e3:6.1f} TF | diff {d:.1e}")
