"""Development tool: ac_gemm (exact f32) vs ac_gemm_bf16x3 on the training step's shapes."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audiocaption_amd import _lib, build
build.build()
lib = _lib.load()
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())
R = 7392
shapes = [("fwd x w^T", R, 768, 256, "nt"), ("fwd ffn1", R, 1024, 256, "nt"), ("fwd ffn2", R, 256, 1024, "nt"),
          ("dgrad qkv", R, 256, 768, "nn"), ("dgrad ffn2", R, 1024, 256, "nn"), ("dgrad ffn1", R, 256, 1024, "nn"),
          ("wgrad qkv", 768, 256, R, "tn"), ("wgrad ffn1", 1024, 256, R, "tn"), ("wgrad kv", 512, 256, 20832, "tn"),
          ("tf qkv", 1344, 768, 256, "nt"), ("tf cls", 1344, 4368, 256, "nt")]
for name, M, N, K, lay in shapes:
    if lay == "nt":
        A, B = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
        sa, sb = (K, 1), (1, K)
    elif lay == "nn":
        A, B = torch.randn(M, K, device="cuda"), torch.randn(K, N, device="cuda")
        sa, sb = (K, 1), (N, 1)
    else:
        A, B = torch.randn(K, M, device="cuda"), torch.randn(K, N, device="cuda")
        sa, sb = (1, M), (N, 1)
    C = torch.zeros(M, N, device="cuda")
    sk = 1
    if lay == "tn":
        blocks = ((M + 63) // 64) * ((N + 63) // 64)
        sk = max(1, min(512 // blocks, K // 128))
    beta = 1.0 if sk > 1 else 0.0
    res = []
    for fn, extra in ((lib.ac_gemm, (None, 0)), (lib.ac_gemm_bf16x3, (None, 0))):
        call = lambda: fn(P(A), sa[0], sa[1], P(B), sb[0], sb[1], P(C), N, M, N, K, None, 0, beta, sk, 0.0, 0, None, 0, *extra, S())
        for _ in range(3):
            assert call() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            call()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        res.append((us, 2.0 * M * N * K / us / 1e6))
    print(f"{name:12s} {M:6d} x {N:5d} x {K:6d} splitk {sk:3d}: f32 {res[0][0]:7.1f} us {res[0][1]:6.1f} TF | bf16x3 {res[1][0]:7.1f} us {res[1][1]:6.1f} TF")
