"""Development probe: the decode classifier product at beam-search row counts: [R x 256] x [256 x 4368]^T, exact-f32 tiled GEMM
(ac_gemm) and split-bf16 GEMM (ac_gemm_bf16x3) against the decode chain's own projection time (rocprofv3: dec_gemm_kernel<2>)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audiocaption_amd import _lib
lib = _lib.load()
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps
for M in (256, 384, 768, 1536):
    for N, K in ((4368, 256), (768, 256), (1024, 256), (256, 1024)):
        x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.1; b = torch.randn(N, device="cuda")
        y = torch.empty(M, N, device="cuda")
        t1 = timeit(lambda: lib.ac_gemm(P(x), K, 1, P(w), 1, K, P(y), N, M, N, K, P(b), 0, 0.0, 1, 0.0, 0, None, 0, None, 0, S()))
        t2 = timeit(lambda: lib.ac_gemm_bf16x3(P(x), K, 1, P(w), 1, K, P(y), N, M, N, K, P(b), 0, 0.0, 1, 0.0, 0, None, 0, None, 0, S()))
        print(f"M {M:5d} N {N:5d} K {K:5d}: ac_gemm {t1:6.1f} us   ac_gemm_bf16x3 {t2:6.1f} us")
