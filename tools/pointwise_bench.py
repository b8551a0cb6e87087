"""Development probe: the two 1x1-convolution kernels on the EfficientNet-B2 layer shapes (batch 128, 10 s clips)."""
import ctypes, torch
from audiocaption_amd import _lib, build
build.build()
lib = _lib.load()
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())
shapes = [(2048000, 16, 32), (2048000, 96, 16), (512000, 24, 96), (512000, 144, 24), (512000, 24, 144), (128000, 48, 144),
          (128000, 288, 48), (128000, 48, 288), (32256, 88, 288), (32256, 528, 88), (32256, 88, 528), (32256, 120, 528),
          (32256, 720, 120), (32256, 120, 720), (8192, 208, 720), (8192, 1248, 208), (8192, 208, 1248), (8192, 352, 1248),
          (8192, 2112, 352), (8192, 352, 2112), (8192, 1408, 352)]
def timeit(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
print(f"{'M':>8} {'N':>5} {'K':>5} {'pointwise us':>13} {'general us':>11} {'GB moved':>9} {'best GB/s':>10}")
for M, N, K in shapes:
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.1; b = torch.randn(N, device="cuda")
    y = torch.empty(M, N, device="cuda")
    t1 = timeit(lambda: lib.ac_pointwise_conv(P(x), P(w), P(b), P(y), M, N, K, 2, 0.0, None, 0, S()))
    t2 = timeit(lambda: lib.ac_gemm(P(x), K, 1, P(w), 1, K, P(y), N, M, N, K, P(b), 2, 0.0, 1, 0.0, 0, None, 0, None, 0, S()))
    gb = 4e-9 * M * (N + K)
    print(f"{M:8d} {N:5d} {K:5d} {t1:13.1f} {t2:11.1f} {gb:9.3f} {gb / min(t1, t2) * 1e6:10.0f}")
