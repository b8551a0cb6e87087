"""Development probe: time of ac_gemm for the shapes of the training step."""
import ctypes, torch
from audiocaption_amd import _lib, build
build.build()
lib = _lib.load()
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())

def run(M, N, K, mode="fwd", splitk=1, reps=50):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")
    y = torch.zeros(M, N, device="cuda")
    def call():
        if mode == "fwd":
            return lib.ac_gemm(P(x), K, 1, P(w), 1, K, P(y), N, M, N, K, P(b), 0, 0.0, 1, 0.0, 0, None, 0, None, 0, S())
        if mode == "dx":   # y[M][N] = dy[M][K] W[K][N]
            return lib.ac_gemm(P(x), K, 1, P(w), N, 1, P(y), N, M, N, K, None, 0, 0.0, 1, 0.0, 0, None, 0, None, 0, S())
        if mode == "dw":   # y[M][N] += A^T B with rows = K
            return lib.ac_gemm(P(x), 1, M, P(w), N, 1, P(y), N, M, N, K, None, 0, 1.0, splitk, 0.0, 0, None, 0, None, 0, S())
    if mode == "dx":
        w = torch.randn(K, N, device="cuda")
    if mode == "dw":
        x = torch.randn(K, M, device="cuda"); w = torch.randn(K, N, device="cuda")
    for _ in range(3): assert call() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): call()
    g.replay(); torch.cuda.synchronize()
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"{mode:3s} M={M:6d} N={N:5d} K={K:5d} splitk={splitk:2d}: {us:8.2f} us  {2e-6 * M * N * K / us:8.2f} TFLOP/s")

for M in (32, 352, 672):
    for N, K in ((256, 256), (768, 256), (1024, 256), (256, 1024)):
        run(M, N, K)
run(32, 4981, 256)
run(7392, 256, 256, "dx"); run(7392, 1024, 256, "dx"); run(7392, 256, 1024, "dx"); run(20832, 256, 512, "dx")
run(256, 256, 7392, "dw", 8); run(1024, 256, 7392, "dw", 8); run(768, 256, 7392, "dw", 8); run(512, 256, 20832, "dw", 16)
run(1536, 2048, 992, "dw", 3)

print("x W^T shapes (EffB2 late stages, training row space):")
for M, N, K in ((32256, 528, 88), (32256, 720, 120), (32256, 120, 720), (8192, 1248, 208), (8192, 208, 1248), (8192, 2112, 352),
                (8192, 352, 2112), (8192, 1408, 352), (7392, 768, 256), (7392, 1024, 256), (7392, 256, 1024), (20832, 512, 256)):
    run(M, N, K)
