import torch, time, sys
sys.path.insert(0, "/root/repo")
from audiocaption_amd import kernels as K, build
build.build()
x = torch.zeros(64, 256, device="cuda")
w = torch.randn(256, 256, device="cuda")
y = torch.empty(64, 256, device="cuda")
def timeit(fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); t1 = time.perf_counter()
    return s.elapsed_time(e) / n * 1000, (t1 - t0) / n * 1e6
print("torch add_ tiny      : gpu %.2f us/launch, host wall %.2f us" % timeit(lambda: x.add_(1.0)))
print("ac_linear 64x256x256 : gpu %.2f us/launch, host wall %.2f us" % timeit(lambda: K.linear(x, w, None, out=y)))
ln_w = torch.ones(256, device="cuda"); ln_b = torch.zeros(256, device="cuda")
print("ac_add_layernorm     : gpu %.2f us/launch, host wall %.2f us" % timeit(lambda: K.add_layernorm(x, x, ln_w, ln_b, out=y)))
# graph replay of 100 linears
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): K.linear(x, w, None, out=y)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        for _ in range(100):
            K.linear(x, w, None, out=y)
    torch.cuda.synchronize()
a, b = timeit(lambda: g.replay(), 50)
print("graph of 100 linears : gpu %.2f us/kernel, host wall %.2f us/kernel" % (a / 100, b / 100))
