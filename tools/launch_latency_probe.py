import torch, time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiocaption_amd import kernels as K, build
build.build()
dev = "cuda"
def graph_time(fn, per=200, reps=20):
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(per): fn()
        torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1000 / (per * reps)
x1 = torch.zeros(1, 1, 64, device=dev); l1 = torch.ones(1, device=dev, dtype=torch.int32)
print("trivial kernel (mean_with_lens 1x1x64)   : %.2f us/node" % graph_time(lambda: K.mean_with_lens(x1, l1)))
x = torch.randn(64, 256, device=dev); w = torch.randn(256, 256, device=dev); y = torch.empty(64, 256, device=dev)
w2 = torch.randn(256, 1024, device=dev); x2 = torch.randn(64, 1024, device=dev)
print("old skinny linear 64x256x256             : %.2f us/node" % graph_time(lambda: K.linear(x, w, None, out=y)))
print("old skinny linear 64x256x1024            : %.2f us/node" % graph_time(lambda: K.linear(x2, w2, None, out=y)))
g_ = torch.ones(256, device=dev); b_ = torch.zeros(256, device=dev)
print("add_layernorm 64x256                     : %.2f us/node" % graph_time(lambda: K.add_layernorm(x, x, g_, b_, out=y)))
z = torch.zeros(1 << 20, device=dev)
print("torch add_ on 1 elem                     : %.2f us/node" % graph_time(lambda: z[:1].add_(1.0)))
