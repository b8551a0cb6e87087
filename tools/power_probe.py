"""Development tool: the SAME conv launches on inputs of different bit activity.  The kernels do the same work whatever the
data; the part clocks them by power, so all-zero / half-zero / dense-random inputs give different times.

    python tools/power_probe.py      (GPU box)   ->  profiles/r05_data_dependent_clock.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from audiocaption_amd import kernels as K

dev = "cuda"


def timed(f, n=10):
    f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1000


def inputs(shape):
    g = torch.Generator(device=dev).manual_seed(1)
    r = torch.randn(shape, device=dev, generator=g)
    return (("zeros", torch.zeros(shape, device=dev)), ("constant 1.0", torch.ones(shape, device=dev)),
            ("relu(randn): half zeros", r.clamp(min=0)), ("randn", r))


print("# same kernel, same shapes, same weights; only the input values differ (us per launch, 10 launches back to back)")
B, Hp, H = 64, 1024, 1001
w1 = torch.randn(64, 9, device=dev) * 0.3
w2 = torch.randn(64, 64, 3, 3, device=dev) * 0.06
s1, t1, s2, t2 = (torch.rand(64, device=dev) + 0.5 for _ in range(4))
wp = K.pack_conv_weight_wino43_frag(w2)
out = torch.empty(B * Hp // 2, 32, 64, device=dev)
print(f"conv block 1, one kernel (conv1 on the matrix cores), {B} x {H} x 64 log-mel values")
for name, x in inputs((B * Hp, 64)):
    t = timed(lambda: K.conv3x3_block1_wino43(x, w1, s1, t1, wp, s2, t2, out, B, Hp, H))
    print(f"  {name:26s} {t:7.1f}")
for (W, Cin, Cout, Hq, mode) in ((16, 256, 256, 256, 1), (4, 1024, 1024, 64, 1)):
    w = torch.randn(Cout, Cin, 3, 3, device=dev) * (2.0 / (9 * Cin)) ** 0.5
    wf = K.pack_conv_weight_wino43_frag(w)
    sc, sh = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev) * 0.1
    o = torch.empty(B * Hq // 2 * (W // 2) * Cout, device=dev)
    print(f"conv3x3_w4_kernel<POOL>, {Cin} -> {Cout} channels, W = {W}, {B} x {Hq} rows")
    for name, x in inputs((B * Hq, W, Cin)):
        t = timed(lambda: K.conv3x3_bn_relu_wino43(x, wf, sc, sh, o, B, Hq, Hq - 4, W, Cin, Cout, mode))
        print(f"  {name:26s} {t:7.1f}")
