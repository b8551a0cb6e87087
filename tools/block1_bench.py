"""Development tool: conv_block1 of the f16x2 tier, conv_first + conv2 as two launches vs the fused kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audiocaption_amd import build, kernels as K
build.build()
B, H, Hp, W = 64, 1001, 1024, 64
dev = "cuda:0"
x0 = torch.randn(B * Hp, W, device=dev)
w1 = torch.randn(64, 9, device=dev) * 0.3
w2 = torch.randn(64, 64, 3, 3, device=dev) * (2.0 / 576) ** 0.5
s1, t1 = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.1
s2, t2 = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.1
wp, inv = K.pack_conv_weight_f16x2_frag(w2)
s2 = (s2 * inv).contiguous()
full = torch.empty(B * Hp, W, 64, device=dev, dtype=torch.float16)
o1 = torch.empty(B * Hp // 2, W // 2, 64, device=dev, dtype=torch.float16)
o2 = torch.empty_like(o1)
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
f = t(lambda: K.conv3x3_first(x0, w1, s1, t1, full, B, Hp, H, W))
c = t(lambda: K.conv3x3_bn_relu_f16x2_gw(full, wp, s2, t2, o1, B, Hp, H, W, 64, 64, 1))
g = t(lambda: K.conv3x3_block1_f16x2(x0, w1, s1, t1, wp, s2, t2, o2, B, Hp, H, W))
print(f"conv_first {f:.1f} us + conv2 {c:.1f} us = {f + c:.1f} us; fused {g:.1f} us; equal {torch.equal(o1, o2)}")
