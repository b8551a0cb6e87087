"""Development tool: conv block 1 (one kernel) on a Clotho-shaped batch with and without the dead-row skip."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from audiocaption_amd import kernels as K
from audiocaption_amd.cnn_encoder import rows_needed

B, dev = 32, "cuda"
rng = np.random.default_rng(0)
dur = rng.uniform(15.0, 30.0, B)
H = int(dur.max() * 100) + 1
Hp = (H + 8 + 31) & ~31
frames = torch.tensor([(int(d * 100) + 1) // 32 for d in dur], dtype=torch.int32, device=dev)
x0 = torch.randn(B * Hp, 64, device=dev)
w1 = torch.randn(64, 9, device=dev) * 0.3
w2 = torch.randn(64, 64, 3, 3, device=dev) * 0.06
s1, t1, s2, t2 = (torch.rand(64, device=dev) + 0.5 for _ in range(4))
wp = K.pack_conv_weight_wino43_frag(w2)
out = torch.empty(B * Hp // 2, 32, 64, device=dev)
mul, add = rows_needed(1, 2)
live = float(((mul * frames + add).clamp(max=H)).sum()) / (B * H)
print(f"H {H} Hp {Hp}; rows needed / rows: {live:.3f}")
xz = x0.clone().view(B, Hp, 64)
for b in range(B):
    xz[b, int(dur[b] * 100) + 1:] = 0
xz = xz.view(B * Hp, 64)
for name, x0 in (("random everywhere", x0), ("zero padding", xz), ("all zeros", torch.zeros_like(xz))):
  print(name)
  for need in (None, (frames, mul, add)):
    for conv1 in ("mfma", "valu"):
        f = lambda x0=x0: K.conv3x3_block1_wino43(x0, w1, s1, t1, wp, s2, t2, out, B, Hp, H, need=need, conv1=conv1)
        f(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): f()
        b.record(); torch.cuda.synchronize()
        print(f"  need {'yes' if need else 'no '} conv1 {conv1}: {a.elapsed_time(b) * 100:.0f} us")
