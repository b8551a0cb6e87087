#!/usr/bin/env python
"""Development tool: is conv3x3_skinny deterministic under load?  Repeats every geometry many times, alone and beside a
bandwidth-heavy kernel on another stream, and compares the outputs bit for bit."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audiocaption_amd import kernels as K

torch.manual_seed(0)
bad = 0
for (B, Hp, H, W, Cin, Cout, mode) in [(2, 12, 9, 2, 1024, 2048, 0), (2, 12, 9, 2, 2048, 2048, 0), (2, 24, 18, 4, 512, 1024, 0),
                                        (2, 24, 18, 4, 1024, 1024, 1), (4, 32, 31, 2, 2048, 2048, 0), (1, 32, 31, 2, 1024, 2048, 0)]:
    x = torch.randn(B * Hp, W, Cin, device="cuda")
    x.view(B, Hp, W, Cin)[:, H:] = 0
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") * math.sqrt(2.0 / (9 * Cin))
    wp = K.pack_conv_weight_bf16x3_frag(w)
    sc, sh = torch.rand(Cout, device="cuda") + 0.5, torch.randn(Cout, device="cuda") * 0.1
    ws = torch.empty(K.skinny_workspace_floats(B, Hp, W, Cin, Cout), device="cuda")
    shape = (B * Hp // 2, W // 2, Cout) if mode == 1 else (B * Hp, W, Cout)
    ref = torch.empty(shape, device="cuda")
    K.conv3x3_bn_relu_skinny(x, wp, sc, sh, ref, B, Hp, H, W, Cin, Cout, mode, ws)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    big = torch.randn(64 << 20, device="cuda")
    n_bad = 0
    for it in range(300):
        if it % 2:
            with torch.cuda.stream(side):
                big.mul_(1.0001)
        ws.fill_(float("nan"))          # a slice nobody wrote would show
        out = torch.empty(shape, device="cuda")
        K.conv3x3_bn_relu_skinny(x, wp, sc, sh, out, B, Hp, H, W, Cin, Cout, mode, ws)
        if not torch.equal(out, ref):
            n_bad += 1
            d = (out - ref).abs()
            if n_bad <= 3:
                print("  mismatch", it, float(d.max()), int((d > 0).sum()), "nan" if torch.isnan(out).any() else "")
    torch.cuda.synchronize()
    print((B, Hp, H, W, Cin, Cout, mode), "mismatches", n_bad, "of 300")
    bad += n_bad
print("TOTAL", bad)
