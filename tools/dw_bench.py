#!/usr/bin/env python
"""Development tool: microseconds per launch of the EfficientNet-B2 stride-1 depthwise layers at 128 clips, rows-in-registers
form (default) vs the sliding-along-mel form (AUDIOCAPTION_DW_ROWS_KERNEL=0; the env var is read once per process: run twice)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audiocaption_amd import _lib
lib = _lib.load()
P = lambda t: ctypes.c_void_p(t.data_ptr())
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
B = 128
print("rows kernel:", os.environ.get("AUDIOCAPTION_DW_ROWS_KERNEL", "1"))
for (k, F, T, C) in [(3, 32, 501, 32), (3, 32, 501, 16), (3, 16, 251, 144), (3, 16, 251, 96), (5, 8, 126, 288), (3, 4, 63, 528), (5, 4, 63, 528), (5, 4, 63, 720), (5, 2, 32, 1248),
                     (3, 2, 32, 1248), (3, 2, 32, 2112)]:
    x = torch.randn(B, T, F, C, device="cuda")
    w = torch.randn(k, k, C, device="cuda") * 0.3
    sc, sh = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1
    y = torch.empty(B, T, F, C, device="cuda")
    pool = torch.zeros(B, C, device="cuda")
    pad = (k - 1) // 2
    def run():
        assert lib.ac_effnet_depthwise(P(x), P(w), P(sc), P(sh), P(y), P(pool), 1.0, B, T, F, C, k, 1, pad, pad, S()) == 0
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    gb = 2 * x.numel() * 4 / 1e9
    print(f"k{k} F{F:2d} T{T:3d} C{C:4d}: {us:7.1f} us  {gb / us * 1e6 / 1e3:5.2f} TB/s")
