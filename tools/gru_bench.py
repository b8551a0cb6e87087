"""Development tool: the GRU recurrence kernel alone (B=64, T=31, H=256), per layer."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audiocaption_amd import build, kernels as K
build.build()
B, T, Hh = 64, 31, 256
dev = "cuda:0"
g = torch.Generator(device="cpu").manual_seed(0)
gx = torch.randn(B * T, 6 * Hh, generator=g).to(dev)
whh = (torch.randn(2, 3 * Hh, Hh, generator=g) / 16).to(dev)
bhh = torch.randn(2, 3 * Hh, generator=g).to(dev)
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
lens[1::3] = 20
pk = K.gru_pack_whh(whh, Hh)
out = K.gru_layer(gx, pk, bhh, lens, B, T, Hh)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    out = K.gru_layer(gx, pk, bhh, lens, B, T, Hh)
b.record(); torch.cuda.synchronize()
print(f"gru_layer {a.elapsed_time(b) / 20 * 1e3:.1f} us; checksum {float(out.double().sum()):.9f} absmax {float(out.abs().max()):.6f}")

out2, ws = K.gru_layer_split(gx, whh, bhh, lens, B, T, Hh)
torch.cuda.synchronize()
a.record()
for _ in range(20):
    out2, ws = K.gru_layer_split(gx, whh, bhh, lens, B, T, Hh, ws)
b.record(); torch.cuda.synchronize()
print(f"gru_layer_split {a.elapsed_time(b) / 20 * 1e3:.1f} us; max|diff| vs single-workgroup kernel "
      f"{float((out2 - out).abs().max()):.2e}; error word {int(K.gru_split_error(ws, B).item())}")
for Bx, Tx in ((3, 7), (130, 93)):
    gxx = torch.randn(Bx * Tx, 6 * Hh, device=dev)
    lx = torch.randint(1, Tx + 1, (Bx,), device=dev, dtype=torch.int32)
    o1 = K.gru_layer(gxx, pk, bhh, lx, Bx, Tx, Hh)
    o2, w2 = K.gru_layer_split(gxx, whh, bhh, lx, Bx, Tx, Hh)
    torch.cuda.synchronize()
    print(f"B={Bx} T={Tx}: max|diff| {float((o2 - o1).abs().max()):.2e} error {int(K.gru_split_error(w2, Bx).item())}")
