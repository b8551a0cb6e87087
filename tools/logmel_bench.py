"""Development: time ac_logmel of the in-tree library, optionally beside another build of csrc/logmel.hip
(``--other path/to/lib.so`` exporting ac_logmel), at the bench workload (64 ten-second clips at 32 kHz) and the EffB2 one
(16 kHz, n_fft 512).  Prints microseconds per launch (HIP events over ``--iters`` launches) and the max |diff| between the
two builds."""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--other", default=None)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--batch", type=int, default=64)
    args = ap.parse_args()
    from audiocaption_amd import _lib, build
    from audiocaption_amd.kernels import ptr, stream
    from audiocaption_amd.mel import MelTables
    build.build()
    libs = {"tree": _lib.load()}
    if args.other:
        o = ctypes.CDLL(os.path.abspath(args.other))
        o.ac_logmel.restype, o.ac_logmel.argtypes = _lib.SIGNATURES["ac_logmel"]
        libs["other"] = o
    for name, sr, L in (("cnn14 32 kHz", 32000, 320000), ("effb2 16 kHz", 16000, 160000)):
        if sr == 32000:
            tables = MelTables(32000, 1024, 320, 50.0, 14000.0, 64, "slaney", "slaney", "cuda")
        else:
            tables = MelTables(16000, 512, 160, 0.0, 8000.0, 64, None, "htk", "cuda")
        wav = torch.randn(args.batch, L, device="cuda") * 0.1
        T = L // tables.hop + 1
        Hp = (T + 31) // 32 * 32 + 32
        sc, sh = torch.rand(64, device="cuda") + 0.5, torch.randn(64, device="cuda")
        outs = {}
        for key, lib in libs.items():
            out = torch.zeros(args.batch * Hp, 64, device="cuda")

            def run():
                rc = lib.ac_logmel(ptr(wav), args.batch, L, tables.n_fft, tables.hop, ptr(tables.window), ptr(tables.twiddle),
                                   ptr(tables.melfb), ptr(tables.mel_lo), ptr(tables.mel_hi), ptr(sc), ptr(sh), ptr(out), Hp,
                                   Hp * 64, 64, 1, stream())
                assert rc == 0, rc
            for _ in range(10):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / args.iters
            gb = (wav.numel() + out.numel()) * 4 / 1e9
            print(f"[{name}] {key:6s} {us:8.1f} us / launch   {gb / us * 1e6:7.0f} GB/s of samples + rows", flush=True)
            outs[key] = out
        if "other" in outs:
            d = (outs["tree"] - outs["other"]).abs()
            print(f"[{name}] max|tree - other| {float(d.max()):.3e} dB  (equal: {bool(torch.equal(outs['tree'], outs['other']))})")


if __name__ == "__main__":
    main()
