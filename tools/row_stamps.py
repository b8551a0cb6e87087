"""Development: where the 21 us of dec_row2_kernel go.  Builds csrc/decoder.hip with -DAC_ROW_STAMPS into a side library
(here, no GPU: ``--build``), then (on the GPU box) runs one blocking greedy decode of 64 clips and prints the phase
timestamps (100 MHz clock) workgroup 0 of the LAST launch left behind: self attention, out-projection, LayerNorm, cross
query projection, cross attention, out-projection, LayerNorm."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "tools", "bin", "libdec_stamps.so")


def build():
    from audiocaption_amd import build as B
    B.build()
    obj = os.path.join(ROOT, "tools", "bin", "decoder_stamps.o")
    cmd = [B._hipcc(), "-x", "hip", "-c", os.path.join(B.CSRC, "decoder.hip"), "-o", obj, "-DAC_ROW_STAMPS"] + B.FLAGS + B.NO_PACKED_F32
    subprocess.check_call(cmd)
    objs = [os.path.join(B.HERE, "build", s.replace(".hip", ".o")) for s in B.SOURCES if s != "decoder.hip"] + [obj]
    subprocess.check_call([B._hipcc(), "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs)
    print(LIB)


def main():
    if "--build" in sys.argv:
        return build()
    os.environ["AUDIOCAPTION_HIP_LIB"] = LIB
    import time
    import torch
    import audiocaption_amd as A
    from audiocaption_amd import _lib, procedural as P
    B = int(os.environ.get("ROWS", "64"))
    dec = A.TransformerDecoder(emb_dim=256, vocab_size=4368, fc_emb_dim=512, attn_emb_dim=512, dropout=0.2, nlayers=2)
    dec.load_state_dict(P.to_torch(P.decoder_state("", 4368)))
    dec = dec.eval().cuda()
    attn = torch.randn(B, 31, 512, device="cuda")
    lens = torch.full((B,), 31)
    for _ in range(3):
        dec.greedy(attn, lens, 20, 1, 2, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        dec.greedy(attn, lens, 20, 1, 2, 0)
    torch.cuda.synchronize()
    print(f"B={B}: greedy decode {1e3 * (time.perf_counter() - t0) / 20:.3f} ms per batch")
    buf = (ctypes.c_ulonglong * 16)()
    lib = _lib.load()
    lib.ac_row_stamps_read.restype = ctypes.c_int
    lib.ac_row_stamps_read.argtypes = [ctypes.c_void_p]
    assert lib.ac_row_stamps_read(ctypes.cast(buf, ctypes.c_void_p)) == 0
    t = [buf[i] for i in range(8)]
    names = ["self attention", "out-projection", "LayerNorm 1", "cross query projection", "cross attention", "out-projection 2", "LayerNorm 2 + store"]
    for n, a, b in zip(names, t[:-1], t[1:]):
        print(f"{n:26s} {(b - a) / 100.0:6.2f} us")
    print(f"{'total inside the kernel':26s} {(t[7] - t[0]) / 100.0:6.2f} us")


if __name__ == "__main__":
    main()
