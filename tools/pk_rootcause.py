"""Development tool: the discriminating experiment for the packed-f32 miscompute of the per-row decode kernels
(audiocaption_amd/build.py NO_PACKED_F32).

    python tools/pk_rootcause.py --build     # here (no GPU): four libraries in tools/bin/ that differ in csrc/decoder.hip only
    python tools/pk_rootcause.py             # on the GPU box: tools/corunner_probe.py against each of them

  scalar   : the shipped build (-packed-fp32-ops): baseline, expected clean
  pk       : decoder.hip WITH packed-f32 VALU instructions (v_pk_fma_f32 from hipcc's SLP vectoriser): the failing build
  pk_wait  : pk + an explicit s_waitcnt vmcnt(0) lgkmcnt(0) between row_gemv256's loads and its multiply-adds
  pk_fence : pk_wait + a workgroup-scope fence
  pk_nop   : pk_wait + sixteen s_nop issue slots before the arithmetic
  pk_mov   : pk_wait + every loaded row copied through a plain v_mov_b32 first (the packed instructions never read a register
             written by a load)

If pk fails and pk_wait is clean, the packed instructions consumed weight rows before the loads had landed: a wait-count
placement problem (the compiler's partial vmcnt(N) for a packed consumer), not a hardware race between kernels."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BIN = os.path.join(ROOT, "tools", "bin")
VARIANTS = {"scalar": (True, []), "pk": (False, []), "pk_wait": (False, ["-DDEC_PK_PROBE=1"]), "pk_fence": (False, ["-DDEC_PK_PROBE=2"]),
            "pk_nop": (False, ["-DDEC_PK_PROBE=3"]), "pk_mov": (False, ["-DDEC_PK_PROBE=4"])}


def build():
    from audiocaption_amd import build as B
    B.build()
    os.makedirs(BIN, exist_ok=True)
    objs = [os.path.join(B.HERE, "build", s.replace(".hip", ".o")) for s in B.SOURCES if s != "decoder.hip"]
    for name, (scalar, flags) in VARIANTS.items():
        obj = os.path.join(BIN, f"decoder_{name}.o")
        cmd = [B._hipcc(), "-x", "hip", "-c", os.path.join(B.CSRC, "decoder.hip"), "-o", obj] + B.FLAGS + flags + \
            (B.NO_PACKED_F32 if scalar else [])
        subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
        so = os.path.join(BIN, f"libpk_{name}.so")
        subprocess.check_call([B._hipcc(), "-shared", "-fPIC", "--offload-arch=gfx950", "-o", so] + objs + [obj])
        dis = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--offloading", obj], capture_output=True, text=True)
        asm = subprocess.run([B._hipcc(), "-x", "hip", "-S", "--cuda-device-only", os.path.join(B.CSRC, "decoder.hip"), "-o", "-"] +
                             B.FLAGS + flags + (B.NO_PACKED_F32 if scalar else []), capture_output=True, text=True).stdout
        print(f"built {so}: v_pk_fma_f32 x {asm.count('v_pk_fma_f32')}, v_pk_mul_f32 x {asm.count('v_pk_mul_f32')}, "
              f"v_pk_add_f32 x {asm.count('v_pk_add_f32')} in decoder.hip", flush=True)
        del dis


def main():
    if "--build" in sys.argv:
        return build()
    for name in VARIANTS:
        so = os.path.join(BIN, f"libpk_{name}.so")
        print(f"== {name}", flush=True)
        env = dict(os.environ, AUDIOCAPTION_HIP_LIB=so)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "corunner_probe.py")], env=env, capture_output=True, text=True)
        print(r.stdout.strip() or r.stderr[-800:], flush=True)


if __name__ == "__main__":
    main()
