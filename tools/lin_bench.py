"""Development tool: the inference path's large linear layers, exact-f32 GEMM vs the split-bf16 one-tap conv instance
vs the training GEMM (ac_gemm).  `python tools/lin_bench.py`"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audiocaption_amd import _lib, build, kernels as K
build.build(); lib=_lib.load()
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)*1e3/reps
def lin(algo, *a, **k):
    saved, K.LINEAR_ALGO = K.LINEAR_ALGO, algo
    try: return K.linear(*a, **k)
    finally: K.LINEAR_ALGO = saved
for M,N,Kd in ((1984,1536,2048),(1984,1536,512),(1984,256,512),(1984,512,256),(3968,512,512),(64,2048,2048)):
    x=torch.randn(M,Kd,device="cuda"); w=torch.randn(N,Kd,device="cuda"); b=torch.randn(N,device="cuda"); y=torch.empty(M,N,device="cuda")
    a=t(lambda: lin("f32",x,w,b,out=y))
    c=t(lambda: lin("bf16x3",x,w,b,out=y))
    g=t(lambda: lib.ac_gemm(P(x),Kd,1,P(w),1,Kd,P(y),N,M,N,Kd,P(b),0,0.0,1,0.0,0,None,0,None,0,S()))
    print(M,N,Kd, f"f32 {a:.1f} us ({2e-6*M*N*Kd/a:.1f} TF)  bf16x3 {c:.1f} us ({2e-6*M*N*Kd/c:.1f} TF)  ac_gemm {g:.1f} us ({2e-6*M*N*Kd/g:.1f} TF)")
