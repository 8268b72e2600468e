"""Diagnostic: kernel time of the projection GEMM shapes in each arithmetic (mdd_diag_gemm_time)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ctc_attention_mispronunciation_amd import _lib
L = _lib.lib()
torch.zeros(1).cuda()
L.mdd_diag_gemm_time.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
shapes = [(128000, 3072, 1952), (128000, 3072, 768), (20480, 3072, 512)]
names = {0: "exact fp32 MFMA", 1: "split-bf16 x3", 3: "f32x6"}
peak = {0: 157.3, 1: 2500.0 / 3, 3: 2500.0 / 6}
for M, N, K in shapes:
    for mode in [int(a) for a in sys.argv[1:]] or [0, 1, 3]:
        ms = C.c_float(0)
        rc = L.mdd_diag_gemm_time(mode, M, N, K, 5, C.byref(ms))
        assert rc == 0, L.mdd_last_error().decode()
        tf = 2.0 * M * N * K / (ms.value * 1e-3) / 1e12
        print("%7d x %4d x %4d  %-16s %8.3f ms  %7.1f TFLOP/s algorithmic  (%.2f of this arithmetic's matrix-core ceiling)" % (M, N, K, names[mode], ms.value, tf, tf / peak[mode]))
