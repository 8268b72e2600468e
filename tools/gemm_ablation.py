"""Diagnostic: time gemm_bf16x3_kernel and ablations (bit0 no MFMA, bit1 no global loads, bit2 no LDS stores)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ctc_attention_mispronunciation_amd import _lib
torch.zeros(1).cuda()
L = _lib.lib()
L.mdd_diag_gemm.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_float)]
for (M, N, K) in ((64000, 3072, 1952), (64000, 3072, 768)):
    fl = 2.0 * M * N * K
    for abl, name in ((0, 'full'), (1, 'no MFMA'), (2, 'no global loads'), (6, 'no loads+stores'), (7, 'LDS reads only'), (3, 'no MFMA, no loads')):
        ms = C.c_float(0)
        L.mdd_diag_gemm(M, N, K, abl, 5, C.byref(ms))
        print('M=%d N=%d K=%d  %-20s %.3f ms  (%.0f TF algorithmic if full)' % (M, N, K, name, ms.value, fl / ms.value / 1e9))
