"""Diagnostic: register-staged vs LDS-DMA GEMM kernel."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ctc_attention_mispronunciation_amd import _lib
torch.zeros(1).cuda()
L = _lib.lib(); L.mdd_diag_gemm.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_float)]
for (M, N, K) in ((64000, 3072, 1952), (64000, 3072, 768)):
    for abl, name in ((0, 'register-staged'), (8, 'LDS-DMA 128x128'), (9, 'LDS-DMA 256x256'), (10, 'LDS-DMA 256 16x16x32')):
        ms = C.c_float(0); L.mdd_diag_gemm(M, N, K, abl, 5, C.byref(ms))
        print('K=%d %-16s %.3f ms  %.0f TF algorithmic' % (K, name, ms.value, 2.0 * M * N * K / ms.value / 1e9))
