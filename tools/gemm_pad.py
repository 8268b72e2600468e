"""Diagnostic: does the leading dimension (L2 channel mapping of the row stride) matter for the GEMM?"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    import ctypes as C, torch
    from ctc_attention_mispronunciation_amd import _lib
    torch.zeros(1).cuda()
    L = _lib.lib(); L.mdd_diag_gemm.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_float)]
    for (M, N, K) in ((64000, 3072, 1952), (64000, 3072, 768)):
        for abl in (0, 1):
            ms = C.c_float(0); L.mdd_diag_gemm(M, N, K, abl, 5, C.byref(ms))
            print('pad=%s K=%d abl=%d  %.3f ms' % (os.environ.get('MDD_DIAG_PAD', '0'), K, abl, ms.value))
else:
    for pad in (0, 8, 64, 72, 136):
        subprocess.call([sys.executable, __file__, 'run'], env=dict(os.environ, MDD_DIAG_PAD=str(pad)))
