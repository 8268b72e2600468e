"""Time the projection GEMM's three forms on the bench's shapes (single-barrier, 8-phase with LDS-DMA issued in the load phases,
8-phase with LDS-DMA issued in the MFMA phases) and screen them against each other:  gpurun -- python tools/gemm_ph8_time.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
from ctc_attention_mispronunciation_amd import _lib  # noqa: E402

L = _lib.lib()
torch.zeros(1).cuda()
for M, N, K in ((128000, 3072, 768), (128000, 3072, 1952), (16000, 3072, 768), (20480, 3072, 512)):
    bad, ms = C.c_uint(0), (C.c_float * 16)()
    rc = L.mdd_diag_gemm_ph8(M, N, K, 6, 1, C.byref(bad), ms)
    fl = 2.0 * M * N * K
    print("M=%d N=%d K=%d rc=%d mismatches=%d  single-barrier %.3f ms (%.0f TF)  ph8/L %.3f ms (%.0f TF)  ph8/M %.3f ms (%.0f TF)" %
          (M, N, K, rc, bad.value, ms[0], fl / ms[0] / 1e9, ms[1], fl / ms[1] / 1e9, ms[2], fl / ms[2] / 1e9))
    if os.environ.get("MDD_GEMM_T128"):
        print("   128x128 kernel, two workgroups per CU: %.3f ms (%.0f TF)" % (ms[14], fl / ms[14] / 1e9))
    if os.environ.get("MDD_GEMM_AFIRST"):
        print("   ph8/L with the A fragment as first MFMA operand (scalar C stores): %.3f ms (%.0f TF)" % (ms[11], fl / ms[11] / 1e9))
    if os.environ.get("MDD_GEMM_NOSTORE"):
        print("   ph8/L (stamped build) WITHOUT its C stores: %.3f ms" % ms[12])
    if os.environ.get("MDD_GEMM_STAMP"):
        for name, o in (("DMA in L", 3), ("DMA in M", 7)):
            print("   %s: cycles per K-tile and wave: load bodies %.0f, waiting at their barriers %.0f, MFMA bodies %.0f (floor 4 x 24 x 16 = 1536), waiting at theirs %.0f"
                  % (name, ms[o], ms[o + 1], ms[o + 2], ms[o + 3]))
