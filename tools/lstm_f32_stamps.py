"""Diagnostic: phase shares of the exact-fp32 persistent BiLSTM kernel (MDD_LSTM_DBG=1 makes the kernel write cycle sums)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['MDD_LSTM_DBG'] = '1'
import torch, numpy as np
from ctc_attention_mispronunciation_amd import synth
from ctc_attention_mispronunciation_amd.hip_model import HipModel
PREC = os.environ.get('STAMP_PRECISION', 'f32')   # f32: lstm_layer_f32_kernel; f32x6: lstm_layer_x6_kernel (teams of 16)
for B in [int(v) for v in (sys.argv[1:] or ['512'])]:
    geom = synth.Geometry(**synth.REFERENCE)
    sd = synth.synth_state_dict(geom)
    x, x1, _, _ = synth.synth_batch(geom, B=B, T=500, L=40, ragged=False)
    m = HipModel(geom, sd, precision=PREC)
    x, x1 = torch.from_numpy(x).cuda(), torch.from_numpy(x1).cuda()
    for _ in range(2): m.forward(x, x1)
    torch.cuda.synchronize()
    ph = m.tap('lstm_dbg').view(torch.int64).view(256, 6).cpu().numpy().astype(np.float64)
    nbt = ((B + 15) // 16 + 15) // 16 if PREC == 'f32' else ((B + 7) // 8 + 15) // 16
    names = ['poll + barrier', 'products (MFMA loop)', 'vmcnt(0) + tag check', 'cell update', 'publish + output stores']
    tot = ph[:, :5].sum(1)
    print('B=%d (%d tiles per team): cycles per (tile, step), mean over workgroups / max workgroup' % (B, nbt))
    for i in [1, 2, 3, 4, 0]:
        print('  %-26s %8.0f  (%4.1f%%)   max %8.0f   min %8.0f' % (names[i], ph[:, i].mean() / 250 / nbt, 100 * ph[:, i].mean() / tot.mean(), ph[:, i].max() / 250 / nbt, ph[:, i].min() / 250 / nbt))
    if PREC != 'f32':   # lstm_layer_x6_kernel: the reporting wave is (blockIdx / 8) % 4 -- wave (0,0) owner | (0,1) owner of the halved pair | (1,0) owner | (1,1) tag checker
        for wv in range(4):
            sel = ph[((np.arange(256) >> 3) & 3) == wv]
            print('    wave %d: ' % wv + '  '.join('%s %6.0f' % (names[i].split()[0], sel[:, i].mean() / 250 / nbt) for i in [1, 2, 3, 4, 0]))
    print('  total %8.0f cycles per (tile, step) = %.0f per step; extra sweep passes per (tile, step): mean %.3f max %.3f'
          % (tot.mean() / 250 / nbt, tot.mean() / 250, ph[:, 5].mean() / 249 / nbt, ph[:, 5].max() / 249 / nbt))
