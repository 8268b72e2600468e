"""Diagnostic: phase shares of the persistent BiLSTM kernel (MDD_LSTM_DBG=1 makes the kernel write cycle sums)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['MDD_LSTM_DBG'] = '1'
import torch, numpy as np
from ctc_attention_mispronunciation_amd import synth
from ctc_attention_mispronunciation_amd.hip_model import HipModel
for B in [int(v) for v in (sys.argv[1:] or ['64', '256'])]:
    geom = synth.Geometry(**synth.REFERENCE)
    sd = synth.synth_state_dict(geom)
    x, x1, _, _ = synth.synth_batch(geom, B=B, T=500, L=40, ragged=False)
    m = HipModel(geom, sd, precision='bf16x3')
    x, x1 = torch.from_numpy(x).cuda(), torch.from_numpy(x1).cuda()
    for _ in range(2): m.forward(x, x1)
    torch.cuda.synchronize()
    ph = m.tap('lstm_dbg').view(torch.int64).view(256, 6).cpu().numpy().astype(np.float64)
    names = ['wait team', '(of which: the counted vmcnt wait)', 'MFMA', 'cell+publish', 'output stores', '-']
    tot = ph[:, [0, 2, 3, 4]].sum(1)
    print('B=%d  cycles per step (mean over WGs / max WG):' % B)
    for i in range(5):
        print('  %-34s %8.0f  (%4.1f%%)   max %8.0f' % (names[i], ph[:, i].mean() / 250, 100 * ph[:, i].mean() / tot.mean(), ph[:, i].max() / 250))
    print('  total %8.0f cycles/step; sweep passes per step: mean %.2f max %.2f' % (tot.mean() / 250, ph[:, 5].mean() / 249, ph[:, 5].max() / 249))
