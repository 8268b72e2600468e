"""Ad-hoc stage timing used while developing (not the contract bench: see bench.py)."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, numpy as np
from ctc_attention_mispronunciation_amd import synth
from ctc_attention_mispronunciation_amd.hip_model import HipModel
from ctc_attention_mispronunciation_amd.utils.ctcDecoder import GreedyDecoder, BeamDecoder
from ctc_attention_mispronunciation_amd.utils.data_loader import stack_features
B=int(os.environ.get('B','64')); steps=int(os.environ.get('STEPS','5'))
geom = synth.Geometry(**synth.REFERENCE)
sd = synth.synth_state_dict(geom, seed=1234)
raw = torch.from_numpy(synth.synth_raw_features(B)).cuda()
_, x1, _, _ = synth.synth_batch(geom, B=B, T=500, L=40, ragged=False)
x1 = torch.from_numpy(x1).cuda()
m = HipModel(geom, sd)
i2c = synth.phone_table_41()
bd = BeamDecoder(i2c, beam_width=10, blank_index=0, space_idx=-1, lm_path=os.path.join(ROOT, 'tests/golden/lm_synth45.arpa'), lm_alpha=0.0)
gd = GreedyDecoder(i2c, space_idx=-1, blank_index=0)
lens = torch.full((B,), 250, dtype=torch.int32, device='cuda')
def step(decode=True):
    x = stack_features(raw)
    lp = m.forward(x, x1)
    if decode:
        return bd.decode_ids(lp, lens), gd.decode_ids(lp, lens)
    return lp
for _ in range(2): step()
torch.cuda.synchronize()
for name, fn in (('forward', lambda: step(False)), ('forward+beam+greedy', step)):
    t0=time.time()
    for _ in range(steps): fn()
    torch.cuda.synchronize()
    dt=(time.time()-t0)/steps
    print('%s: %.3f ms/step  %.0f frames/s' % (name, dt*1e3, B*250/dt))
lp = step(False)
for name, fn in (('beam', lambda: bd.decode_ids(lp, lens)), ('greedy', lambda: gd.decode_ids(lp, lens))):
    torch.cuda.synchronize(); t0=time.time()
    for _ in range(steps): fn()
    torch.cuda.synchronize(); print('%s: %.3f ms' % (name, (time.time()-t0)/steps*1e3))
