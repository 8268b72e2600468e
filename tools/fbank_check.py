"""Diagnostic: max |HIP fbank - numpy restatement| and kernel time for a 10 s utterance."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import oracle
from ctc_attention_mispronunciation_amd.utils import fbank as fb
wav, sr = fb.read_wav(os.path.join(ROOT, "tests/golden/vocabulary_single_1.wav"))
d = np.abs(fb.compute_fbank_feats(wav).cpu().numpy() - oracle.fbank(wav))
print("word: max |diff| %.3g  (energy col %.3g)" % (d.max(), d[:, 0].max()))
x = (np.random.Generator(np.random.PCG64(1)).standard_normal(160000) * 2000).astype(np.float32)
xd = torch.from_numpy(x).cuda()
d = np.abs(fb.compute_fbank_feats(xd).cpu().numpy() - oracle.fbank(x))
print("noise 10 s: max |diff| %.3g" % d.max())
for _ in range(3): fb.compute_fbank_feats(xd)
torch.cuda.synchronize(); t = time.time()
for _ in range(100): fb.compute_fbank_feats(xd)
torch.cuda.synchronize(); print("10 s utterance: %.1f us per call (998 frames)" % ((time.time() - t) / 100 * 1e6))
