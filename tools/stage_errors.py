"""Diagnostic: where does the distance to a float64 evaluation come from?  Per stage tap (conv1, rnn0..3, text, key, log-probs):
mean |x - x64| / rms(x64) for ATen fp32 on the CPU and for each arithmetic mode of the library."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import ref_port
from ctc_attention_mispronunciation_amd import synth
from ctc_attention_mispronunciation_amd.hip_model import HipModel
geom = synth.Geometry(**synth.REFERENCE)
sd = synth.synth_state_dict(geom, seed=1234)
B, T, L = int(os.environ.get("B", "16")), 500, 40
x, x1, frac, _ = synth.synth_batch(geom, B=B, T=T, L=L, seed=1234, ragged=True)
torch.set_num_threads(16)
t64, t32 = {}, {}
lp64 = ref_port.forward(sd, x, x1, dtype=torch.float64, taps=t64).numpy(); t64["logp"] = lp64
lp32 = ref_port.forward(sd, x, x1, taps=t32).numpy(); t32["logp"] = lp32
names = ["conv1", "rnn0", "rnn1", "rnn2", "rnn3", "text", "key", "logp"]
rows = {"aten_f32": t32}
for prec in (sys.argv[1:] or ["f32", "f32x6", "bf16x3"]):
    m = HipModel(geom, sd, precision=prec, taps=True)
    lp = m.forward(torch.from_numpy(x).cuda(), torch.from_numpy(x1).cuda(), sync_errors=True).cpu().numpy()
    d = {"logp": lp}
    for n in names[:-1]:
        d[n] = m.tap(n).cpu().numpy()
    rows["hip_" + prec] = d
print("%-12s" % "" + "".join("%11s" % n for n in names))
for k, d in rows.items():
    out = []
    for n in names:
        a, r = np.asarray(d[n], dtype=np.float64).reshape(-1), t64[n].astype(np.float64).reshape(-1)
        out.append(np.abs(a - r).mean() / np.sqrt((r ** 2).mean()))
    print("%-12s" % k + "".join("%11.2e" % v for v in out))
