"""Diagnostic: does any kernel of the forward read memory it was not given?  Every call's inputs are placed in freshly poisoned allocator blocks
(the caching allocator's free blocks are filled with 1e30 / huge ints first), so that a read past an input's end changes the result."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ctc_attention_mispronunciation_amd import synth
from ctc_attention_mispronunciation_amd.hip_model import HipModel
prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
H = int(sys.argv[2]) if len(sys.argv) > 2 else 384
geom = synth.Geometry(**dict(synth.REFERENCE, hidden=H))
sd = synth.synth_state_dict(geom, seed=1234)
shapes = [(5, 120, 9), (3, 64, 4), (7, 100, 12), (2, 120, 12), (4, 30, 1)]
rs = np.random.Generator(np.random.PCG64(7))
Bt, Tm, Lm = sum(s[0] for s in shapes), max(s[1] for s in shapes), max(s[2] for s in shapes)
X = np.zeros((Bt, Tm, geom.feat), dtype=np.float32); X1 = np.zeros((Bt, Lm), dtype=np.int64)
frames, canon = np.zeros(Bt, dtype=np.int32), np.zeros(Bt, dtype=np.int32)
r = 0
for (b, T, L) in shapes:
    X[r:r + b, :T] = rs.standard_normal((b, T, geom.feat)).astype(np.float32); X1[r:r + b, :L] = rs.integers(2, 44, size=(b, L))
    frames[r:r + b] = T // 2; canon[r:r + b] = L; r += b
m = HipModel(geom, sd, precision=prec)
def poison(val):
    blocks = [torch.full((n,), val, device="cuda") for n in (64, 600, 5000, 70000, 700000, 3000000)]
    blocks += [torch.full((n,), 2 ** 40 + 12345, dtype=torch.int64, device="cuda") for n in (16, 300, 4000)]
    torch.cuda.synchronize()
    del blocks
def run(val):
    if val is not None: poison(val)
    lp = m.forward_fused(torch.from_numpy(X).cuda(), torch.from_numpy(X1).cuda(), torch.from_numpy(frames).cuda(), torch.from_numpy(canon).cuda(), sync_errors=True).cpu().numpy()
    return lp
ref = run(None)
nbad = 0
for it, val in enumerate([0.0, 1e30, -1e30, float("nan"), 1e30, 3.0, 1e30, 1e30]):
    cur = run(val)
    bad = []
    for b in range(Bt):
        d = np.abs(cur[:frames[b], b] - ref[:frames[b], b])
        if not (d == 0).all(): bad.append((b, float(np.nanmax(d)), int((d != 0).sum())))
    nbad += bool(bad)
    print("poison %-6s run %d: %s" % (val, it, "identical" if not bad else "DIFFERS rows (b, max, n): %s" % bad[:6]))
print("%s H=%d: %d of 8 poisoned runs differ" % (prec, H, nbad))
