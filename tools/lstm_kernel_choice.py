"""Diagnostic: time of one BiLSTM layer (stage replay) with the f32x6 recurrence (lstm_layer_x6_kernel) beside the exact-fp32 layer kernel,
over batch sizes and both hidden sizes -- the table behind mdd_model::lx6()'s choice.  Each configuration runs in a child process
(MDD_LSTM_X6 is read when the handle is created)."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch, numpy as np
    from ctc_attention_mispronunciation_amd import synth
    from ctc_attention_mispronunciation_amd.hip_model import HipModel
    H = int(sys.argv[2]); res = {}
    geom = synth.Geometry(**dict(synth.REFERENCE, hidden=H))
    sd = synth.synth_state_dict(geom)
    m = HipModel(geom, sd, precision="f32x6")
    for B in [int(v) for v in sys.argv[3:]]:
        x, x1, _, _ = synth.synth_batch(geom, B=B, T=500, L=40, ragged=False)
        x, x1 = torch.from_numpy(x).cuda(), torch.from_numpy(x1).cuda()
        best = None
        for _ in range(3):
            st = m.profile(x, x1)
            v = float(np.mean([t[1] for t in st if t[0].startswith("lstm") and t[0] != "lstm_text"]))
            best = v if best is None else min(best, v)
        res[B] = best
    print("RESULT " + json.dumps(res))
    sys.exit(0)
Bs = [32, 64, 128, 192, 256, 320, 384, 512, 640, 768, 1024]
for H in (384, 256):
    rows = {}
    for x6 in ("1", "0"):
        env = dict(os.environ, MDD_LSTM_X6=x6)
        out = subprocess.run([sys.executable, __file__, "child", str(H)] + [str(b) for b in Bs], env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print(out.stdout[-2000:], out.stderr[-2000:]); sys.exit(1)
        rows[x6] = json.loads(line[0][7:])
    print("H=%d  ms per BiLSTM layer (T'=250), mean of the four layers, best of 3" % H)
    print("   B      x6 kernel   fp32 kernel   x6 / fp32")
    for b in Bs:
        a, c = rows["1"][str(b)], rows["0"][str(b)]
        print("%5d   %10.3f   %10.3f   %8.2f" % (b, a, c, a / c))
