#!/usr/bin/env python3
"""Scripted CPU timing of the REFERENCE ITSELF -- TEST INFRASTRUCTURE, build container only.

Imports the reference's own Python from /root/reference exactly as oracle/gen_golden.py does (CTC_Model, GreedyDecoder,
BeamDecoder), feeds it the bench workload (synthetic 10 s x 81-dim utterances, seeded state_dict, L=40) and times, with
1 warm-up and REPS timed repetitions (median reported):
  forward  B=64 H=384 / B=32 H=256,  greedy decode,  beam(10) decode on the model's own (flat) posteriors and on the
  'peaky' set of SURVEY.md 8(d),  and one training step (forward + nn.CTCLoss(sum)/B + backward + Adam) at B=4.
Writes oracle/reference_cpu_timing.json; BASELINE.md quotes it.  Usage: python oracle/time_reference.py [REPS]
"""
import json
import os
import statistics
import sys
import time
import types

import numpy as np

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AA = "/root/reference/egs/attention_aug"
sys.path.insert(0, os.path.join(ROOT, "ctc-attention-mispronunciation_amd"))
import synth  # noqa: E402
sys.path.pop(0)
if not os.path.isdir(AA):
    sys.exit("reference tree not present: this measurement only exists in the build container")
sys.modules.setdefault("editdistance", types.ModuleType("editdistance"))
sys.path.insert(0, AA)
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
from models.model_ctc import CTC_Model  # noqa: E402  (reference)
from utils.ctcDecoder import GreedyDecoder, BeamDecoder  # noqa: E402  (reference)

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 5
NCORES = os.cpu_count() or 1
torch.set_num_threads(NCORES)


def timed(fn, reps=REPS, warm=1):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts), ts


def model_for(geom, seed):
    sd = synth.synth_state_dict(geom, seed=seed)
    m = CTC_Model(add_cnn=True, cnn_param=geom.cnn_param(nn), rnn_param=geom.rnn_param(nn), num_class=geom.num_class, drop_out=0.2)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    return m


def main():
    out = {"host": {"cores": NCORES, "torch": torch.__version__, "threads": torch.get_num_threads()}, "reps": REPS, "rows": []}
    i2c = synth.phone_table_41()
    arpa = os.path.join(ROOT, "tests", "golden", "lm_synth45.arpa")
    greedy = GreedyDecoder(i2c, space_idx=-1, blank_index=0)
    beam = BeamDecoder(i2c, beam_width=10, blank_index=0, space_idx=-1, lm_path=arpa, lm_alpha=0.0)
    Tp = 250
    for tag, H, B in (("joint64_h384", 384, 64), ("greedy32_h256", 256, 32)):
        geom = synth.Geometry(feat=243, hidden=H, layers=4, num_class=45)
        m = model_for(geom, 1234).eval()
        raw = synth.synth_raw_features(B, 1000, 81, seed=1234)
        import utils.tools as ref_tools  # reference make_context / skip_feat (AA/utils/data_loader.py:138)
        x = np.stack([ref_tools.skip_feat(ref_tools.make_context(raw[b], 0, 2), 2) for b in range(B)]).astype(np.float32)
        _, x1, _, _ = synth.synth_batch(geom, B=B, T=500, L=40, seed=1234, ragged=False)
        xt, x1t = torch.from_numpy(x), torch.from_numpy(x1)
        with torch.no_grad():
            t_fwd, _ = timed(lambda: m(xt, x1t))
            logp = m(xt, x1t)
        lens = [Tp] * B
        t_gr, _ = timed(lambda: greedy.decode(logp, lens))
        row = dict(config=tag, B=B, H=H, forward_s=t_fwd, greedy_s=t_gr, frames=B * Tp,
                   greedy_path_frames_per_s=B * Tp / (t_fwd + t_gr))
        if tag == "joint64_h384":
            t_bf, _ = timed(lambda: beam.decode(logp, lens), reps=max(3, min(REPS, 3)), warm=0)     # ~1 min per repetition
            peaky = torch.from_numpy(np.stack([synth.peaky_logp(Tp, 45, 35, seed=100 + b) for b in range(B)], axis=1))
            t_bp, _ = timed(lambda: beam.decode(peaky, lens))
            row.update(beam_flat_s=t_bf, beam_peaky_s=t_bp, beam_path_flat_frames_per_s=B * Tp / (t_fwd + t_bf),
                       beam_path_peaky_frames_per_s=B * Tp / (t_fwd + t_bp))
        out["rows"].append(row)
        print(json.dumps(row), flush=True)
    # one training step at B=4 (AA/steps/train_ctc.py:63-74,186-187)
    geom = synth.Geometry(feat=243, hidden=384, layers=4, num_class=45)
    m = model_for(geom, 1234).train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=5e-4)
    loss_fn = nn.CTCLoss(reduction="sum")
    B = 4
    x, x1, frac, tlen = synth.synth_batch(geom, B=B, T=500, L=40, seed=7, ragged=False)
    rs = np.random.Generator(np.random.PCG64(1))
    tg = torch.from_numpy(rs.integers(2, 44, size=(B, 40)))
    xt, x1t = torch.from_numpy(x), torch.from_numpy(x1)

    def step():
        out_ = m(xt, x1t)
        il = (torch.from_numpy(frac) * out_.size(0)).long()
        loss = loss_fn(out_, tg, il, torch.full((B,), 40, dtype=torch.long)) / B
        opt.zero_grad()
        loss.backward()
        opt.step()
    t_tr, _ = timed(step, reps=3)
    out["rows"].append(dict(config="train_step_b4_h384", B=B, step_s=t_tr, frames=B * Tp, frames_per_s=B * Tp / t_tr))
    print(json.dumps(out["rows"][-1]), flush=True)
    json.dump(out, open(os.path.join(ROOT, "oracle", "reference_cpu_timing.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
