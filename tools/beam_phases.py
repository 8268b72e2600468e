"""Diagnostic: time mdd_beam with phases skipped (MDD_BEAM_SKIP bit mask; results are wrong, only time matters)."""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    import torch, numpy as np
    from ctc_attention_mispronunciation_amd import synth
    from ctc_attention_mispronunciation_amd.utils.ctcDecoder import BeamDecoder
    T, B, C = 250, 64, 45
    z = np.random.Generator(np.random.PCG64(1)).standard_normal((T, B, C)).astype(np.float32)
    lp = torch.log_softmax(torch.from_numpy(z), -1).cuda()
    bd = BeamDecoder(synth.phone_table_41(), beam_width=10, blank_index=0, space_idx=-1, lm_path=os.path.join(ROOT, 'tests/golden/lm_synth45.arpa'), lm_alpha=0.0)
    lens = torch.full((B,), T, dtype=torch.int32, device='cuda')
    for _ in range(2): bd.decode_ids(lp, lens)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(5): bd.decode_ids(lp, lens)
    torch.cuda.synchronize(); print('skip=%s: %.3f ms' % (os.environ.get('MDD_BEAM_SKIP', '0'), (time.time() - t0) / 5 * 1e3))
else:
    for mask in (0, 1, 2, 4, 8, 16, 32, 63):
        subprocess.call([sys.executable, __file__, 'run'], env=dict(os.environ, MDD_BEAM_SKIP=str(mask)))
