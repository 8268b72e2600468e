#!/usr/bin/env python3
"""Contract bench: phoneme-frames/s of the joint CTC-attention decode hot path on MI355X.

One "step" = one pass of the hot path over one reference-sized batch of synthetic utterances that is
already resident in HBM: stack/skip (A1) -> CTC_Model.forward (A2-A7) -> CTC prefix beam search, width 10
(A9) -> decoded ids back on the host -> edit-distance alignment against the canonical phonemes (A10).
Workload (BASELINE.json configs[2], the configuration the metric is quoted on): B=64 utterances of 10 s
(1000 x 81 log-mel+energy frames -> 250 posterior frames each), 4xBiLSTM-384, 41-phone set (45 classes),
canonical length 40, beam 10, lm_alpha 0.  `--workload greedy32` runs configs[1] (B=32, greedy decode),
`--workload ctc256` the CTC alpha/beta lattice (loss + gradient) of configs[4]'s global batch.

The headline `value` carries `--fuse` (default 8) reference batches per launch sequence (every utterance full length).  The
reference masks nothing, so an utterance's result depends on its batch's padded lengths; mdd_forward_fused carries batches of
DIFFERENT padded lengths in one launch sequence with per-row (T_g, L_g), bit-identical to running each batch alone.  What real
(ragged) data gets is reported in the same JSON line under `variants` (same run, rank 0, N=1): `ragged` (len ~ U[0.5,1] x 10 s,
every batch padded to its own maximum, 8 such batches per fused pass, frames counted UNPADDED), `ragged_fuse1` / `fuse1` (a lone
B=64 batch per pass: the latency-bound case), `bf16x3_mode` (the flagged split-bf16 arithmetic, narrower than the reference's:
MDD_PRECISION=bf16x3), `greedy32_h256` (configs[1]) and `train32_f32` / `train32_bf16x3` (configs[4]'s per-GPU shard: a full training
step in exact fp32 and in the flagged split-bf16 variant; `--workload train32 [--train-precision bf16x3]` prints that line on its
own, with the stage split).  The headline itself is REFERENCE-WIDTH arithmetic (the library's default mode "f32x6"): every operand with
its full 24-bit significand and fp32 accumulation, as the reference's ATen fp32 ops (AA/models/model_ctc.py:27-29,59-66,149-158) --
the large contractions as three bf16 planes per operand / six products on the bf16 matrix cores, the rest as exact fp32 MFMAs -- and the
line carries, measured in the same run against a float64 evaluation, how far its log-probs and ATen-fp32's are from it (`accuracy`);
`f32_mfma_mode` is the same path with every product an exact fp32 MFMA.

N>1 (launched by torch.distributed.run, one rank per GPU): utterance batches shard across ranks (weak
scaling: every rank decodes its own 64-utterance batch) and the posteriors of all shards are all-gathered
over RCCL/xGMI each step on a stream of their own, behind the next pass's forward (`--gather posteriors`, what BASELINE.json's north_star
names: 2.9 MB per rank and batch, 23 MB per rank and 8-batch pass), or -- the default, since nothing downstream of the decode
consumes the other ranks' posteriors -- the decoded ids (`--gather ids`: [B, T'] int32 + lengths, 0.5 MB per rank and pass).

Prints ONE JSON line on rank 0 (see the driver contract in the task statement) including `roofline`
for the dominant kernel and a `cpu_baseline` measured in the same run on the host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_F32_MATRIX_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_* dense peak
PEAK_BF16_MATRIX_TFLOPS = 2500.0 # dense bf16 MFMA peak (not the 2:1-sparsity figure)
PEAK_HBM_GBS = 8000.0
PMC_FILE = os.path.join(ROOT, "profiles", "round3_pmc_traffic.json")
T_RAW, D_RAW, L_CANON, BEAM_W, N_CLASS = 1000, 81, 40, 10, 45


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--workload", default="joint64", choices=["joint64", "greedy32", "ctc256", "train32"])
    ap.add_argument("--train-precision", default="f32", choices=["f32", "bf16x3"],
                    help="train32: exact fp32 (as the reference trains) or the large contractions as split-bf16 x3 (BASELINE configs[4] names bf16)")
    ap.add_argument("--hidden", type=int, default=384)
    ap.add_argument("--fuse", type=int, default=8, help="reference-sized batches carried by one launch sequence")
    ap.add_argument("--ragged", action="store_true", help="headline on the ragged set (per-batch padding, unpadded frames counted)")
    ap.add_argument("--precision", default=None, choices=[None, "f32", "f32x6", "bf16x3"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the fuse1 / ragged / f32_mode / greedy32_h256 lines")
    ap.add_argument("--gather", default="ids", choices=["ids", "posteriors", "none"],
                    help="what the ranks exchange per pass at N>1 (own stream, overlapped with the next forward): decoded ids (default) or the posteriors")
    ap.add_argument("--no-gather", action="store_true", help="same as --gather none")
    ap.add_argument("--decoder", default=None, choices=[None, "beam", "greedy", "none"], help="diagnostic override of the decode stage")
    ap.add_argument("--no-roofline", action="store_true", help="skip the stage-replay pass (for clean traces)")
    return ap.parse_args()


class Ctx(object):
    """Process-wide pieces: rank layout, torch.distributed handle."""

    def __init__(self, args):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != args.gpus and self.world > 1:
            raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, self.world))
        ndev = torch.cuda.device_count()
        if self.local >= ndev:        # rehearsal of the N>1 path on a box with fewer GPUs than ranks (MDD_DIST_BACKEND=gloo)
            self.local = self.local % ndev
        torch.cuda.set_device(self.local)
        self.dist = None
        if self.world > 1 or os.environ.get("MDD_FORCE_DIST"):    # MDD_FORCE_DIST=1: a one-rank process group, to exercise the RCCL calls on a one-GPU box
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            backend = os.environ.get("MDD_DIST_BACKEND", "nccl")   # "nccl" is RCCL on ROCm
            # The pool exports NCCL_DEBUG=VERSION and RCCL prints its banner on stdout when the communicator is created; stdout must carry
            # ONE JSON line, so file descriptor 1 points at stderr until the first collective has run (C stdio flushed before it is restored).
            import ctypes
            sys.stdout.flush()
            saved_fd = os.dup(1)
            os.dup2(2, 1)
            try:
                if backend == "nccl":
                    dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=torch.device("cuda", self.local))
                else:
                    dist.init_process_group(backend, rank=self.rank, world_size=self.world)
                dist.barrier()
                torch.cuda.synchronize()
            finally:
                ctypes.CDLL(None).fflush(None)
                os.dup2(saved_fd, 1)
                os.close(saved_fd)
            self.dist = dist

    def barrier(self):
        torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        torch.cuda.synchronize()


class DecodeJob(object):
    """One configuration of the hot path: model, resident inputs, two-stream pipeline (forward of pass i+1 on s_fwd
    beside the beam search + D2H of pass i on s_dec; the host aligns pass i meanwhile)."""

    def __init__(self, ctx, args, hidden, joint, fuse, ragged, precision, decoder_kind=None, gather="ids"):
        from ctc_attention_mispronunciation_amd import synth, dist as mdist
        from ctc_attention_mispronunciation_amd.hip_model import HipModel
        from ctc_attention_mispronunciation_amd.utils.ctcDecoder import GreedyDecoder, BeamDecoder
        from ctc_attention_mispronunciation_amd.utils.data_loader import stack_features
        self.ctx, self.joint, self.ragged, self.fuse = ctx, joint, ragged, fuse
        self.mdist = mdist
        self.B0 = 64 if joint else 32
        self.geom = synth.Geometry(feat=3 * D_RAW, hidden=hidden, layers=4, num_class=N_CLASS)
        self.sd = synth.synth_state_dict(self.geom, seed=1234)
        self.model = HipModel(self.geom, self.sd, device=ctx.local, precision=precision)
        self.i2c = synth.phone_table_41()
        arpa = os.path.join(ROOT, "tests", "golden", "lm_synth45.arpa")
        kind = decoder_kind or ("beam" if joint else "greedy")
        self.decoder_kind = kind
        self.decoder = (BeamDecoder(self.i2c, beam_width=BEAM_W, blank_index=0, space_idx=-1, lm_path=arpa, lm_alpha=0.0)
                        if kind == "beam" else GreedyDecoder(self.i2c, space_idx=-1, blank_index=0))
        self.gather = gather if (ctx.dist is not None and gather != "none") else None
        if self.gather == "ids" and kind == "none":
            self.gather = None
        self.s_fwd, self.s_dec, self.s_comm = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
        self.ev_tf0 = [torch.cuda.Event(enable_timing=True) for _ in range(4)]   # device timestamps around each forward (by pass number & 3)
        self.ev_tf1 = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        self.ev_fwd = [torch.cuda.Event() for _ in range(2)]
        self.ev_dec = [torch.cuda.Event() for _ in range(2)]
        self.ev_free = [torch.cuda.Event() for _ in range(2)]
        self.ev_c0 = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        self.ev_comm = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        self.comm_pending = [False, False]
        self.gather_ms, self.fwd_gap_ms, self.last_slot = [], [], None
        self.count = 0
        self.aligned = []
        self.frames_done = 0
        seed = 1234 + ctx.rank                                   # every rank decodes a different shard, same shapes
        if not ragged:
            # fuse G same-shape batches per pass; a step stays one B0-utterance batch
            nb = fuse
            self.raw = torch.from_numpy(synth.synth_raw_features(self.B0 * nb, T_RAW, D_RAW, seed=seed)).cuda()
            _, x1_np, _, _ = synth.synth_batch(self.geom, B=self.B0 * nb, T=T_RAW // 2, L=L_CANON, seed=seed, ragged=False)
            self.x1_np = x1_np
            self.x1 = torch.from_numpy(x1_np).cuda()
            self.sets = {}
        else:
            # 8 different reference batches: len ~ U[0.5,1] x 10 s, each utterance stacked/skipped on its own (the last
            # frame replicated at ITS edge, data_loader.py:138-142), each batch zero-padded to its own maximum (:159,173)
            # and run alone (a fused pass needs identical T and Lmax).  Canonical lengths ~ U[0.5,1] x 40, padded per batch.
            rs = np.random.Generator(np.random.PCG64(seed))
            batches = []
            for k in range(8 if fuse == 1 else 2 * fuse):
                lens_raw = rs.integers(T_RAW // 2, T_RAW + 1, size=self.B0)
                raw = synth.synth_raw_features(self.B0, T_RAW, D_RAW, seed=seed + 100 * (k + 1))
                t_st = [int((n + 1) // 2 + ((n + 1) // 2) % 2) for n in lens_raw]        # mdd_stack_len(n, 2, 2)
                T = max(t_st)
                x = torch.zeros((self.B0, T, 3 * D_RAW), device="cuda")
                for b in range(self.B0):
                    x[b, :t_st[b]] = stack_features(torch.from_numpy(raw[b, :lens_raw[b]]))
                llen = rs.integers(L_CANON // 2, L_CANON + 1, size=self.B0)
                Lm = int(llen.max())
                x1 = np.zeros((self.B0, Lm), dtype=np.int64)
                for b in range(self.B0):
                    x1[b, :llen[b]] = rs.integers(2, 44, size=llen[b])
                frac = torch.tensor([t / T for t in t_st]).float()                          # data_loader.py:177
                lens = (frac * (T // 2)).long().to(torch.int32)                             # infer.py:296-297
                batches.append(dict(x=x, x1_np=x1, lens=lens, T=T, L=Lm, canon_len=llen.astype(np.int32)))
            # a pass = `fuse` of those batches in one launch sequence (mdd_forward_fused): common T / L, per-row (T_g/2, L_g)
            self.rag = []
            for k0 in range(0, len(batches), fuse):
                grp = batches[k0:k0 + fuse]
                Tm, Lm, nb = max(g["T"] for g in grp), max(g["L"] for g in grp), self.B0 * len(grp)
                X = torch.zeros((nb, Tm, 3 * D_RAW), device="cuda")
                X1 = np.zeros((nb, Lm), dtype=np.int64)
                frames, canon = np.zeros(nb, dtype=np.int32), np.zeros(nb, dtype=np.int32)
                for i, g in enumerate(grp):
                    r = slice(i * self.B0, (i + 1) * self.B0)
                    X[r, :g["T"]] = g["x"]; X1[r, :g["L"]] = g["x1_np"]; frames[r] = g["T"] // 2; canon[r] = g["L"]
                self.rag.append(dict(x=X, x1=torch.from_numpy(X1).cuda(), x1_np=X1, lens=torch.cat([g["lens"] for g in grp]).cuda(), T=Tm,
                                     canon_len=np.concatenate([g["canon_len"] for g in grp]), frames=int(sum(int(g["lens"].sum()) for g in grp)),
                                     padded=int(sum(g["T"] // 2 * self.B0 for g in grp)), nb=nb, batches=len(grp),
                                     bframes=torch.from_numpy(frames).cuda(), bcanon=torch.from_numpy(canon).cuda()))
            del batches
        torch.cuda.synchronize()

    class Bufs(object):
        pass

    def _bufs(self, key):
        if key in self.sets:
            return self.sets[key]
        bf = DecodeJob.Bufs()
        g = key
        b = g * self.B0
        Tp = T_RAW // 4
        bf.b, bf.Tp = b, Tp
        bf.raw, bf.x1 = self.raw[:b], self.x1[:b]
        bf.lens = torch.full((b,), Tp, dtype=torch.int32, device="cuda")
        bf.canon = np.ascontiguousarray(self.x1_np[:b], dtype=np.int32)
        bf.canon_len = np.full((b,), bf.canon.shape[1], dtype=np.int32)
        bf.frames = b * Tp
        self._alloc_out(bf)
        self.sets[key] = bf
        return bf

    def _alloc_out(self, bf):
        w = self.ctx.world
        bf.logp = [torch.empty((bf.Tp, bf.b, N_CLASS), device="cuda") for _ in range(2)]
        bf.gathered = [torch.empty((w, bf.Tp, bf.b, N_CLASS), device="cuda") for _ in range(2)] if self.gather == "posteriors" else None
        bf.g_ids = [torch.empty((w, bf.b, bf.Tp), dtype=torch.int32, device="cuda") for _ in range(2)] if self.gather == "ids" else None
        bf.g_n = [torch.empty((w, bf.b), dtype=torch.int32, device="cuda") for _ in range(2)] if self.gather == "ids" else None
        bf.dev_out = [None, None]
        bf.h_ids = [torch.empty((bf.b, bf.Tp), dtype=torch.int32).pin_memory() for _ in range(2)]
        bf.h_n = [torch.empty((bf.b,), dtype=torch.int32).pin_memory() for _ in range(2)]

    def _rag_bufs(self, k):
        r = self.rag[k]
        if "bf" not in r:
            bf = DecodeJob.Bufs()
            bf.b, bf.Tp = r["nb"], r["T"] // 2
            bf.x, bf.x1, bf.lens, bf.bframes, bf.bcanon = r["x"], r["x1"], r["lens"], r["bframes"], r["bcanon"]
            bf.canon = np.ascontiguousarray(r["x1_np"], dtype=np.int32)
            bf.canon_len = r["canon_len"]
            bf.frames = r["frames"]
            self._alloc_out(bf)
            r["bf"] = bf
        return r["bf"]

    def enqueue(self, item):
        """item: number of fused batches (uniform set) or the index of a ragged batch."""
        bf = self._rag_bufs(item) if self.ragged else self._bufs(item)
        k, q = self.count & 1, self.count & 3
        self.count += 1
        with torch.cuda.stream(self.s_fwd):
            self.s_fwd.wait_event(self.ev_free[k])          # slot k's buffers no longer read by the decoder two passes ago
            if self.comm_pending[k]:
                self.s_fwd.wait_event(self.ev_comm[k])      # ... nor by that pass's result exchange
            self.ev_tf0[q].record(self.s_fwd)
            if self.ragged:
                if self.fuse > 1:
                    self.model.forward_fused(bf.x, bf.x1, bf.bframes, bf.bcanon, out=bf.logp[k])
                else:
                    self.model.forward(bf.x, bf.x1, out=bf.logp[k])
            else:
                self.model.forward_raw(bf.raw, bf.x1, out=bf.logp[k])    # stack/skip folded into the front-end's tile load
            self.ev_fwd[k].record(self.s_fwd)
            self.ev_tf1[q].record(self.s_fwd)
        with torch.cuda.stream(self.s_dec):
            self.s_dec.wait_event(self.ev_fwd[k])
            if self.decoder_kind != "none":
                out = self.decoder.decode_ids(bf.logp[k], bf.lens)
                bf.dev_out[k] = out                          # kept until the slot is reused: the exchange reads them on another stream
                bf.h_ids[k].copy_(out[0], non_blocking=True)
                bf.h_n[k].copy_(out[1], non_blocking=True)
            self.ev_free[k].record(self.s_dec)
            self.ev_dec[k].record(self.s_dec)
        return (bf, k, q)

    def exchange(self, pend):
        """N>1: the result exchange of a pass, on its own stream.  It is issued AFTER the next pass's forward has been enqueued
        (host order) and depends only on its own pass's events (device order), so it runs beside that forward instead of in front of
        it; the forward that reuses the slot two passes later waits for it."""
        if self.gather is None:
            return
        bf, k, _ = pend
        delay = float(os.environ.get("MDD_BENCH_GATHER_DELAY_MS", "0"))   # test aid: a spinning kernel in front of the collective makes
        with torch.cuda.stream(self.s_comm):                               # the exchange long enough to show up in the stream timestamps
            if self.gather == "posteriors":
                self.s_comm.wait_event(self.ev_fwd[k])
                self.ev_c0[k].record(self.s_comm)
                if delay > 0:
                    torch.cuda._sleep(int(delay * 1.5e6))
                self.mdist.gather_posteriors(bf.logp[k], out=bf.gathered[k])
            else:
                self.s_comm.wait_event(self.ev_dec[k])
                self.ev_c0[k].record(self.s_comm)
                self.mdist.gather_decoded(bf.dev_out[k][0], bf.dev_out[k][1], out=(bf.g_ids[k], bf.g_n[k]))
            self.ev_comm[k].record(self.s_comm)
        self.comm_pending[k] = True

    def finish(self, pend):
        from ctc_attention_mispronunciation_amd.utils.ctcDecoder import align_ids_batch
        bf, k, q = pend
        self.ev_dec[k].synchronize()
        if self.comm_pending[k]:
            self.ev_comm[k].synchronize()
            self.gather_ms.append(self.ev_c0[k].elapsed_time(self.ev_comm[k]))
        if self.gather is not None and self.last_slot == ((q - 1) & 3):   # idle time of the forward stream between two consecutive passes
            self.fwd_gap_ms.append(self.ev_tf1[self.last_slot].elapsed_time(self.ev_tf0[q]))
        self.last_slot = q
        tot = 0
        if self.decoder_kind != "none":
            d = align_ids_batch(bf.h_ids[k].numpy(), bf.h_n[k].numpy(), bf.canon, bf.canon_len)[0]   # one native call per pass
            tot = int(d[d >= 0].sum())
        self.aligned.append(tot)
        self.frames_done += bf.frames

    def items(self, nsteps):
        if self.ragged:          # a step is one reference batch; a pass carries `fuse` of them
            return [i % len(self.rag) for i in range((nsteps + self.fuse - 1) // self.fuse)]
        G = max(1, min(self.fuse, nsteps))
        return [G] * (nsteps // G) + ([nsteps % G] if nsteps % G else [])

    def run(self, items):
        pending = []
        for it in items:
            cur = self.enqueue(it)
            if pending:
                self.exchange(pending[-1])                       # the previous pass's results, behind this pass's forward
            pending.append(cur)
            if len(pending) > 1:                                 # keep at most one finished-but-unaligned pass
                self.finish(pending.pop(0))
        if pending:
            self.exchange(pending[-1])
        for p in pending:
            self.finish(p)

    def check_errors(self):
        """mdd_sync: asynchronous errors of the handle (a persistent BiLSTM launch that timed out waiting for its team
        leaves garbage posteriors; the throughput of such a run must not be reported)."""
        from ctc_attention_mispronunciation_amd import _lib
        for st in (self.s_fwd, self.s_dec):
            rc = _lib.lib().mdd_sync(self.model.handle, _lib.C.c_void_p(st.cuda_stream))
            if rc != 0:
                raise SystemExit("bench: the library reported an asynchronous error: %s" % _lib.lib().mdd_last_error().decode())

    def timed(self, steps, warmup):
        """W untimed warm-up steps (every pass shape of the timed region captured beforehand), then exactly `steps` steps
        between barrier + synchronize on both sides; MAX over ranks.  Returns (seconds, frames decoded by this rank)."""
        timed = self.items(steps)
        warm = self.items(max(warmup, 1))
        for g in sorted(set(timed) - set(warm)):
            warm.append(g)
        self.run(warm)
        self.ctx.barrier()
        self.aligned.clear()
        self.frames_done = 0
        self.gather_ms, self.fwd_gap_ms, self.last_slot = [], [], None
        t0 = time.perf_counter()
        self.run(timed)
        self.ctx.barrier()
        dt = time.perf_counter() - t0
        self.check_errors()
        if self.ctx.dist is not None:
            t = torch.tensor([dt], device="cuda", dtype=torch.float64)
            if self.ctx.dist.get_backend() == "gloo":
                t = t.cpu()
            self.ctx.dist.all_reduce(t, op=self.ctx.dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, self.frames_done

    def verify_gather(self):
        """N>1: every rank's slice of the last exchanged buffers equals its own results, and the slices of the other ranks
        differ from it (they decode different shards).  Returns True/False agreed over all ranks (None at N=1)."""
        if self.gather is None:
            return None
        ok = True
        for bf in ([r["bf"] for r in self.rag if "bf" in r] if self.ragged else list(self.sets.values())):
            for k in range(2):
                if self.count < 2 and k >= self.count:
                    continue                                  # slot never used
                if self.gather == "posteriors":
                    own, mine = bf.gathered[k][self.ctx.rank], bf.logp[k]
                    others = [bf.gathered[k][r] for r in range(self.ctx.world) if r != self.ctx.rank]
                else:
                    own, mine = bf.g_ids[k][self.ctx.rank], bf.dev_out[k][0]
                    ok = ok and bool(torch.equal(bf.g_n[k][self.ctx.rank], bf.dev_out[k][1]))
                    others = [bf.g_ids[k][r] for r in range(self.ctx.world) if r != self.ctx.rank]
                ok = ok and bool(torch.equal(own, mine))
                ok = ok and all(not torch.equal(o, own) for o in others)
        t = torch.tensor([1.0 if ok else 0.0], device="cpu" if self.ctx.dist.get_backend() == "gloo" else "cuda")
        self.ctx.dist.all_reduce(t, op=self.ctx.dist.ReduceOp.MIN)
        return bool(t.item() > 0.5)

    def roofline(self, G):
        """Roofline of the dominant kernel class, measured with HIP events on the launch stream (stage replay of one
        uniform pass of G batches)."""
        from ctc_attention_mispronunciation_amd.utils.data_loader import stack_features
        model = self.model
        B = G * self.B0
        Tp, L = T_RAW // 4, L_CANON
        if self.ragged:
            x, x1 = self.rag[0]["x"], self.rag[0]["x1"]
            Tp, L = x.shape[1] // 2, x1.shape[1]
        else:
            x, x1 = stack_features(self.raw[:B]), self.x1[:B]
        torch.cuda.synchronize()
        reps = [model.profile(x, x1) for _ in range(3)]
        stages = [(reps[0][i][0], float(np.median([r[i][1] for r in reps])), reps[0][i][2], reps[0][i][3]) for i in range(len(reps[0]))]
        x3 = model.precision == "bf16x3"
        x6 = model.precision == "f32x6"

        def kernel_of(name):   # which hand-written kernel a stage runs (depends on the precision mode in use)
            if name.startswith("lstm"):
                if os.environ.get("MDD_LSTM") == "step" or B > 1024:
                    return "lstm_step_x3_kernel" if x3 else "lstm_step_packed_kernel"
                if x6 and os.environ.get("MDD_LSTM_X6", "1")[0] != "0" and (self.geom.hidden == 384 and B <= 1024 or self.geom.hidden == 256 and B <= 128):
                    return "lstm_layer_x6_kernel"      # mdd_model::lx6(): the f32x6 recurrence where it is the faster reference-width kernel
                return "lstm_layer_granule_kernel" if x3 else "lstm_layer_f32_kernel"
            if name.startswith("gemm"):
                if x6 and name.startswith(("gemm_ih", "gemm_text")):
                    return "gemm_f32x6_kernel"
                if not x3:
                    return "gemm_nt_f32_kernel"
                # launch_gemm_bf16x3's dispatch: fp32-output, unbatched, >= 1024 x 512 problems take the 256x256 8-phase kernel
                return "gemm_bf16x3_ph8_kernel" if name.startswith(("gemm_ih", "gemm_text")) and B * min(Tp, L) >= 1024 else "gemm_bf16x3_glds_kernel"
            return {"conv_fused": "conv_fused_kernel", "attn_tail": "attn_tail_mfma_kernel", "embed": "embed_kernel"}.get(name, name)
        groups = {}
        for name, ms, launches, flops in stages:
            g = groups.setdefault(kernel_of(name), [0.0, 0, 0.0])
            g[0] += ms; g[1] += launches; g[2] += flops
        cand = {k: v for k, v in groups.items() if k.startswith(("lstm_", "gemm_"))}
        kern = max(cand, key=lambda k: cand[k][0])
        ms, launches, flops = cand[kern]
        bf16_kernel = kern in ("gemm_bf16x3_ph8_kernel", "gemm_bf16x3_glds_kernel", "lstm_layer_granule_kernel", "gemm_f32x6_kernel", "lstm_layer_x6_kernel")
        mult = 6 if kern in ("gemm_f32x6_kernel", "lstm_layer_x6_kernel") else (3 if bf16_kernel else 1)     # bf16 MFMA flops issued per algorithmic (fp32-equivalent) flop
        peak = PEAK_BF16_MATRIX_TFLOPS if bf16_kernel else PEAK_F32_MATRIX_TFLOPS
        achieved = flops / (ms * 1e-3) / 1e12
        traffic = None
        if os.path.exists(PMC_FILE):      # HBM bytes per launch from a separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE run of this command
            traffic = json.load(open(PMC_FILE)).get(kern)
        return {"kernel": kern, "bound": "mfma", "achieved": round(achieved, 3), "peak": peak,
                "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic,
                "note": ("algorithmic fp32-equivalent flops; this kernel issues %d bf16 MFMA flops per algorithmic flop "
                         "(matrix_core_frac = frac x%d = matrix-core utilisation)" % (mult, mult) if bf16_kernel else "fp32 MFMA"),
                "matrix_core_frac": round(achieved * mult / peak, 4),
                "frac_of_fp32_mfma_peak": round(achieved / PEAK_F32_MATRIX_TFLOPS, 3),
                "launches_per_pass": launches, "avg_launch_us": round(ms * 1e3 / max(launches, 1), 3), "batches_per_pass": G,
                "flops_per_launch": flops / max(launches, 1),
                "kernel_classes_ms": {k: round(v[0], 4) for k, v in groups.items()},
                "stage_ms": {n: round(m, 4) for n, m, _, _ in stages}}

    def close(self):
        self.model.close()
        self.sets = {}
        self.rag = []
        torch.cuda.empty_cache()


DTYPE_X3 = ("f32 results from split-bf16 arithmetic: every contraction of the model -- conv0/conv1, the BiLSTM input projections AND the "
            "recurrent W_hh.h products, text projection, score and attention-score GEMMs -- runs as bf16 hi/lo x3 on the bf16 MFMA pipes "
            "with fp32 accumulate (operands ~16 significant bits, the recurrent state h is re-split to hi+lo every step); cell state, gates, "
            "softmax and the classifier tail are fp32.  Log-probs within 1e-4 of the fp32 reference at T'=250 (tests/golden/g9_chain).  "
            "NARROWER than the reference's fp32 arithmetic: a flagged variant, not the headline")
DTYPE_F32 = ("f32 -- reference width: every contraction of the model (conv0/conv1, BiLSTM input projections, the recurrent W_hh.h products, "
             "text projection, score, attention scores, classifier) is an exact fp32 MFMA (v_mfma_f32_32x32x2_f32 / 16x16x4_f32, fp32 "
             "accumulate), gates / softmax fp32, beam scores f64 -- the arithmetic of the reference's ATen fp32 ops")
DTYPE_X6 = ("f32 -- reference width (operands with all 24 significand bits, fp32 accumulate; see `accuracy`: closer to a float64 evaluation than "
            "ATen's own fp32): the large time-batched contractions (conv0/conv1, BiLSTM and text input projections) run on the bf16 matrix cores "
            "as f32x6 = each fp32 operand as three bf16 planes hi+mid+lo (exact), the six cross products down to 2^-24, hi.hi in its own "
            "accumulator; so do the recurrent W_hh.h products (W_hh and the state h as three planes each; lstm_layer_x6_kernel); score / attention / classifier are exact fp32 MFMAs; gates / softmax fp32, beam "
            "scores f64")


def variant(ctx, args, name, **kw):
    """One secondary configuration measured in the same run (rank 0, N=1)."""
    steps, warmup = kw.pop("steps"), kw.pop("warmup")
    want_roof = kw.pop("roofline", False)
    job = DecodeJob(ctx, args, **kw)
    dt, frames = job.timed(steps, warmup)
    out = {"value": round(frames / dt, 1), "unit": "phoneme-frames/s", "ms_per_step": round(dt / steps * 1e3, 4), "steps": steps,
           "batch": job.B0, "batches_fused_per_pass": max(1, min(job.fuse, steps)), "hidden": kw["hidden"],
           "decoder": job.decoder_kind, "precision": job.model.precision}
    if job.ragged:
        out["padded_frames_per_s"] = round(frames / dt * sum(r["padded"] for r in job.rag) / sum(r["frames"] for r in job.rag), 1)
    if want_roof and not args.no_roofline:
        r = job.roofline(max(1, min(job.fuse, steps)))
        out["roofline"] = {k: r[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_launch_us", "kernel_classes_ms")}
    job.close()
    return out


def ctc_workload(ctx, args):
    """configs[4]'s CTC lattice: nn.CTCLoss(reduction='sum') forward + the gradient autograd deposits on the log-probs, global
    batch 256 x T'=250 x L=40 (one step = one loss+grad over the batch).  HBM-bound scan: the kernel must read logp once for
    alpha/beta and write the gradient once: 2 x T' x B x C x 4 bytes algorithmic."""
    from ctc_attention_mispronunciation_amd.hip_model import ctc_loss
    rs = np.random.Generator(np.random.PCG64(12 + ctx.rank))
    T, B, Cn, L = T_RAW // 4, 256, N_CLASS, L_CANON
    lp = torch.log_softmax(torch.from_numpy(rs.standard_normal((T, B, Cn)).astype(np.float32)), -1).cuda()
    tg = torch.from_numpy(rs.integers(1, Cn, size=(B, L))).cuda()
    il = torch.from_numpy(rs.integers(2 * L + 1, T + 1, size=B)).cuda()
    tl = torch.from_numpy(rs.integers(L // 2, L + 1, size=B)).cuda()
    for _ in range(max(args.warmup, 1)):
        ctc_loss(lp, tg, il, tl)
    ctx.barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        nll, grad = ctc_loss(lp, tg, il, tl)
    e1.record()
    ctx.barrier()
    dt = time.perf_counter() - t0
    kern_ms = e0.elapsed_time(e1) / args.steps
    alg_bytes = 2.0 * T * B * Cn * 4
    return dt, B * T * args.steps, {"kernel": "ctc_kernel", "bound": "hbm", "achieved": round(alg_bytes / (kern_ms * 1e-3) / 1e9, 2),
                                    "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(alg_bytes / (kern_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                                    "traffic": None, "avg_launch_us": round(kern_ms * 1e3, 2), "bytes_per_launch": alg_bytes,
                                    "note": "lattice recurrences are T' dependent steps per utterance: latency-, not bandwidth-bound at B=256"}


def train_workload(ctx, args):
    """BASELINE configs[4] per GPU: one training step of run_epoch (AA/steps/train_ctc.py:28-105) on a 32-utterance shard of the
    global batch 256 = 8 x 32: train-mode forward (batch-statistics BatchNorm, dropout 0.2), CTCLoss(sum)/B, backward, gradient
    all-reduce over the ranks (RCCL), Adam(lr 1e-3, weight_decay 5e-4).  fp32 arithmetic, as the reference trains (its config has no
    attention-CE term and no bf16: SURVEY.md 8(a) A12)."""
    import torch.nn as nn
    from ctc_attention_mispronunciation_amd import synth
    from ctc_attention_mispronunciation_amd.models.model_ctc import CTC_Model
    from ctc_attention_mispronunciation_amd.steps.train_ctc import build_training, allreduce_gradients
    B, T, L = 32, T_RAW // 2, L_CANON
    geom = synth.Geometry(feat=3 * D_RAW, hidden=args.hidden, layers=4, num_class=N_CLASS)
    sd = synth.synth_state_dict(geom, seed=1234)
    model = CTC_Model(add_cnn=True, cnn_param=geom.cnn_param(nn), rnn_param=geom.rnn_param(nn), num_class=N_CLASS, drop_out=0.2)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    model = model.cuda().train()
    model.strict_errors = False                                     # no host synchronisation inside the step
    model.train_precision = args.train_precision
    loss_fn, opt = build_training(model)
    x, x1, frac, _ = synth.synth_batch(geom, B=B, T=T, L=L, seed=1234 + ctx.rank, ragged=False)
    rs = np.random.Generator(np.random.PCG64(7 + ctx.rank))
    xd, x1d = torch.from_numpy(x).cuda(), torch.from_numpy(x1).cuda()
    tg = torch.from_numpy(rs.integers(2, 44, size=(B, L))).cuda()
    il = torch.full((B,), T // 2, dtype=torch.int64).cuda()
    tl = torch.full((B,), L, dtype=torch.int64).cuda()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    stage = np.zeros(4)

    def step(timed):
        if timed:
            ev[0].record()
        out = model(xd, x1d)
        if timed:
            ev[1].record()
        loss = loss_fn(out, tg, il, tl) / B
        opt.zero_grad()
        loss.backward()
        if timed:
            ev[2].record()
        allreduce_gradients(model)
        if timed:
            ev[3].record()
        opt.step()
        if timed:
            ev[4].record()
            torch.cuda.synchronize()
            for k in range(4):
                stage[k] += ev[k].elapsed_time(ev[k + 1])
        return loss
    for _ in range(max(args.warmup, 1)):
        first = step(False)
    ctx.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step(False)
    ctx.barrier()
    dt = time.perf_counter() - t0
    for _ in range(3):                                              # stage split, outside the timed region (it synchronises per step)
        step(True)
    if ctx.dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if ctx.dist.get_backend() == "gloo" else "cuda")
        ctx.dist.all_reduce(t, op=ctx.dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, B * (T // 2) * args.steps, {"forward_ms": round(stage[0] / 3, 3), "loss_backward_ms": round(stage[1] / 3, 3),
                                              "allreduce_ms": round(stage[2] / 3, 3), "adam_ms": round(stage[3] / 3, 3),
                                              "loss_first": round(float(first), 4), "loss_last": round(float(last), 4), "batch_per_gpu": B}


def main():
    args = parse()
    from ctc_attention_mispronunciation_amd import _lib
    _lib.require_gpu()
    ctx = Ctx(args)
    rank, world = ctx.rank, ctx.world

    if args.workload == "ctc256":
        dt, frames, roof = ctc_workload(ctx, args)
        if rank == 0:
            print(json.dumps({"metric": "phoneme-frames/sec CTC loss+grad lattice", "value": round(world * frames / dt, 1), "unit": "phoneme-frames/s",
                              "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
                              "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 log-probs, f64 lattice",
                              "data": "synthetic", "config": {"workload": "CTC alpha/beta lattice loss+grad, B=256 x T'=250 x L=40, C=45"},
                              "roofline": roof, "cpu_baseline": None}))
        return

    if args.workload == "train32":
        dt, frames, info = train_workload(ctx, args)
        if rank == 0:
            flop = 3.0 * 38.2e6 * (frames / args.steps)             # forward + ~2x for the backward, per step (38.2 MFLOP / frame)
            print(json.dumps({"metric": "phoneme-frames/sec training step (fwd + CTC + bwd + all-reduce + Adam)", "value": round(world * frames / dt, 1),
                              "unit": "phoneme-frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                              "dtype": ("f32 (exact fp32 MFMA; the reference trains in fp32)" if args.train_precision == "f32" else
                                        "f32 with the projections (forward, dX, dW_ih) and both recurrences (W_hh.h forward, DG.W_hh backward, in the persistent "
                                        "layer kernels, h and the gate gradients carried as bf16 hi/lo) as split-bf16 x3 on the bf16 matrix cores; "
                                        "conv, BatchNorm, attention, cell math, weight gradients of W_hh, CTC in fp32/fp64"), "data": "synthetic",
                              "config": {"workload": "training step, B=32 per GPU x 10 s (global batch 32 x n_gpus; BASELINE configs[4] = 8 x 32), H=%d, L=40, dropout 0.2, "
                                                     "Adam lr 1e-3 wd 5e-4" % args.hidden, **info},
                              "roofline": {"kernel": "training step (all kernels)", "bound": "mfma", "achieved": round(flop / (dt / args.steps) / 1e12, 2),
                                           "peak": PEAK_F32_MATRIX_TFLOPS, "unit": "TFLOP/s", "frac": round(flop / (dt / args.steps) / 1e12 / PEAK_F32_MATRIX_TFLOPS, 4),
                                           "traffic": None, "note": "algorithmic flops of forward x3 over the whole step; the BiLSTM recurrences run as per-step launches"},
                              "cpu_baseline": None}))
        if ctx.dist is not None:
            ctx.dist.destroy_process_group()
        return

    joint = args.workload == "joint64"
    job = DecodeJob(ctx, args, hidden=args.hidden, joint=joint, fuse=args.fuse, ragged=args.ragged, precision=args.precision,
                    decoder_kind=args.decoder, gather="none" if args.no_gather else args.gather)
    dt, frames = job.timed(args.steps, args.warmup)
    G = max(1, min(args.fuse, args.steps))
    if ctx.dist is not None:
        ft = torch.tensor([frames], dtype=torch.float64, device="cpu" if ctx.dist.get_backend() == "gloo" else "cuda")
        ctx.dist.all_reduce(ft)
        frames_all = float(ft.item())
    else:
        frames_all = float(frames)
    value = frames_all / dt
    precision = job.model.precision
    checksum = int(sum(job.aligned))
    gather_ok = job.verify_gather()
    gather_kind = job.gather
    gather_timing = None
    if job.gather is not None and job.gather_ms:
        gather_timing = {"what": job.gather, "gather_ms_median": round(float(np.median(job.gather_ms)), 4),
                         "forward_stream_gap_ms_median": round(float(np.median(job.fwd_gap_ms)), 4) if job.fwd_gap_ms else None,
                         "note": "gap = idle time of the forward stream between consecutive passes (device timestamps); the exchange of pass i runs on "
                                 "its own stream beside the forward of pass i+1"}

    roof = job.roofline(G) if (rank == 0 and not args.no_roofline) else None

    # ---- CPU baseline: the torch-CPU + Python-beam port of the reference path on a bounded sample
    cpu = None
    accuracy = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.ragged:
        from oracle import ref_port
        from ctc_attention_mispronunciation_amd.utils.data_loader import stack_features
        ncores = min(os.cpu_count() or 1, 16)
        torch.set_num_threads(ncores)
        sb, Tp = job.B0, T_RAW // 4                                    # one reference-sized batch: ~10 s of CPU work
        xs = stack_features(job.raw[:sb]).cpu().numpy()
        ref_port.forward(job.sd, xs[:1, :64], job.x1_np[:1])          # warm-up
        t1 = time.perf_counter()
        lp_cpu = ref_port.forward(job.sd, xs, job.x1_np[:sb])
        t_fwd = time.perf_counter() - t1
        # the same batch in float64 (the yardstick) beside ATen's fp32 and this run's posteriors: the accuracy the dtype string claims
        lp64 = ref_port.forward(job.sd, xs, job.x1_np[:sb], dtype=torch.float64).numpy()
        lp_gpu = job.model.forward_raw(job.raw[:sb], job.x1[:sb]).cpu().numpy().astype(np.float64)
        d_at, d_gpu = np.abs(lp_cpu.numpy().astype(np.float64) - lp64), np.abs(lp_gpu - lp64)
        accuracy = {"what": "distance of the log-probs [250, 64, 45] of one benchmarked batch to a float64 evaluation of the reference's graph (oracle/ref_port.py, CPU)",
                    "aten_fp32_cpu": {"max": float(d_at.max()), "mean": float(d_at.mean())},
                    "this_run": {"max": float(d_gpu.max()), "mean": float(d_gpu.mean()), "precision": job.model.precision},
                    "tolerance_north_star": 1e-4}
        t1 = time.perf_counter()
        if job.decoder_kind == "beam":
            ref_port.beam(lp_cpu, [Tp] * sb, job.i2c, job.decoder.lm, BEAM_W, 0.0)
        else:
            ref_port.greedy(lp_cpu, [Tp] * sb, job.i2c)
        t_dec = time.perf_counter() - t1
        cpu = {"value": round(sb * Tp / (t_fwd + t_dec), 1), "unit": "phoneme-frames/s", "cores": ncores, "kind": "port",
               "sample": "%d of the same synthetic 10 s utterances (one batch): torch-CPU forward %.2f s + %s %.2f s"
                         % (sb, t_fwd, "pure-Python beam(10)" if job.decoder_kind == "beam" else "greedy", t_dec)}
    decoder_kind = job.decoder_kind
    job.close()

    # ---- the same hot path in the configurations real data and the other BASELINE configs get (rank 0, N=1 only)
    variants = None
    if rank == 0 and world == 1 and not args.no_variants and joint and not args.ragged:
        variants = {
            "fuse1": variant(ctx, args, "fuse1", hidden=args.hidden, joint=True, fuse=1, ragged=False, precision=args.precision, steps=32, warmup=8),
            "ragged": variant(ctx, args, "ragged", hidden=args.hidden, joint=True, fuse=args.fuse, ragged=True, precision=args.precision, steps=64, warmup=16),
            "ragged_fuse1": variant(ctx, args, "ragged_fuse1", hidden=args.hidden, joint=True, fuse=1, ragged=True, precision=args.precision, steps=32, warmup=8),
            "f32_mfma_mode": variant(ctx, args, "f32_mfma_mode", hidden=args.hidden, joint=True, fuse=args.fuse, ragged=False, precision="f32", steps=32, warmup=8, roofline=True),
            "bf16x3_mode": variant(ctx, args, "bf16x3_mode", hidden=args.hidden, joint=True, fuse=args.fuse, ragged=False, precision="bf16x3", steps=64, warmup=16, roofline=True),
            "greedy32_h256": variant(ctx, args, "greedy32_h256", hidden=256, joint=False, fuse=args.fuse, ragged=False, precision=args.precision,
                                     steps=64, warmup=16),
            "greedy32_h256_fuse1": variant(ctx, args, "greedy32_h256_fuse1", hidden=256, joint=False, fuse=1, ragged=False, precision=args.precision,
                                           steps=32, warmup=8),
        }
        for mode in ("f32", "bf16x3"):          # BASELINE configs[4]'s per-GPU shard: one full training step (--workload train32 has the stage split)
            targs = argparse.Namespace(**dict(vars(args), steps=10, warmup=3, train_precision=mode))
            tdt, tframes, tinfo = train_workload(ctx, targs)
            variants["train32_" + mode] = {"value": round(tframes / tdt, 1), "unit": "phoneme-frames/s", "ms_per_step": round(tdt / targs.steps * 1e3, 3),
                                           "steps": targs.steps, "batch": 32, "precision": mode, "forward_ms": tinfo["forward_ms"],
                                           "loss_backward_ms": tinfo["loss_backward_ms"], "adam_ms": tinfo["adam_ms"]}
            torch.cuda.empty_cache()

    if rank == 0:
        B0 = 64 if joint else 32
        line = {
            "metric": "phoneme-frames/sec joint CTC-attn decode, 41-phone vocab",
            "value": round(value, 1), "unit": "phoneme-frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": {"bf16x3": DTYPE_X3, "f32": DTYPE_F32, "f32x6": DTYPE_X6}[precision],
            "data": "synthetic",
            "config": {"workload": ("joint CTC-attn decode: stack/skip + forward + beam(10) + align, B=64 x 10 s x 81-dim log-mel"
                                    if joint else "CTC-only greedy decode: stack/skip + forward + greedy + align, B=32 x 10 s"),
                       "batch_per_gpu": B0, "global_batch": B0 * world, "batches_fused_per_pass": G,
                       "lengths": "ragged: len ~ U[0.5,1] x 10 s, per-batch padding, unpadded frames counted" if args.ragged else
                                  "all utterances full length (ragged batches, fused with per-row lengths: variants.ragged; a lone batch per pass: variants.fuse1)",
                       "t_raw": T_RAW, "posterior_frames": T_RAW // 4,
                       "hidden": args.hidden, "layers": 4, "num_class": N_CLASS, "canonical_len": L_CANON,
                       "decoder": decoder_kind, "beam": BEAM_W if decoder_kind == "beam" else 0, "lm_alpha": 0.0,
                       "parallelism": "utterance-batch shards x%d%s" % (world, (", all-gather of %s on its own stream" % gather_kind) if gather_kind else ""),
                       "posteriors": "random-weight model output (flat: every frame live, beam worst case)"},
            "roofline": roof, "cpu_baseline": cpu, "accuracy": accuracy, "variants": variants,
            "edit_distance_checksum": checksum,
        }
        if gather_ok is not None:
            line["gather_verified"] = gather_ok
            line["gather_timing"] = gather_timing
        if variants:
            for k in ("fuse1", "ragged", "f32_mfma_mode", "bf16x3_mode", "greedy32_h256", "train32_f32", "train32_bf16x3"):
                if k in variants:
                    line[k] = variants[k]["value"]
        print(json.dumps(line))
    if ctx.dist is not None:
        ctx.dist.destroy_process_group()


if __name__ == "__main__":
    main()
