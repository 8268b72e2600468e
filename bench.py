#!/usr/bin/env python3
"""Contract bench: phoneme-frames/s of the joint CTC-attention decode hot path on MI355X.

One "step" = one pass of the hot path over one reference-sized batch of synthetic utterances that is
already resident in HBM: stack/skip (A1) -> CTC_Model.forward (A2-A7) -> CTC prefix beam search, width 10
(A9) -> decoded ids back on the host -> edit-distance alignment against the canonical phonemes (A10).
Workload (BASELINE.json configs[2], the configuration the metric is quoted on): B=64 utterances of 10 s
(1000 x 81 log-mel+energy frames -> 250 posterior frames each), 4xBiLSTM-384, 41-phone set (45 classes),
canonical length 40, beam 10, lm_alpha 0.  `--workload greedy32` runs configs[1] (B=32, greedy decode).

N>1 (launched by torch.distributed.run, one rank per GPU): utterance batches shard across ranks (weak
scaling: every rank decodes its own 64-utterance batch) and the posteriors of all shards are all-gathered
over RCCL/xGMI each step, as BASELINE.json's north_star describes.

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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--workload", default="joint64", choices=["joint64", "greedy32"])
    ap.add_argument("--hidden", type=int, default=384)
    ap.add_argument("--lanes", type=int, default=1, help="(must be 1) independent stream pipelines per GPU: two forwards in flight on one "
                    "device would put two persistent BiLSTM launches side by side, each waiting for CUs the other holds")
    ap.add_argument("--fuse", type=int, default=8, help="reference-sized batches carried by one launch sequence")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true", help="skip the posterior all-gather at N>1")
    ap.add_argument("--decoder", default=None, choices=[None, "beam", "greedy", "none"], help="diagnostic override of the decode stage")
    ap.add_argument("--no-roofline", action="store_true", help="skip the stage-replay pass (for clean traces)")
    args = ap.parse_args()
    if args.lanes != 1:
        ap.error("--lanes must be 1: a persistent BiLSTM layer needs every CU of the device; batch more work per pass with --fuse instead")
    return args


def main():
    args = parse()
    from ctc_attention_mispronunciation_amd import synth, _lib
    from ctc_attention_mispronunciation_amd.hip_model import HipModel
    from ctc_attention_mispronunciation_amd.utils.ctcDecoder import GreedyDecoder, BeamDecoder, align_ids_batch
    from ctc_attention_mispronunciation_amd.utils.data_loader import stack_features

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    _lib.require_gpu()
    ndev = torch.cuda.device_count()
    if local >= ndev:            # rehearsal of the N>1 path on a box with fewer GPUs than ranks (MDD_DIST_BACKEND=gloo)
        local = local % ndev
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("MDD_DIST_BACKEND", "nccl")   # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    joint = args.workload == "joint64"
    import math
    G = max(1, min(args.fuse, args.steps))            # batches fused per pass; a step stays one 64-utterance batch.  A step
    # count that is not a multiple of G ends with one smaller pass (its own buffers and captured graph, warmed up too).
    B0 = 64 if joint else 32
    B = B0 * G
    T_raw, D, L, beam_w = 1000, 81, 40, 10
    geom = synth.Geometry(feat=243, hidden=args.hidden, layers=4, num_class=45)
    sd = synth.synth_state_dict(geom, seed=1234)
    # every rank decodes a different shard (different seed), same shapes
    raw = torch.from_numpy(synth.synth_raw_features(B, T_raw, D, seed=1234 + rank)).cuda()
    _, x1_np, _, _ = synth.synth_batch(geom, B=B, T=T_raw // 2, L=L, seed=1234 + rank, ragged=False)
    x1 = torch.from_numpy(x1_np).cuda()
    model = HipModel(geom, sd, device=local)
    i2c = synth.phone_table_41()
    arpa = os.path.join(ROOT, "tests", "golden", "lm_synth45.arpa")
    use_beam = joint if args.decoder is None else args.decoder == "beam"
    decoder = (BeamDecoder(i2c, beam_width=beam_w, blank_index=0, space_idx=-1, lm_path=arpa, lm_alpha=0.0)
               if use_beam else GreedyDecoder(i2c, space_idx=-1, blank_index=0))
    Tp = T_raw // 4
    lens = torch.full((B,), Tp, dtype=torch.int32, device="cuda")
    canon_mat = np.ascontiguousarray(x1_np, dtype=np.int32)
    canon_len = np.full((B,), canon_mat.shape[1], dtype=np.int32)

    # Software pipeline.  `lanes` independent batch pipelines are in flight at once (each with its own library
    # handle = its own workspace and captured graphs): the BiLSTM recurrence is a chain of ~1000 dependent
    # ~5 us launches per batch that leaves most CUs idle, so interleaving independent batches on separate
    # HIP streams hides that latency.  Inside a lane the forward of its next batch (stream s_fwd) overlaps the
    # latency-bound beam search of its previous one (s_dec); the host aligns finished batches meanwhile.
    class Lane(object):
        class Bufs(object):
            def __init__(self, g):
                b = g * B0
                self.b = b
                self.raw = raw[:b]
                self.x1 = x1[:b]
                self.lens = lens[:b]
                self.logp = [torch.empty((Tp, b, geom.num_class), device="cuda") for _ in range(2)]
                self.gathered = ([torch.empty((world, Tp, b, geom.num_class), device="cuda") for _ in range(2)]
                                 if world > 1 else None)
                self.h_ids = [torch.empty((b, Tp), dtype=torch.int32).pin_memory() for _ in range(2)]
                self.h_n = [torch.empty((b,), dtype=torch.int32).pin_memory() for _ in range(2)]

        def __init__(self, mdl):
            self.model = mdl
            self.s_fwd, self.s_dec = torch.cuda.Stream(), torch.cuda.Stream()
            self.bufs = {}
            self.ev_fwd = [torch.cuda.Event() for _ in range(2)]
            self.ev_dec = [torch.cuda.Event() for _ in range(2)]
            self.ev_free = [torch.cuda.Event() for _ in range(2)]
            self.count = 0

        def enqueue(self, g):
            if g not in self.bufs:
                self.bufs[g] = Lane.Bufs(g)
            bf = self.bufs[g]
            k = self.count & 1
            self.count += 1
            with torch.cuda.stream(self.s_fwd):
                self.s_fwd.wait_event(self.ev_free[k])          # slot k's buffers no longer read by the decoder two passes ago
                self.model.forward_raw(bf.raw, bf.x1, out=bf.logp[k])    # stack/skip folded into the front-end's tile load
                if bf.gathered is not None and not args.no_gather:
                    dist.all_gather_into_tensor(bf.gathered[k].view(-1, bf.b, geom.num_class), bf.logp[k])
                self.ev_fwd[k].record(self.s_fwd)
            with torch.cuda.stream(self.s_dec):
                self.s_dec.wait_event(self.ev_fwd[k])
                if args.decoder != "none":
                    out = decoder.decode_ids(bf.logp[k], bf.lens)
                    bf.h_ids[k].copy_(out[0], non_blocking=True)
                    bf.h_n[k].copy_(out[1], non_blocking=True)
                self.ev_free[k].record(self.s_dec)
                self.ev_dec[k].record(self.s_dec)
            return (g, k)

        def finish(self, gk):
            g, k = gk
            bf = self.bufs[g]
            self.ev_dec[k].synchronize()
            tot = 0
            if args.decoder != "none":
                dist = align_ids_batch(bf.h_ids[k].numpy(), bf.h_n[k].numpy(), canon_mat[:bf.b], canon_len[:bf.b])[0]   # one native call per pass
                tot = int(dist[dist >= 0].sum())
            aligned.append(tot)

    aligned = []
    lanes = [Lane(model)] + [Lane(HipModel(geom, sd, device=local)) for _ in range(args.lanes - 1)]

    def passes(nsteps):
        return [G] * (nsteps // G) + ([nsteps % G] if nsteps % G else [])

    def run(sizes):
        pending = []
        for i, g in enumerate(sizes):
            ln = lanes[i % len(lanes)]
            pending.append((ln, ln.enqueue(g)))
            if len(pending) > len(lanes):                    # keep at most one finished-but-unaligned pass per lane
                l0, k0 = pending.pop(0)
                l0.finish(k0)
        for l0, k0 in pending:
            l0.finish(k0)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    timed = passes(args.steps)
    warm = passes(max(args.warmup, 1))
    for g in set(timed) - set(warm):                          # every pass size of the timed region is captured beforehand
        warm.append(g)
    run(warm)
    barrier()
    aligned.clear()
    t0 = time.perf_counter()
    run(timed)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    frames = world * B0 * Tp * args.steps
    value = frames / dt

    # ---- roofline of the dominant kernel, measured with HIP events on the launch stream (stage replay)
    roof = None
    stages = None
    if rank == 0 and not args.no_roofline:
        x = stack_features(raw)
        torch.cuda.synchronize()
        reps = [model.profile(x, x1) for _ in range(3)]
        stages = [(reps[0][i][0], float(np.median([r[i][1] for r in reps])), reps[0][i][2], reps[0][i][3]) for i in range(len(reps[0]))]
        # kernel classes: which hand-written kernel a stage runs (depends on the precision mode in use)
        x3 = model.precision == "bf16x3"
        def kernel_of(name):
            if name.startswith("lstm"):
                return "lstm_layer_granule_kernel" if (x3 and B <= 512) else ("lstm_layer_persistent_kernel" if x3 else "lstm_step_packed_kernel")
            if name.startswith("gemm"):
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
        bf16_kernel = kern in ("gemm_bf16x3_ph8_kernel", "gemm_bf16x3_glds256_kernel", "gemm_bf16x3_glds_kernel", "lstm_layer_granule_kernel", "lstm_layer_persistent_kernel")
        peak = PEAK_BF16_MATRIX_TFLOPS if bf16_kernel else PEAK_F32_MATRIX_TFLOPS
        achieved = flops / (ms * 1e-3) / 1e12
        traffic = None
        pmc_file = os.path.join(ROOT, "profiles", "round1_pmc_traffic.json")
        if os.path.exists(pmc_file):      # HBM bytes per launch from a separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE run of this command
            traffic = json.load(open(pmc_file)).get(kern)
        roof = {"kernel": kern, "bound": "mfma", "achieved": round(achieved, 3), "peak": peak,
                "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic,
                "note": ("algorithmic fp32-equivalent flops; the split-bf16 kernels issue 3 bf16 MFMA flops per algorithmic flop "
                         "(frac x3 = matrix-core utilisation)" if bf16_kernel else "fp32 MFMA"),
                "matrix_core_frac": round(achieved * (3 if bf16_kernel else 1) / peak, 4),
                "frac_of_fp32_mfma_peak": round(achieved / PEAK_F32_MATRIX_TFLOPS, 3),
                "launches_per_pass": launches, "avg_launch_us": round(ms * 1e3 / max(launches, 1), 3), "batches_per_pass": G,
                "flops_per_launch": flops / max(launches, 1),
                "kernel_classes_ms": {k: round(v[0], 4) for k, v in groups.items()},
                "stage_ms": {n: round(m, 4) for n, m, _, _ in stages}}

    # ---- CPU baseline: the torch-CPU + Python-beam port of the reference path on a bounded sample
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import ref_port
        ncores = min(os.cpu_count() or 1, 16)
        torch.set_num_threads(ncores)
        sb = min(B0, 64)                                              # one reference-sized batch: ~10 s of CPU work
        xs = stack_features(raw[:sb]).cpu().numpy()
        ref_port.forward(sd, xs[:1, :64], x1_np[:1])                       # warm-up
        t1 = time.perf_counter()
        lp_cpu = ref_port.forward(sd, xs, x1_np[:sb])
        t_fwd = time.perf_counter() - t1
        t1 = time.perf_counter()
        if joint:
            ref_port.beam(lp_cpu, [Tp] * sb, i2c, decoder.lm, beam_w, 0.0)
        else:
            ref_port.greedy(lp_cpu, [Tp] * sb, i2c)
        t_dec = time.perf_counter() - t1
        cpu = {"value": round(sb * Tp / (t_fwd + t_dec), 1), "unit": "phoneme-frames/s", "cores": ncores, "kind": "port",
               "sample": "%d of the same synthetic 10 s utterances (one batch): torch-CPU forward %.2f s + %s %.2f s"
                         % (sb, t_fwd, "pure-Python beam(10)" if joint else "greedy", t_dec)}

    if rank == 0:
        line = {
            "metric": "phoneme-frames/sec joint CTC-attn decode, 41-phone vocab",
            "value": round(value, 1), "unit": "phoneme-frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (time-batched contractions as split-bf16 x3 on bf16 MFMA with fp32 accumulate; log-probs within 1e-5 of fp32)" if model.precision == "bf16x3" else "f32",
            "data": "synthetic",
            "config": {"workload": ("joint CTC-attn decode: stack/skip + forward + beam(10) + align, B=64 x 10 s x 81-dim log-mel"
                                    if joint else "CTC-only greedy decode: stack/skip + forward + greedy + align, B=32 x 10 s"),
                       "batch_per_gpu": B0, "global_batch": B0 * world, "batches_fused_per_pass": G, "t_raw": T_raw, "posterior_frames": Tp,
                       "hidden": args.hidden, "layers": 4, "num_class": 45, "canonical_len": L,
                       "beam": beam_w if joint else 0, "lm_alpha": 0.0, "batches_in_flight": args.lanes,
                       "parallelism": "utterance-batch shards x%d%s" % (world, ", all-gather posteriors" if world > 1 and not args.no_gather else ""),
                       "posteriors": "random-weight model output (flat: every frame live, beam worst case)"},
            "roofline": roof, "cpu_baseline": cpu,
            "edit_distance_checksum": int(sum(aligned)),
        }
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
