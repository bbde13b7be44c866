#!/usr/bin/env python3
"""Headline benchmark: candidate link-pairs scored per second on a synthetic ogbl-collab-shaped graph.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config collab] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = the pair stage of the scoring path over one batch of candidate pairs with the encoder output
resident in HBM (the reference's HeaRT / citation2 evaluation pattern, src/train/testing.py:96-121:
``propagate()`` once, then per batch ``elementwise_lin(h[a]*h[b])``, ``calc_pairwise``, ``score_func``):
endpoint gathers, q projection, PPR-thresholded node selection, PE + attention, count features, ``pairwise_lin``,
``elementwise_lin`` and the ``mlp_score`` head, scores landing in device memory.  The encoder (L x GEMM + CSR SpMM,
row-sharded with an RCCL all-gather per layer when N > 1) is timed separately and reported as ``encoder_ms``; the
throughput including one encoder pass per batch (the reference's ``test_edge`` pattern) is ``value_incl_encoder``.

Data are synthetic (no datasets offline): Chung-Lu power-law graph with ogbl-collab's node/edge counts and integer
edge weights, N(0,1) features, PPR from the library's own push (alpha 0.15, eps 5e-5), random-init weights.
Pairs are half existing edges, half uniform random; several distinct batches are cycled.

Rank 0 prints ONE JSON line.  N > 1: weak scaling, every rank scores its own 32,768-pair batches, no collective
on the pair path; time = max over ranks, value = pairs of all ranks / time.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import lpformer_amd  # noqa: E402
from lpformer_amd import data as D  # noqa: E402
from lpformer_amd import dist as LD  # noqa: E402
from lpformer_amd.profile import KernelTimer  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 = fp32 vector rate


def build_problem(cfg, rank, world, threads, ppr_device=None):
    n = cfg["n"]
    t0 = time.time()
    ei, w = D.chung_lu_graph(n, cfg["edges"], gamma=cfg["gamma"], seed=0, max_weight=cfg["max_weight"])
    x = np.random.default_rng(1).standard_normal((n, cfg["f_in"])).astype(np.float32)
    t1 = time.time()
    data = D.build_data(ei, x, n, edge_weight=w, eps=cfg["eps"], ppr_threads=threads, ppr_device=ppr_device)
    t2 = time.time()
    return ei, w, x, data, {"graph_s": t1 - t0, "ppr_s": t2 - t1}


def pair_stats(data, batch, thresholds, p1):
    """Per-batch structural sizes that the algorithmic byte/FLOP counts are built from."""
    adj, ppr = data["adj_mask"], data["ppr"]
    a, b = batch[0], batch[1]
    deg = np.diff(adj.rowptr)
    plen = np.diff(ppr.rowptr)
    p1len = np.minimum(np.diff(p1.rowptr), 256)  # longer P1 rows are binary-searched in place, not streamed
    return {"sum_deg": int(deg[a].sum() + deg[b].sum()), "sum_ppr_len": int(plen[a].sum() + plen[b].sum()),
            "sum_p1_len": int(p1len[a].sum() + p1len[b].sum())}


def slot_count(model, batch_t):
    """Candidate slots of one batch (adjacency rows of both endpoints + the shorter T0 row, at least one per pair)."""
    ws = model._select_device(batch_t, False, None)
    return int(ws.ctl[0].item())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=120)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="collab", choices=sorted(D.CONFIGS))
    ap.add_argument("--batches", type=int, default=5,
                    help="distinct candidate batches cycled through (coprime with --streams: every stream sees every batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=98304,
                    help="pairs timed on the CPU oracle (taken from the bench's own batches; about 10-15 s of CPU work)")
    ap.add_argument("--cpu-threads", type=int, default=32, help="host threads the CPU baseline may use")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-bf16", action="store_true", help="skip the extra bf16 throughput-mode measurement")
    ap.add_argument("--ppr", default="gpu", choices=("gpu", "host"),
                    help="PPR producer for the (untimed) setup: lpf_ppr_push_f64 on the GPU or the OpenMP host push")
    ap.add_argument("--spinup", type=float, default=1.0,
                    help="seconds of untimed steps after the W warm-up steps, so that the device holds its clocks")
    ap.add_argument("--no-side-stream", action="store_true",
                    help="keep the elementwise / q branches on the step's own stream (model.use_side_stream = False)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="timed windows of K steps in total; the FIRST is `value`, min/median/max of all are reported")
    ap.add_argument("--encoder", default="auto", choices=("auto", "replicated", "sharded"),
                    help="N > 1: every rank runs the whole encoder, or rows are sharded with an all-gather per layer; "
                         "auto = the cheaper one by lpformer_amd.dist.encoder_plan (measured encoder time and "
                         "measured all-gather rate)")
    ap.add_argument("--streams", type=int, default=6,
                    help="HIP streams the timed steps rotate over (consecutive batches overlap; 1 = strictly serial)")
    args = ap.parse_args()

    rank, world, local = LD.init_from_env()
    if world != args.gpus:
        if rank == 0:
            print(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE", file=sys.stderr)
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    local = int(os.environ.get("LPF_LOCAL_DEVICE", local))  # (functional multi-rank runs on a one-GPU box)
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)
    cfg = D.CONFIGS[args.config]
    n, d, bs = cfg["n"], cfg["dim"], cfg["batch"]
    host_threads = max(1, (os.cpu_count() or 8) // world)

    ei, w, x, data, setup = build_problem(cfg, rank, world, host_threads, dev if args.ppr == "gpu" else None)
    targs = D.train_args_for(cfg)
    torch.manual_seed(0)
    model = lpformer_amd.LinkTransformer(targs, data, device=dev).to(dev).eval()
    score = lpformer_amd.mlp_score(model.out_dim, model.out_dim, 1, 2).to(dev).eval()
    model.use_side_stream = not args.no_side_stream
    enc_plan = None
    if world > 1:
        # encoder layout: measure the whole encoder on one GPU (replicated mode) and the all-gather of an [N, D] fp32
        # matrix on this process group, then take the cheaper layout (or the one asked for)
        model.set_row_shard(rank, world, "replicated")
        for _ in range(2):
            model.propagate()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            model.propagate()
        torch.cuda.synchronize()
        enc1 = LD.max_over_ranks((time.perf_counter() - t0) * 1e3 / 3, dev)
        ag = LD.measure_allgather_gbps(n, d, dev)
        enc_plan = LD.encoder_plan(enc1, n, d, cfg["gnn_layers"], world, ag)
        enc_plan["allgather_gbps"] = round(ag, 1)
        enc_plan["chosen"] = enc_plan["mode"] if args.encoder == "auto" else args.encoder
        model.set_row_shard(rank, world, enc_plan["chosen"])

    # candidate batches resident in HBM before the timed region; distinct per rank
    batches_np = [D.sample_pairs(ei, n, bs, seed=1000 * rank + i) for i in range(args.batches)]
    batches = [torch.from_numpy(b).to(dev) for b in batches_np]

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    # ---- encoder output (resident for the pair stage; the encoder itself is timed after the pair stage, warm)
    for _ in range(2):
        h = model.propagate()
    torch.cuda.synchronize()

    def step(i):  # = score(cat(elementwise_lin(x_a * x_b), calc_pairwise(...))) with the boundary Linears folded
        return model.score_pairs(batches[i % len(batches)], h, score)

    # consecutive steps rotate over `--streams` HIP streams: the selection kernels of one batch (latency / issue
    # bound) run under the MFMA kernels of the previous one.  Every step still does all of its work.
    lanes = model.lanes(max(1, args.streams))

    def step_on(i):
        with torch.cuda.stream(lanes[i % len(lanes)]):
            return step(i)

    for i in range(max(args.warmup, len(lanes))):
        step_on(i)
    torch.cuda.synchronize()
    # device spin-up (untimed): a freshly started MI355X needs ~0.5 s of sustained work before it holds its clocks --
    # the first process on a cold box measures 55 M pairs/s in a 50 ms timed window and 82 M after this
    t_spin, n_spin = time.perf_counter(), 0
    while args.spinup > 0 and time.perf_counter() - t_spin < args.spinup:
        for i in range(16):
            step_on(i)
        torch.cuda.synchronize()
        n_spin += 16

    # ---- timed region: EXACTLY `steps` steps, nothing but the scoring path (no event recording)
    KernelTimer.enabled = False
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step_on(i)
    torch.cuda.synchronize()
    barrier()
    elapsed = LD.max_over_ranks(time.perf_counter() - t0, dev)
    assert torch.isfinite(out).all()

    # ---- the same window repeated (reported only: `value` is the window above); spread of the measurement
    rep_ms = [elapsed * 1e3 / args.steps]
    for _ in range(max(0, args.repeats - 1)):
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step_on(i)
        torch.cuda.synchronize()
        barrier()
        rep_ms.append(LD.max_over_ranks(time.perf_counter() - t0, dev) * 1e3 / args.steps)

    # ---- encoder, timed separately (same output: h stays valid)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    enc_reps = 5
    for _ in range(enc_reps):
        h2 = model.propagate()
    torch.cuda.synchronize()
    barrier()
    encoder_ms = LD.max_over_ranks((time.perf_counter() - t0) * 1e3 / enc_reps, dev)
    # the same with the gathered per-layer table stored in bf16 (lpf_gemm_f32_out_bf16 + lpf_spmm_csr_bf16)
    encoder_bf16 = None
    if not args.no_bf16:
        model.encoder_precision = "bf16"
        h16 = model.propagate()
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        for _ in range(enc_reps):
            h16 = model.propagate()
        torch.cuda.synchronize()
        barrier()
        encoder_bf16 = {"encoder_ms": round(LD.max_over_ranks((time.perf_counter() - t0) * 1e3 / enc_reps, dev), 4),
                        "max_abs_diff_x_node_vs_f32": float((h16 - h).abs().max())}
        model.encoder_precision = "f32"
        del h16
    del h2

    # ---- bf16 throughput mode (extra keys; the headline stays fp32 = the reference's precision): same steps with
    #      model.precision = "bf16" (bf16 storage of Z + bf16 matrix cores in the attention kernel)
    bf16 = None
    if not args.no_bf16 and d in (32, 64, 128):
        model.precision = model.tail_precision = "bf16"
        for i in range(max(args.warmup, len(lanes))):
            step_on(i)
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            out16 = step_on(i)
        torch.cuda.synchronize()
        barrier()
        el16 = LD.max_over_ranks(time.perf_counter() - t0, dev)
        model.precision = model.tail_precision = "f32"
        ref_l = model.score_pairs(batches[0], h, score, logits=True).clone()
        model.precision = model.tail_precision = "bf16"
        got_l = model.score_pairs(batches[0], h, score, logits=True)
        bf16 = {"value": round(world * bs * args.steps / el16, 1), "unit": "pairs/s",
                "ms_per_step": round(el16 * 1e3 / args.steps, 4),
                "max_abs_logit_diff_vs_f32": float((got_l - ref_l).abs().max()),
                "what": "bf16 storage of the node table Z + v_mfma_f32_32x32x16_bf16 for Wfold h, bf16 weights and "
                        "v_mfma_f32_16x16x16_bf16 for the two GEMMs of the dense tail (fp32 accumulate everywhere); "
                        "selection, q, softmax, record merge, LayerNorms in fp32; selected index sets identical to fp32"}
        KernelTimer.reset()
        KernelTimer.enabled = True
        for i in range(args.steps):
            step(i)
        kt16 = KernelTimer.summary()
        KernelTimer.enabled = False
        if "pair_attention_fused" in kt16:
            bf16["pair_attention_fused_ms"] = round(kt16["pair_attention_fused"][2], 4)
        if "tail_chain" in kt16:
            bf16["tail_chain_ms"] = round(kt16["tail_chain"][2], 4)
        model.precision = model.tail_precision = "f32"
    if bf16 is not None and encoder_bf16 is not None:
        bf16.update(encoder_bf16)
    elif encoder_bf16 is not None:
        bf16 = dict(encoder_bf16, what="bf16 storage of the aggregation's gathered table only (D = 256: the pair stage "
                                      "has no bf16 kernels)")

    # ---- instrumented replay of the same steps: HIP events around every kernel launch on the launch stream
    # (recording ~50 events per step costs ~0.1 ms per step, so it is kept out of the headline timing)
    instrumented_ms = None
    if not args.no_kernel_timing:
        KernelTimer.reset()
        KernelTimer.enabled = True
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        torch.cuda.synchronize()
        instrumented_ms = (time.perf_counter() - t1) * 1e3 / args.steps
        KernelTimer.enabled = False

    ms_per_step = elapsed * 1e3 / args.steps
    pairs_per_s = world * bs * args.steps / elapsed
    value_incl_encoder = world * bs / ((ms_per_step + encoder_ms) * 1e-3)

    kt = KernelTimer.summary() if not args.no_kernel_timing else {}
    # one instrumented encoder pass for the aggregation kernel's roofline -- on EVERY rank: a row-sharded encoder
    # all-gathers, and a collective issued by rank 0 alone would wait for ever
    KernelTimer.reset()
    KernelTimer.enabled = True
    model.propagate()
    enc = KernelTimer.summary()
    KernelTimer.enabled = False

    result = None
    if rank == 0:
        # ---- roofline: every modelled kernel, the dominant one of the timed region reported as "roofline"
        roofline, rooflines, kernels = None, {}, {}
        if kt:
            tot = sum(v[1] for v in kt.values())
            kernels = {k: {"launches": v[0], "ms_per_step": round(v[1] / args.steps, 4),
                           "share": round(v[1] / tot, 3)} for k, v in sorted(kt.items(), key=lambda kv: -kv[1][1])}
            # structural totals over the batches actually timed
            tp = [model.compute_node_mask(b) for b in batches]
            nsel = [sum(int(t[0].shape[1]) for t in sel if t is not None) for sel in tp]
            p1 = model._device_graph("p1", data["ppr"]).to_host_compact()
            stats = [dict(pair_stats(data, b, cfg["thresholds"], p1), slots=slot_count(model, bt))
                     for b, bt in zip(batches_np, batches)]
            used = [i % len(batches) for i in range(args.steps)]
            mean = lambda arr: float(np.mean([arr[i] for i in used]))  # noqa: E731
            n_sel, sum_deg = mean(nsel), mean([s["sum_deg"] for s in stats])
            sum_ppr, sum_p1 = mean([s["sum_ppr_len"] for s in stats]), mean([s["sum_p1_len"] for s in stats])
            # algorithmic work per launch: bytes for the HBM-bound kernels, FLOPs for MFMA ones
            slots = mean([s["slots"] for s in stats])
            c = model.count_dim
            work = {
                # one-pass attention (score + segment softmax + weighted sum): SURVEY 8(d) n_sel * (2 D^2 + ~20 D)
                "pair_attention_fused": ("mfma", n_sel * (2.0 * d * d + 20.0 * d)),
                # legacy two-pass kernels (D = 256 and the module-by-module API)
                "pair_scores": ("mfma", n_sel * (2.0 * d * d + 20.0 * d)),
                "pair_softmax_gather": ("hbm", n_sel * (4.0 * d + 16.0) + bs * (4.0 * (4 * d + 4) + 48.0)),
                # selection, run kernel: 8 B per candidate slot (column + aligned self-PPR, or T0 column + value),
                # one 8-byte index lookup (column + value) per adjacency candidate, 148 B per pair (descriptor, offset,
                # three segment starts), one 16-byte record per selected entry
                "select_run": ("hbm", 8.0 * slots + 8.0 * sum_deg + 148.0 * bs + 16.0 * n_sel),
                # plan kernel: two node ids, twelve row pointers, descriptor + offset per pair
                "select_plan": ("hbm", (16.0 + 12 * 8.0 + 128.0 + 8.0) * bs),
                "select_export": ("hbm", 2 * 16.0 * n_sel + 40.0 * bs),
                # q = Y[a] + Y[b]: two gathered rows in, one row out per pair
                "pair_gather_q": ("hbm", 3.0 * 4.0 * d * bs + 16.0 * bs),
                "dense_chain_score": ("mfma", 2.0 * bs * (2 * d) * (2 * d + c + 1)),  # folded first layer
                "dense_chain_mlp": ("mfma", (2.0 * bs * (2 * d * d) + 2.0 * bs * (d + c) * (2 * d + c)) / 2.0),
                "dense_chain_attn_out": ("mfma", 2.0 * bs * d * (3 * d + 4)),
                # merged dense tail: record merge + post-norm (no GEMM), first layer of pairwise_lin, folded score head
                "tail_chain": ("mfma", 2.0 * bs * ((d + c) ** 2 + 2 * d * (2 * d + c))),
                # first layer of elementwise_lin alone (score_pairs): D x D
                "dense_chain_mlp_hidden": ("mfma", 2.0 * bs * d * d),
            }
            for name, (bound, units) in work.items():
                if name not in kt:
                    continue
                dur_s = kt[name][2] * 1e-3
                peak = HBM_PEAK_GBS if bound == "hbm" else F32_MFMA_PEAK_TFLOPS
                ach = units / dur_s / (1e9 if bound == "hbm" else 1e12)
                rooflines[name] = {"kernel": name, "bound": bound, "achieved": round(ach, 2), "peak": peak,
                                   "unit": "GB/s" if bound == "hbm" else "TFLOP/s", "frac": round(ach / peak, 4),
                                   "traffic": None, "launch_ms": round(kt[name][2], 4),
                                   "launches_per_step": kt[name][0] / args.steps}
            # HBM traffic per launch: NOT measured in this run -- read from the committed rocprofv3 PMC passes
            # (FETCH_SIZE / WRITE_SIZE, gfx950-corrected; tools/collect_profiles.sh) and tagged with that file
            try:
                if args.config != "collab":
                    raise KeyError("the committed PMC passes were collected on the collab-like workload only")
                pmc_file = os.path.join("profiles", "r02_pmc_traffic.json")
                pmc = json.load(open(os.path.join(ROOT, pmc_file)))
                for name, r in rooflines.items():
                    if name in pmc["kernels"]:
                        r["traffic"] = pmc["kernels"][name]["hbm_bytes_per_launch_corrected"]
                        r["traffic_source"] = f"{pmc_file} (offline rocprofv3 PMC passes at {pmc.get('commit', '?')})"
            except (OSError, KeyError, ValueError):
                pass
            if bf16 is not None and "pair_attention_fused_ms" in bf16:
                # bf16 mode: the attention kernel is HBM-bound -- one bf16 Z row + one 16-byte record per selected
                # entry, one fp32 q row read and about one fp32 record written per pair
                byts = n_sel * (2.0 * d + 16.0) + bs * (4.0 * d + 4.0 * d + 16.0)
                ach = byts / (bf16["pair_attention_fused_ms"] * 1e-3) / 1e9
                bf16["roofline"] = {"kernel": "pair_attention_fused (bf16)", "bound": "hbm", "achieved": round(ach, 1),
                                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                                    "traffic": None, "launch_ms": bf16["pair_attention_fused_ms"]}
            modelled = [k for k in sorted(kt, key=lambda k: -kt[k][1]) if k in rooflines]
            if modelled:
                roofline = dict(rooflines[modelled[0]])
                roofline["batch_stats"] = {"n_sel": n_sel, "sum_deg": sum_deg, "sum_ppr_len": sum_ppr,
                                           "sum_p1_len": sum_p1, "slots": slots}
        # encoder aggregation kernel (outside the timed pair-stage region): SURVEY 8(d) bytes per layer
        if "spmm_csr" in enc:
            a_hat = model._device_graph("prop", data["adj_t"])
            nnz, n_rows = a_hat.nnz, n
            if world > 1 and model.encoder_mode == "sharded":  # this rank aggregates its row block only
                lo, hi = LD.row_range(n, world, rank)
                nnz, n_rows = int(a_hat.rowptr[hi] - a_hat.rowptr[lo]), hi - lo
            byts = nnz * 8.0 + 8.0 * (n_rows + 1) + 4.0 * d * nnz + 4.0 * d * n_rows
            ach = byts / (enc["spmm_csr"][2] * 1e-3) / 1e9
            # every feature row read once (all of them: any row can be a neighbour) + the local rows written once
            floor = nnz * 8.0 + 8.0 * (n_rows + 1) + 4.0 * d * n + 4.0 * d * n_rows
            ach_floor = floor / (enc["spmm_csr"][2] * 1e-3) / 1e9
            rooflines["spmm_csr"] = {"kernel": "spmm_csr (encoder, per layer)", "bound": "hbm",
                                     "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                                     "launch_ms": round(enc["spmm_csr"][2], 4),
                                     # gathered bytes (every neighbour row counted, L2 / Infinity Cache
                                     # serve most of them) above; the compulsory floor (each row once) here
                                     "achieved_compulsory_floor": round(ach_floor, 1),
                                     "frac_compulsory_floor": round(ach_floor / HBM_PEAK_GBS, 4)}
            try:
                pmc_file = os.path.join("profiles", "r02_pmc_traffic.json")
                pmc = json.load(open(os.path.join(ROOT, pmc_file)))
                if args.config == "collab" and "spmm_csr" in pmc["kernels"]:
                    rooflines["spmm_csr"]["traffic"] = pmc["kernels"]["spmm_csr"]["hbm_bytes_per_launch_corrected"]
                    rooflines["spmm_csr"]["traffic_source"] = f"{pmc_file} (offline rocprofv3 PMC passes)"
            except (OSError, KeyError, ValueError):
                pass

        # ---- CPU baseline: the oracle's pair stage on a bounded sample of the same workload
        cpu = None
        if not args.no_cpu_baseline:
            from oracle import lpformer_oracle as O
            torch.set_num_threads(os.cpu_count() or 8)
            P = {f"model.{k}": v.detach().cpu().numpy() for k, v in model.state_dict().items()}
            P.update({f"score.{k}": v.detach().cpu().numpy() for k, v in score.state_dict().items()})
            n_take = min(args.cpu_sample, bs * len(batches_np))
            sample = np.ascontiguousarray(np.concatenate(batches_np, axis=1)[:, :n_take])
            hx = h.cpu().numpy()
            mask, ppr = data["adj_mask"], data["ppr"]
            okw = dict(x=None, adj_norm=None, adj_mask=(mask.rowptr, mask.col.astype(np.int64)),
                       ppr=(ppr.rowptr, ppr.col.astype(np.int64), ppr.val), P=P, cfg=dict(targs, pred_layers=2),
                       x_node=hx)
            # the numpy port is one thread per call: run it over chunks of the sample on a pool of host threads (numpy
            # releases the GIL inside its kernels), BLAS pinned to one thread per call; `cores` = threads used
            from concurrent.futures import ThreadPoolExecutor
            n_thr = max(1, min(os.cpu_count() or 1, args.cpu_threads))
            try:
                from threadpoolctl import threadpool_limits
                limiter = threadpool_limits(limits=1)
            except ImportError:
                limiter = None
            chunk = max(256, n_take // (4 * n_thr))
            spans = [(i, min(i + chunk, n_take)) for i in range(0, n_take, chunk)]
            t0 = time.perf_counter()
            with ThreadPoolExecutor(max_workers=n_thr) as pool:
                refs = list(pool.map(lambda se: O.forward(sample[:, se[0]:se[1]], **okw)["logit"], spans))
            cpu_s = time.perf_counter() - t0
            if limiter is not None:
                limiter.restore_original_limits()
            ref_logit = np.concatenate(refs)
            # check the GPU scores of the same pairs against it while we are here
            gl = np.concatenate([model.score_pairs(torch.from_numpy(sample[:, i:i + bs]).to(dev), h, score, logits=True)
                                 .cpu().numpy() for i in range(0, n_take, bs)])
            cpu = {"value": round(n_take / cpu_s, 1), "unit": "pairs/s", "cores": n_thr,
                   "kind": "port",
                   "sample": f"first {n_take} pairs of the bench's batches (pair stage, encoder output resident) in "
                             f"chunks of {chunk} on {n_thr} host threads of {os.cpu_count()} cores; numpy restatement "
                             f"oracle/lpformer_oracle.py, {cpu_s:.1f} s",
                   "max_abs_logit_diff_vs_gpu": float(np.abs(gl - ref_logit).max())}

        result = {
            "metric": "candidate link-pairs scored/sec (whole node)", "value": round(pairs_per_s, 1),
            "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}-like synthetic graph (N={n}, {cfg['edges']} undirected edges, "
                                   f"F={cfg['f_in']}, D={d}, L={cfg['gnn_layers']}, thresholds={cfg['thresholds']}, "
                                   f"PPR eps={cfg['eps']}), {bs} candidate pairs per GPU per step, pair stage with "
                                   "encoder output resident",
                       "pairs_per_step_per_gpu": bs, "distinct_batches": len(batches),
                       "streams": len(lanes), "spinup_s": args.spinup,
                       "parallelism": (f"pairs sharded x{world}, encoder {enc_plan['chosen']}" +
                                       (" (rows + all-gather per layer)" if enc_plan["chosen"] == "sharded" else
                                        " (every rank runs it, no exchange)")) if world > 1 else "single GPU",
                       "encoder_plan": enc_plan},
            "encoder_ms": round(encoder_ms, 4), "value_incl_encoder": round(value_incl_encoder, 1),
            "ms_per_step_instrumented": None if instrumented_ms is None else round(instrumented_ms, 4),
            "ms_per_step_repeats": {"n": len(rep_ms), "min": round(min(rep_ms), 4),
                                    "median": round(float(np.median(rep_ms)), 4), "max": round(max(rep_ms), 4)},
            "roofline": roofline, "cpu_baseline": cpu, "bf16_mode": bf16, "kernels": kernels, "rooflines": rooflines,
            "setup_s": dict({k: round(v, 2) for k, v in setup.items()}, ppr_producer=args.ppr),
        }
        print(json.dumps(result), flush=True)
    barrier()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
