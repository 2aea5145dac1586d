"""bench.py -- decode-steps/s of the caption-decoder hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W] [--mode forward|greedy] [--config cfg2]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One *step* = one pass of the hot path over one batch of synthetic input resident in HBM, on cfg2 =
B 64 x L 20 decode positions, 14x14x2048 features, K=20 knowledge rows, vocab 10k => 1280 decode-steps
per step and per GPU (weak scaling: every rank owns its own 64-sample shard):
  --mode train (default)  Encoder.conv1 + DecoderTransformer forward + packed cross entropy + backward +
                          ONE all-reduce of the flat gradient bucket (RCCL) + clamp + Adam: the body of
                          train.py's loop (geo-aware/train.py:269-292), dropout on as the reference has it
  --mode forward          Encoder.conv1 + teacher-forced forward only (validate(), no collective)
  --mode greedy           Encoder.conv1 + predict() greedy decode, KV-cached (eval.py path; cfg5)
Rank 0 prints ONE JSON line with the whole-job rate plus
  roofline      the dominant kernel (fp32 MFMA GEMM of the feature projection) timed with HIP events
                on its launch stream inside the timed region, against the 157.3 TFLOP/s fp32 matrix peak
  cpu_baseline  the same workload on the host cores through oracle/stock.py (a port of the
                reference's PyTorch-CPU path), bounded to ~10-20 s.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONV1_HBM_BYTES = 161.9e6   # measured, see profiles/r01_f_pmc_conv1.txt
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, 64 FLOP/clk/SIMD
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", default="train", choices=["train", "forward", "greedy"])
    ap.add_argument("--config", default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def usable_cores():
    """Host cores this process may really use: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except (OSError, ValueError, IndexError):
            pass
    return int(os.environ.get("ICK_CPU_THREADS", min(n, 16)))


def cpu_baseline(cfg, seed, budget_s, mode="forward"):
    """Time the oracle's stock-module port of the reference on the host cores (same seeded workload):
    forward = conv1 + teacher-forced forward (eval); train = conv1 + forward + CE + backward + clamp + Adam
    with the reference's default dropouts (train.py never passes its own, geo-aware/train.py:71-78)."""
    import ick_amd.synth as synth
    from oracle.stock import StockDecoder
    variant, B, L, K, V, Fn = cfg["variant"], cfg["B"], cfg["L"], cfg["K"], cfg["V"], cfg["F"]
    cores = usable_cores()
    torch.set_num_threads(cores)
    P = synth.make_params(variant, V, seed)
    cw, cb = synth.make_conv1(seed)
    wm = synth.make_word_map(V)
    batch = synth.make_batch(variant, B, L, K, V, Fn, seed)
    feats = synth.make_feats(B, seed)
    if mode == "train":
        from oracle import restatement as R
        m = StockDecoder(variant, wm, dropout=0.5).load_reference_params(P, cw, cb).train()
        params = [p for n, p in m.named_parameters() if not n.startswith("conv1")]
        opt = torch.optim.Adam(params, lr=4e-4)

        def one():
            with torch.no_grad():
                enc = m.encode_image(feats)
            scores, caps, dl = m(batch["captions"], enc, batch["caption_masks"], batch["caption_lengths"],
                                 batch["entities"], batch.get("facts"))
            loss = R.packed_ce_loss(m.cfg, scores, caps, dl)
            opt.zero_grad()
            loss.backward()
            for p in params:
                if p.grad is not None:
                    p.grad.clamp_(-5.0, 5.0)
            opt.step()
    else:
        m = StockDecoder(variant, wm).load_reference_params(P, cw, cb).eval()

        def one():
            with torch.no_grad():
                enc = m.encode_image(feats)
                return m(batch["captions"], enc, batch["caption_masks"], batch["caption_lengths"], batch["entities"],
                         batch.get("facts"))

    one()
    one()
    times = []
    t_end = time.time() + budget_s
    while time.time() < t_end or len(times) < 3:
        t0 = time.time()
        one()
        times.append(time.time() - t0)
    best = min(times)
    return {"value": B * L / best, "unit": "decode-steps/s", "cores": cores, "kind": "port",
            "sample": "%d full %s %s passes (B=%d L=%d V=%d) in %.1f s; best pass %.1f ms"
                      % (len(times), variant, "train steps (conv1 + fwd + CE + bwd + clamp + Adam, dropout on)"
                         if mode == "train" else "conv1 + teacher-forced forward", B, L, V, sum(times), best * 1e3),
            "threads": torch.get_num_threads()}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with %d ranks" % (args.gpus, args.gpus))
    import torch.distributed as dist
    torch.cuda.set_device(local_rank % max(1, torch.cuda.device_count()))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("ICK_BENCH_BACKEND", "nccl")   # "gloo" lets several ranks share one GPU (tests)
        kw = {"device_id": torch.device("cuda", local_rank)} if backend == "nccl" else {}
        try:
            dist.init_process_group(backend=backend, **kw)
            probe = torch.ones(1, device="cuda")
            dist.all_reduce(probe)          # first collective builds the RCCL communicators: fail here, not mid-run
        except Exception as e:              # noqa: BLE001 -- keep the scaling run alive on a broken RCCL setup
            sys.stderr.write("bench.py: backend %s failed (%s); falling back to gloo\n" % (backend, e))
            if dist.is_initialized():
                dist.destroy_process_group()
            dist.init_process_group(backend="gloo")

    import ick_amd
    import ick_amd.ops as ops
    import ick_amd.synth as synth

    cfgname = args.config or ("cfg5" if args.mode == "greedy" else "cfg2")
    cfg = dict(synth.CONFIGS[cfgname])
    variant, B, L, K, V, Fn = cfg["variant"], cfg["B"], cfg["L"], cfg["K"], cfg["V"], cfg["F"]
    seed = 100 + rank  # every rank owns a different shard of the global batch
    m = ick_amd.load_models(variant)
    dec = m.DecoderTransformer(synth.make_word_map(V), 300, 512, 512, 10, 3)
    dec.load_state_dict(synth.make_params(variant, V, 0), strict=False)  # same weights on every rank
    dec = dec.cuda()
    dec = dec.train() if args.mode == "train" else dec.eval()
    enc = m.Encoder(emb_dim=300)
    cw, cb = synth.make_conv1(0)
    with torch.no_grad():
        enc.conv1.weight.copy_(cw)
        enc.conv1.bias.copy_(cb)
    enc = enc.cuda().eval()
    batch = {k: v.cuda() for k, v in synth.make_batch(variant, B, L, K, V, Fn, seed).items()}
    feats = synth.make_feats(B, seed).cuda()
    extra = [batch["facts"]] if variant != "geo" else []

    if args.mode == "train":
        from ick_amd.training import TrainStep
        ts = TrainStep(dec, lr=4e-4, grad_clip=5.0, seed=rank)   # all-reduces its bucket when world > 1

        def step():
            with torch.no_grad():
                e = enc(feats)
            return ts(batch["captions"], e, batch["caption_masks"], batch["caption_lengths"], batch["entities"],
                      *extra)
        units_per_step = B * L
    elif args.mode == "forward":
        def step():
            with torch.no_grad():
                e = enc(feats)
                return dec(batch["captions"], e, batch["caption_masks"], batch["caption_lengths"], batch["entities"],
                           *extra)
        units_per_step = B * L
    else:
        def step():
            e = enc(feats)
            return dec.predict(e, L, batch["entities"], *extra)
        units_per_step = B * L

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    P = 196
    dom_shape = (B * P, 300, 2048)  # Encoder.conv1 as a GEMM: the largest single kernel of the pass
    ops.TIMED = {"shape": dom_shape, "events": []}
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    events = ops.TIMED["events"]
    ops.TIMED = None
    tmax = torch.tensor([dt], device="cuda", dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = tmax.item()

    if rank == 0:
        kern_ms = sum(a.elapsed_time(b) for a, b in events) / max(1, len(events))
        flops = 2.0 * dom_shape[0] * dom_shape[1] * dom_shape[2]
        achieved = flops / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else 0.0
        out = {
            "metric": "decode_steps_per_sec", "value": world * units_per_step * args.steps / dt,
            "unit": "decode-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %s variant, per-GPU batch %d x %d decode positions, 14x14x2048 features, "
                                   "K=%d knowledge rows%s, vocab %d" % (cfgname, variant, B, L, K,
                                                                         (", F=%d facts" % Fn) if Fn else "", V),
                       "mode": {"train": "train_step (Encoder.conv1 + forward + packed CE + backward + RCCL all-reduce "
                                         "+ clamp + Adam; dropout 0.5/0.5/0.1 as the reference's train.py builds it)",
                                "forward": "teacher_forced_forward (Encoder.conv1 + DecoderTransformer.forward)",
                                "greedy": "greedy_decode (Encoder.conv1 + predict, KV-cached)"}[args.mode],
                       "global_batch": world * B,
                       "parallelism": ("dp%d (one flat-bucket all-reduce per step)" if args.mode == "train"
                                       else "dp%d (independent shards)") % world},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_MFMA_TFLOPS,
                         # HBM bytes per launch from the PMC counters of this kernel on this workload (separate
                         # rocprofv3 --pmc passes, FETCH_SIZE x2 + WRITE_SIZE as MI355X_MICROARCH.md prescribes;
                         # tools/pmc_conv1.sh -> profiles/r01_f_pmc_conv1.txt); algorithmic 120.3 MB
                         "traffic": CONV1_HBM_BYTES if dom_shape == (64 * 196, 300, 2048) else None,
                         "kernel": "gemm_kernel<4,2,2,2,A k-major,B k-contig,vec> 128x64 tiles, 8 waves (Encoder.conv1: [%d x %d] x [%d x %d])"
                                   % (dom_shape[0], dom_shape[2], dom_shape[2], dom_shape[1]),
                         "kernel_ms": kern_ms, "flops_per_launch": flops, "launches_timed": len(events)},
        }
        # whole-pass view: measured decode-steps/s against the bound of the pass's algorithmic work
        # at the same peaks -- teacher-forced passes are MFMA-bound (a train step is ~3x the forward FLOPs), the
        # KV-cached greedy decode is HBM-bound
        bounds = {("cfg2", "forward"): 3.96e6, ("cfg2", "train"): 3.96e6 / 3, ("cfg4", "forward"): 1.99e6,
                  ("cfg4", "train"): 1.99e6 / 3, ("cfg5", "greedy"): 3.39e6}
        bnd = bounds.get((cfgname, args.mode))
        if bnd is not None:
            out["roofline"]["pass_bound_steps_per_s"] = bnd * world
            out["roofline"]["pass_frac"] = out["value"] / (bnd * world)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(cfg, seed, args.cpu_seconds, args.mode if args.mode == "train" else "forward")
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
