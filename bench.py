"""bench.py -- decode-steps/s of the caption-decoder hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W] [--mode train|forward|greedy|beam] [--config cfg2]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One *step* = one pass of the hot path over one batch of synthetic input resident in HBM, on cfg2 =
B 64 x L 20 decode positions, 14x14x2048 features, K=20 knowledge rows, vocab 10k => 1280 decode-steps
per step and per GPU (weak scaling: every rank owns its own 64-sample shard):
  --mode train (default)  Encoder.conv1 + DecoderTransformer forward + packed cross entropy + backward +
                          ONE all-reduce of the flat gradient bucket (RCCL) + clamp + Adam: the body of
                          train.py's loop (geo-aware/train.py:269-292), dropout on as the reference has it
  --mode forward          Encoder.conv1 + teacher-forced forward only (validate(), no collective)
  --mode greedy           Encoder.conv1 + predict() greedy decode, KV-cached (eval.py path; cfg5)
  --mode beam             Encoder.conv1 + predict_beam() (beam 5, batch 32: north_star cfg5; parity-unpinned)
Rank 0 prints ONE JSON line with the whole-job rate plus
  roofline      by_kernel: every kernel class of the step (launches per step, mean duration from HIP events on
                the launch stream in an eager single-stream pass, algorithmic FLOP or bytes, fraction of the fp32
                MFMA / HBM peak); the headline kernel/achieved/frac describe the class with the LARGEST share of
                the step's kernel time; pass_frac = the whole step against the bound of its algorithmic work
  cpu_baseline  the same workload on the host cores through oracle/stock.py (a port of the reference's
                PyTorch-CPU path): all usable cores and a 1-thread leg, each bounded to a few seconds.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TRAFFIC_TABLE = os.path.join(ROOT, "profiles", "r02_traffic.json")   # PMC bytes per launch, see its "_source"
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, 64 FLOP/clk/SIMD
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", default="train", choices=["train", "forward", "greedy", "beam"])
    ap.add_argument("--config", default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-profile", action="store_true", help="skip the eager per-kernel pass (roofline.by_kernel)")
    ap.add_argument("--profile-steps", type=int, default=3)
    return ap.parse_args()


def usable_cores():
    """Host cores this process may really use: affinity mask, capped by the cgroup CPU quota (no other cap)."""
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
    return int(os.environ.get("ICK_CPU_THREADS", n))


def cpu_baseline(cfg, seed, budget_s, mode="forward"):
    """Time the oracle's stock-module port of the reference on the host cores (same seeded workload):
    forward = conv1 + teacher-forced forward (eval); train = conv1 + forward + CE + backward + clamp + Adam
    with the reference's default dropouts (train.py never passes its own, geo-aware/train.py:71-78); greedy =
    conv1 + predict() exactly as the reference decodes (batch 1, full recompute of all positions per step, no KV
    cache), on a sample of the batch's captions.  Two legs: all usable cores, then one thread."""
    import ick_amd.synth as synth
    from oracle.stock import StockDecoder
    variant, B, L, K, V, Fn = cfg["variant"], cfg["B"], cfg["L"], cfg["K"], cfg["V"], cfg["F"]
    P = synth.make_params(variant, V, seed)
    cw, cb = synth.make_conv1(seed)
    wm = synth.make_word_map(V)
    batch = synth.make_batch(variant, B, L, K, V, Fn, seed)
    feats = synth.make_feats(B, seed)
    units = B * L
    if mode == "train":
        from oracle import restatement as R
        m = StockDecoder(variant, wm, dropout=0.5).load_reference_params(P, cw, cb).train()
        params = [p for n, p in m.named_parameters() if not n.startswith("conv1")]
        opt = torch.optim.Adam(params, lr=4e-4)
        what = "train steps (conv1 + fwd + CE + bwd + clamp + Adam, dropout on)"

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
    elif mode == "greedy":
        m = StockDecoder(variant, wm).load_reference_params(P, cw, cb).eval()
        nb = 2                          # captions decoded per timed pass (the reference decodes them one by one)
        units = nb * L
        what = "conv1 + predict() of %d captions, batch 1, full recompute per step (reference semantics)" % nb

        def one():
            with torch.no_grad():
                enc = m.encode_image(feats[:nb])
                for b in range(nb):
                    m.predict(enc[b:b + 1], L, batch["entities"][b:b + 1],
                              None if "facts" not in batch else batch["facts"][b:b + 1])
    else:
        m = StockDecoder(variant, wm).load_reference_params(P, cw, cb).eval()
        what = "conv1 + teacher-forced forward"

        def one():
            with torch.no_grad():
                enc = m.encode_image(feats)
                return m(batch["captions"], enc, batch["caption_masks"], batch["caption_lengths"], batch["entities"],
                         batch.get("facts"))

    def leg(threads, budget, min_runs):
        torch.set_num_threads(threads)
        one()
        times = []
        t_end = time.time() + budget
        while time.time() < t_end or len(times) < min_runs:
            t0 = time.time()
            one()
            times.append(time.time() - t0)
        best = min(times)
        return {"value": units / best, "threads": torch.get_num_threads(),
                "sample": "%d %s %s (B=%d L=%d V=%d) in %.1f s; best pass %.1f ms"
                          % (len(times), variant, what, B, L, V, sum(times), best * 1e3)}

    cores = usable_cores()
    full = leg(cores, budget_s * 0.6, 3)
    single = leg(1, budget_s * 0.4, 1) if cores > 1 else dict(full)
    return {"value": full["value"], "unit": "decode-steps/s", "cores": cores, "kind": "port", "sample": full["sample"],
            "threads": full["threads"], "one_thread": {"value": single["value"], "cores": 1, "sample": single["sample"]}}


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
    backend = "none"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm.  ICK_BENCH_BACKEND=gloo is an explicit opt-in (several ranks sharing one GPU in the
        # tests); there is NO silent fallback: if RCCL cannot initialise, the run fails with RCCL's error.
        backend = os.environ.get("ICK_BENCH_BACKEND", "nccl")
        kw = {"device_id": torch.device("cuda", local_rank)} if backend == "nccl" else {}
        dist.init_process_group(backend=backend, **kw)
        probe = torch.ones(1, device="cuda")
        dist.all_reduce(probe)              # the first collective builds the communicators: fail here, not mid-run
        torch.cuda.synchronize()
        assert probe.item() == world, "all-reduce probe returned %s for %d ranks" % (probe.item(), world)
        backend = dist.get_backend()

    import ick_amd
    import ick_amd.profiling as prof
    import ick_amd.synth as synth

    cfgname = args.config or ("cfg5" if args.mode in ("greedy", "beam") else "cfg2")
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
    beam = 5

    def make_step(use_graph=True):
        if args.mode == "train":
            from ick_amd.training import TrainStep
            # encoder=: the step takes the feature map itself; Encoder.conv1 runs inside the captured step and writes
            # the image rows straight into the decoder's memory buffer.  All-reduces its bucket when world > 1.
            ts = TrainStep(dec, lr=4e-4, grad_clip=5.0, seed=rank, use_graph=use_graph, encoder=enc)
            live = (batch["captions"], feats, batch["caption_masks"], batch["caption_lengths"], batch["entities"], *extra)
            state = {"args": live}

            legacy = os.environ.get("ICK_BENCH_SEPARATE_ENCODER") == "1"   # A/B: eager Encoder + per-step input copies

            def step():
                if legacy:
                    with torch.no_grad():
                        e = enc(feats)
                    return ts(batch["captions"], e, batch["caption_masks"], batch["caption_lengths"],
                              batch["entities"], *extra)
                out = ts(*state["args"])
                if state["args"] is live and ts.use_graph:
                    # from now on the (HBM-resident) batch lives in the step's own input buffers -- where a loader's
                    # host-to-device copy would put the next batch -- so no per-step device-to-device input copy
                    state["args"] = ts.input_buffers()
                return out
        elif args.mode == "forward":
            def step():
                with torch.no_grad():
                    e = enc(feats)
                    return dec(batch["captions"], e, batch["caption_masks"], batch["caption_lengths"],
                               batch["entities"], *extra)
        elif args.mode == "greedy":
            def step():
                e = enc(feats)
                return dec.predict(e, L, batch["entities"], *extra)
        else:
            def step():
                e = enc(feats)
                return dec.predict_beam(e, L, batch["entities"], *extra, beam_size=beam)
        return step

    step = make_step()
    units_per_step = B * L

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device="cuda", dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = tmax.item()

    # ---- per-kernel pass: the same step launched eagerly on one stream, every C-ABI launch bracketed by HIP events
    # (every rank runs it -- the train step holds a collective -- but only rank 0 records)
    by_kernel = []
    if not args.no_profile:
        dec.use_hip_graphs = False
        os.environ["ICK_GROUP_SAME_STREAM"] = "1"     # the captured step's launches (grouped weight gradients, staged
                                                       # packs, 8-wave context chains), one stream
        pstep = make_step(use_graph=False)
        pstep()
        fence()
        if rank == 0:
            prof.start()
        for _ in range(args.profile_steps):
            pstep()
        if rank == 0:
            by_kernel = prof.summarise(prof.stop(), args.profile_steps)
        fence()
        os.environ.pop("ICK_GROUP_SAME_STREAM", None)
        dec.use_hip_graphs = True

    if rank == 0:
        out = {
            "metric": "decode_steps_per_sec", "value": world * units_per_step * args.steps / dt,
            "unit": "decode-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "timed_region_s": dt, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %s variant, per-GPU batch %d x %d decode positions, 14x14x2048 features, "
                                   "K=%d knowledge rows%s, vocab %d" % (cfgname, variant, B, L, K,
                                                                         (", F=%d facts" % Fn) if Fn else "", V),
                       "mode": {"train": "train_step (Encoder.conv1 + forward + packed CE + backward + gradient "
                                         "all-reduce + clamp + Adam; dropout 0.5/0.5/0.1 as the reference's train.py "
                                         "builds it)",
                                "forward": "teacher_forced_forward (Encoder.conv1 + DecoderTransformer.forward)",
                                "greedy": "greedy_decode (Encoder.conv1 + predict, KV-cached)",
                                "beam": "beam_decode (Encoder.conv1 + predict_beam, beam %d, KV-cached)" % beam}[args.mode],
                       "global_batch": world * B,
                       "collective_backend": backend if args.mode == "train" else "none",
                       "parallelism": ("dp%d (one flat-bucket all-reduce per step)" if args.mode == "train"
                                       else "dp%d (independent shards)") % world},
        }
        # whole-pass view: measured decode-steps/s against the bound of the pass's algorithmic work
        # at the same peaks -- teacher-forced passes are MFMA-bound (a train step is counted as 3x the forward FLOPs;
        # the step EXECUTES less: no conv1 backward, no data gradient into the image rows), the KV-cached greedy
        # decode is HBM-bound
        bounds = {("cfg2", "forward"): 3.96e6, ("cfg2", "train"): 3.96e6 / 3, ("cfg4", "forward"): 1.99e6,
                  ("cfg4", "train"): 1.99e6 / 3, ("cfg5", "greedy"): 3.39e6}
        executed_gflop = {("cfg2", "train"): 108.0, ("cfg2", "forward"): 50.85, ("cfg4", "forward"): 101.3}
        roof = {}
        traffic = {}
        if os.path.exists(TRAFFIC_TABLE):
            traffic = json.load(open(TRAFFIC_TABLE))
        for r in by_kernel:
            # HBM bytes per launch from the PMC counters (valid for the workload they were collected on: cfg2 / cfg5)
            if r["name"] in traffic and cfgname in ("cfg2", "cfg5") and args.mode != "beam":
                r["traffic"] = traffic[r["name"]]
        if by_kernel:
            dom = by_kernel[0]
            roof = {"bound": dom.get("bound", "latency"), "achieved": dom.get("achieved"), "peak": dom.get("peak"),
                    "unit": dom.get("unit"), "frac": dom.get("frac"), "kernel": dom["name"],
                    "kernel_avg_us": dom["avg_us"], "launches_per_step": dom["launches_per_step"],
                    "share_of_kernel_time": dom["us_per_step"] / max(1e-9, sum(r["us_per_step"] for r in by_kernel)),
                    "traffic": dom.get("traffic"),
                    "sum_kernel_us_per_step": sum(r["us_per_step"] for r in by_kernel),
                    "by_kernel": by_kernel,
                    "how": "HIP events on the launch stream around every C-ABI launch of %d eager single-stream steps "
                           "after the timed region; work = algorithmic FLOP (fp32 MFMA peak %.1f TFLOP/s) or bytes "
                           "(HBM peak %.0f GB/s)" % (args.profile_steps, PEAK_FP32_MFMA_TFLOPS, PEAK_HBM_GBS)}
        bnd = bounds.get((cfgname, args.mode))
        if bnd is not None:
            roof["pass_bound_steps_per_s"] = bnd * world
            roof["pass_frac"] = out["value"] / (bnd * world)
        ex = executed_gflop.get((cfgname, args.mode))
        if ex is not None:
            roof["pass_frac_executed"] = ex * 1e9 * world / (dt / args.steps) / (PEAK_FP32_MFMA_TFLOPS * 1e12 * world)
        out["roofline"] = roof
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(cfg, seed, args.cpu_seconds,
                                               {"train": "train", "greedy": "greedy", "beam": "greedy"}.get(args.mode, "forward"))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
