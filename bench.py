"""bench.py -- decode-steps/s of the caption-decoder hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W] [--mode train|forward|greedy|beam] [--config cfg2] [--no-modes]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
(`python bench.py --gpus N` with N > 1 and no WORLD_SIZE starts that torch.distributed.run line itself, as a child
process, and relays rank 0's JSON line; `ranks_seen` in the line is what the probe all-reduce returned.)

One *step* = one pass of the hot path over one batch of synthetic input resident in HBM, on cfg2 =
B 64 x L 20 decode positions, 14x14x2048 features, K=20 knowledge rows, vocab 10k => 1280 decode-steps
per step and per GPU (weak scaling: every rank owns its own 64-sample shard):
  --mode train (default)  Encoder.conv1 + DecoderTransformer forward + packed cross entropy + backward +
                          ONE all-reduce of the flat gradient bucket (RCCL) + clamp + Adam: the body of
                          train.py's loop (geo-aware/train.py:269-292), dropout on as the reference has it
  --mode forward          Encoder.conv1 + teacher-forced forward only (validate(), no collective)
  --mode greedy           Encoder.conv1 + predict() greedy decode, KV-cached (eval.py path; cfg5)
  --mode beam             Encoder.conv1 + predict_beam() (beam 5, batch 32: north_star cfg5; parity-unpinned)
Timing: W warm-up steps, then blocks of EXACTLY K steps, each bracketed by barrier + synchronize on both sides (max
over ranks); blocks are repeated until 2 s of timed device work and the MEDIAN block is reported (`repeats`).
Rank 0 prints ONE JSON line with the whole-job rate plus
  roofline      by_kernel: every kernel class of the step (launches per step, mean duration from HIP events on
                the launch stream in an eager single-stream pass, algorithmic FLOP or bytes, fraction of the fp32
                MFMA / HBM peak); the headline kernel/achieved/frac describe the class with the LARGEST share of
                the step's kernel time; pass_frac_executed = the whole step's EXECUTED FLOPs (or algorithmic bytes)
                per second against the peak
  modes         (one GPU, unless --no-modes) the other workloads of SURVEY.md §8(d) in the same run: forward cfg2,
                greedy cfg5, beam cfg5, train cfg4 -- ms_per_step, decode-steps/s, pass_frac, dominant kernel class
  cpu_baseline  the same workload on the host cores through oracle/stock.py (a port of the reference's
                PyTorch-CPU path): all usable cores and a 1-thread leg, each bounded to a few seconds.
"""
import argparse
import gc
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Tables made from profiler runs of THIS command (tools/r5_profiles.sh), one pair per (mode, config):
#   traffic  PMC bytes per launch and kernel class (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE; tools/traffic_table.py)
#   in_step  kernel time per class INSIDE the captured step (rocprofv3 kernel trace; tools/in_step_table.py): the eager
#            single-stream pass below times every kernel alone, the captured step runs two streams beside each other
# Each records the build id (ick_amd.build.source_id) it was made with; a table made from another tree is still attached
# but the line says so (`in_step_stale` / `traffic_stale`).
def _table(kind, mode, cfgname):
    return os.path.join(ROOT, "profiles", "r05_%s_%s_%s.json" % (kind, mode, cfgname))
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, 64 FLOP/clk/SIMD
PEAK_HBM_GBS = 8000.0
BEAM = 5

# FLOPs one step EXECUTES (SURVEY.md §8(d) closed forms; train = 3 x forward minus what the step skips: Encoder.conv1's
# backward, which the reference computes but never uses with fine_tune_encoder=False, and the image rows' data gradient)
EXECUTED_GFLOP = {("cfg2", "forward"): 50.85, ("cfg2", "train"): 3 * 50.85 - 2 * 15.41 - 13.5,
                  ("cfg4", "forward"): 101.3, ("cfg4", "train"): 3 * 101.3 - 2 * 15.41 - 13.5}
# algorithmic HBM bytes per decoded token step of the whole batch (cross + self K/V, per-step weights,
# logits); beam 5: the hypotheses of a caption share its cross K/V, 160 rows of logits
ALGO_MB_PER_TOKEN = {("cfg5", "greedy"): 75.6, ("cfg5", "beam"): 53.1 + 22.3 + 160 * 10020 * 4 / 1e6}


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
    ap.add_argument("--no-modes", action="store_true", help="headline only: skip the forward / greedy / beam / cfg4 legs")
    ap.add_argument("--min-seconds", type=float, default=2.0, help="repeat the K-step block until this much timed work")
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


class Workload:
    """One (mode, config) on this rank's GPU: synthetic batch resident in HBM, random-init weights of the reference's
    architecture (same seed on every rank), the step function and its eager twin for the per-kernel pass."""

    def __init__(self, mode, cfgname, rank):
        import ick_amd
        import ick_amd.synth as synth
        self.mode, self.cfgname, self.rank = mode, cfgname, rank
        self.cfg = cfg = dict(synth.CONFIGS[cfgname])
        variant, B, L, K, V, Fn = cfg["variant"], cfg["B"], cfg["L"], cfg["K"], cfg["V"], cfg["F"]
        self.seed = seed = 100 + rank            # every rank owns a different shard of the global batch
        m = ick_amd.load_models(variant)
        dec = m.DecoderTransformer(synth.make_word_map(V), 300, 512, 512, 10, 3)
        dec.load_state_dict(synth.make_params(variant, V, 0), strict=False)      # same weights on every rank
        dec = dec.cuda()
        self.dec = dec.train() if mode == "train" else dec.eval()
        enc = m.Encoder(emb_dim=300)
        cw, cb = synth.make_conv1(0)
        with torch.no_grad():
            enc.conv1.weight.copy_(cw)
            enc.conv1.bias.copy_(cb)
        self.enc = enc.cuda().eval()
        self.batch = {k: v.cuda() for k, v in synth.make_batch(variant, B, L, K, V, Fn, seed).items()}
        self.feats = synth.make_feats(B, seed).cuda()
        self.extra = [self.batch["facts"]] if variant != "geo" else []
        self.units_per_step = B * L
        self.train_step = None

    def describe(self):
        c = self.cfg
        return "%s: %s variant, per-GPU batch %d x %d decode positions, 14x14x2048 features, K=%d knowledge rows%s, vocab %d" % (
            self.cfgname, c["variant"], c["B"], c["L"], c["K"], (", F=%d facts" % c["F"]) if c["F"] else "", c["V"])

    def make_step(self, use_graph=True, adopt_buffers=True):
        mode, dec, enc, batch, feats, extra, L = self.mode, self.dec, self.enc, self.batch, self.feats, self.extra, self.cfg["L"]
        if mode == "train":
            from ick_amd.training import TrainStep
            # encoder=: the step takes the feature map itself; Encoder.conv1 runs inside the captured step and writes
            # the image rows straight into the decoder's memory buffer.  All-reduces its bucket when world > 1.
            if use_graph:
                os.environ.setdefault("ICK_ALLREDUCE_PROBE", "1")     # several ranks: time one all-reduce of the bucket
            # lazy_update: step i's clamp + Adam runs at the head of step i + 1's graph beside Encoder.conv1 (as
            # `python -m ick_amd.train` runs it); run_workload() flushes in front of every timed block and again INSIDE it
            # behind its last step, so a block of K steps holds exactly K forward / backward passes and K optimizer updates
            ts = TrainStep(dec, lr=4e-4, grad_clip=5.0, seed=self.rank, use_graph=use_graph, encoder=enc,
                           lazy_update=use_graph and os.environ.get("ICK_BENCH_EAGER_UPDATE") != "1")
            if use_graph:
                self.train_step = ts
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
        else:
            # attach_encoder: the calls below take the feature map itself and Encoder.conv1 runs inside the captured
            # graph beside the context-encoder chain (ICK_BENCH_SEPARATE_ENCODER=1: the stand-alone Encoder launch in
            # front of the graph, as before round 4).  After the first call the (HBM-resident) batch lives in the graph's
            # own input buffers -- where a loader's host-to-device copy would put the next batch -- so no step pays a
            # device-to-device copy of its inputs.
            legacy = os.environ.get("ICK_BENCH_SEPARATE_ENCODER") == "1" or not use_graph
            if not legacy:
                dec.attach_encoder(enc)
            cur = {"feats": feats, "ent": batch["entities"], "facts": extra[0] if extra else None,
                   "caps": batch["captions"], "masks": batch["caption_masks"]}

            def adopt():
                bufs = dec.input_buffers() if (use_graph and adopt_buffers) else None
                if bufs is None or cur.get("adopted"):
                    return
                if mode == "forward":
                    cur["caps"], cur["masks"], cur["ent"], cur["facts"], img = bufs
                else:
                    img, cur["ent"], cur["facts"] = bufs
                if not legacy:
                    cur["feats"] = img
                cur["adopted"] = True

            def image():
                return enc(cur["feats"]) if legacy else cur["feats"]

            def fx():
                return [cur["facts"]] if extra else []

            if mode == "forward":
                def step():
                    with torch.no_grad():
                        out = dec(cur["caps"], image(), cur["masks"], batch["caption_lengths"], cur["ent"], *fx())
                    adopt()
                    return out
            elif mode == "greedy":
                def step():
                    out = dec.predict(image(), L, cur["ent"], *fx())
                    adopt()
                    return out
            else:
                def step():
                    out = dec.predict_beam(image(), L, cur["ent"], *fx(), beam_size=BEAM)
                    adopt()
                    return out
        return step

    def graph_in_use(self):
        """Did the timed steps replay captured hipGraphs (True) or fall back to eager launches (False)?"""
        if self.mode == "train":
            return bool(self.train_step is not None and self.train_step.use_graph and self.train_step._graphs)
        return bool(self.dec.use_hip_graphs and self.dec.__dict__.get("_graphs"))


def run_workload(wl, steps, warmup, min_seconds, world, profile_steps, max_repeats=400):
    """-> dict(ms_per_step, value, repeats, timed_region_s, graph, by_kernel)."""
    import torch.distributed as dist
    import ick_amd.profiling as prof

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def flush():
        if wl.train_step is not None:
            wl.train_step.flush()          # (lazy_update) the last step's optimizer update

    def block(step):
        flush()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        flush()
        fence()
        dt = torch.tensor([time.perf_counter() - t0], device="cuda", dtype=torch.float64)
        if world > 1:
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        return dt.item()

    step = wl.make_step()
    for _ in range(warmup):
        step()
    times = [block(step)]
    # every rank derives the same repeat count from the (max-reduced) first block
    repeats = int(min(max_repeats, max(1, -(-min_seconds // max(times[0], 1e-6)))))
    for _ in range(repeats - 1):
        times.append(block(step))
    dt = statistics.median(times)
    graph = wl.graph_in_use()
    # The timed steps above find their batch in the graph's own input buffers (where a loader's host-to-device copy puts
    # it: eval.py / train.py do).  A caller that hands over OTHER device tensors pays a device-to-device copy of the
    # inputs in front of every replay (103 MB of features at B = 64): that number too, one block, not the headline.
    dt_copy = None
    if wl.mode != "train":
        step_copy = wl.make_step(adopt_buffers=False)
        step_copy()
        dt_copy = block(step_copy)

    # ---- per-kernel pass: the same step launched eagerly on one stream, every C-ABI launch bracketed by HIP events
    # (every rank runs it -- the train step holds a collective -- but only rank 0 records)
    by_kernel = []
    if profile_steps > 0:
        wl.dec.use_hip_graphs = False
        os.environ["ICK_GROUP_SAME_STREAM"] = "1"     # the captured step's launches (grouped weight gradients, staged
                                                       # packs, 8-wave context chains), one stream
        pstep = wl.make_step(use_graph=False)
        pstep()
        fence()
        if wl.rank == 0:
            prof.start()
        for _ in range(profile_steps):
            pstep()
        if wl.rank == 0:
            by_kernel = prof.summarise(prof.stop(), profile_steps)
        fence()
        os.environ.pop("ICK_GROUP_SAME_STREAM", None)
        wl.dec.use_hip_graphs = True
    return {"ms_per_step": dt / steps * 1e3, "value": world * wl.units_per_step * steps / dt, "repeats": len(times),
            "timed_region_s": sum(times), "block_s_min_median_max": [min(times), dt, max(times)], "graph": graph,
            "by_kernel": by_kernel, "ms_per_step_with_input_copy": None if dt_copy is None else dt_copy / steps * 1e3}


def pass_fraction(cfgname, mode, ms_per_step, L):
    """The whole step against the peak for the work it executes: FLOPs for the teacher-forced passes (fp32 MFMA
    peak), algorithmic bytes per token step for the KV-cached decodes (HBM peak).  None when no closed form is kept."""
    ex = EXECUTED_GFLOP.get((cfgname, mode))
    if ex is not None:
        return {"bound": "mfma", "frac": ex * 1e9 / (ms_per_step * 1e-3) / (PEAK_FP32_MFMA_TFLOPS * 1e12),
                "executed_gflop_per_step": ex}
    mb = ALGO_MB_PER_TOKEN.get((cfgname, mode))
    if mb is not None:
        # one bench step = conv1 + prefill + L token steps; the bound counts the token steps only
        return {"bound": "hbm", "frac": mb * 1e6 * L / (ms_per_step * 1e-3) / (PEAK_HBM_GBS * 1e9),
                "algorithmic_mb_per_token_step": mb}
    return None


def _gemm_mode():
    from ick_amd import ops
    return ops.gemm_split_mode()


def self_launch(args):
    """`python bench.py --gpus N` without torchrun: start `python -m torch.distributed.run --nproc-per-node N bench.py ...`
    as a CHILD process and relay rank 0's JSON line.  The parent makes no GPU call before this (importing torch does
    not initialise HIP) and never replaces itself: an exec from a process that has touched the GPU takes the node down."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL's intra-node transport needs it here
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    for ln in r.stdout.splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    raise SystemExit(r.returncode if r.returncode else (0 if lines else 1))


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch one rank per GPU" % (args.gpus, world))
    import torch.distributed as dist
    torch.cuda.set_device(local_rank % max(1, torch.cuda.device_count()))
    backend = "none"
    ranks_seen = 1
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
        ranks_seen = int(probe.item())
        assert ranks_seen == world, "all-reduce probe returned %s for %d ranks" % (probe.item(), world)
        backend = dist.get_backend()

    cfgname = args.config or ("cfg5" if args.mode in ("greedy", "beam") else "cfg2")
    wl = Workload(args.mode, cfgname, rank)
    res = run_workload(wl, args.steps, args.warmup, args.min_seconds, world, 0 if args.no_profile else args.profile_steps)
    cfg, seed = wl.cfg, wl.seed
    B, L = cfg["B"], cfg["L"]

    if rank == 0:
        out = {
            "metric": "decode_steps_per_sec", "value": res["value"],
            "unit": "decode-steps/s", "n_gpus": world, "ranks_seen": ranks_seen, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": res["ms_per_step"], "repeats": res["repeats"], "timed_region_s": res["timed_region_s"],
            "block_s_min_median_max": res["block_s_min_median_max"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            # arithmetic type: fp32 in, fp32 out, fp32 accumulation everywhere; with the split product mode the large GEMM
            # tiles form each fp32 product from bf16 partial products of the exact 3-way split (config.gemm_products)
            "dtype": "f32" if _gemm_mode() == 0 else "f32 (3xbf16 split products, fp32 accumulate)",
            "data": "synthetic",
            "config": {"workload": wl.describe(),
                       "mode": {"train": "train_step (Encoder.conv1 + forward + packed CE + backward + gradient "
                                         "all-reduce + clamp + Adam; dropout 0.5/0.5/0.1 as the reference's train.py "
                                         "builds it)",
                                "forward": "teacher_forced_forward (Encoder.conv1 + DecoderTransformer.forward)",
                                "greedy": "greedy_decode (Encoder.conv1 + predict, KV-cached)",
                                "beam": "beam_decode (Encoder.conv1 + predict_beam, beam %d, KV-cached)" % BEAM}[args.mode],
                       "global_batch": world * B,
                       "graph": res["graph"],      # False = hipGraph capture failed and the steps were launched eagerly
                       # forward / greedy / beam: the same step when the inputs are NOT already in the graph's input buffers
                       # (one device-to-device copy of the feature map per call in front of the replay)
                       "ms_per_step_with_input_copy": res["ms_per_step_with_input_copy"],
                       # how the large GEMM tiles form their fp32 products (csrc/gemm.hip; ICK_GEMM_SPLIT)
                       "gemm_products": {0: "exact fp32 MFMA (v_mfma_f32_16x16x4_f32)",
                                         1: "split: 6 bf16 MFMA partial products of the exact 3-way bf16 split, fp32 "
                                            "accumulate (forward layouts)",
                                         2: "split: 6 bf16 MFMA partial products of the exact 3-way bf16 split, fp32 "
                                            "accumulate (all large tiles)"}[_gemm_mode()],
                       "collective_backend": backend if args.mode == "train" else "none",
                       # several ranks: milliseconds one all-reduce of the flat gradient bucket took at construction (median
                       # of 3, max over ranks) and whether the step overlaps half of it with the backward pass
                       "allreduce_probe_ms": getattr(wl.train_step, "allreduce_probe_ms", None),
                       "split_allreduce": bool(getattr(wl.train_step, "split", False)),
                       "parallelism": ("dp%d (one flat-bucket all-reduce per step)" if args.mode == "train"
                                       else "dp%d (independent shards)") % world},
        }
        by_kernel = res["by_kernel"]
        roof = {}
        import ick_amd.build as _build
        build_id = _build.source_id()
        traffic = in_step = {}
        if _gemm_mode() == 1 and os.path.exists(_table("traffic", args.mode, cfgname)):
            traffic = json.load(open(_table("traffic", args.mode, cfgname)))
        for r in by_kernel:
            # HBM bytes per launch from the PMC counters of this workload
            if r["name"] in traffic:
                r["traffic"] = traffic[r["name"]]
                # the decode step's bytes are per token (all of its launches), every other class's per launch
                r["traffic_per"] = "token" if r["name"].startswith("fused decode step") else "launch"
        if _gemm_mode() == 1 and os.path.exists(_table("in_step", args.mode, cfgname)):
            in_step = json.load(open(_table("in_step", args.mode, cfgname)))
        for r in by_kernel:
            t = in_step.get(r["name"])
            if t and r.get("peak") and r.get("work_per_launch"):
                # the same algorithmic work over the class's kernel time inside the captured step
                r["in_step_us_per_step"] = t["us_per_step"]
                ach = r["work_per_launch"] * r["launches_per_step"] / (t["us_per_step"] * 1e-6) / (1e12 if r["unit"] == "TFLOP/s" else 1e9)
                r["achieved_in_step"] = ach
                r["frac_in_step"] = ach / r["peak"]
        if by_kernel:
            dom = by_kernel[0]
            roof = {"bound": dom.get("bound", "latency"), "achieved": dom.get("achieved"), "peak": dom.get("peak"),
                    "unit": dom.get("unit"), "frac": dom.get("frac"), "kernel": dom["name"],
                    "kernel_avg_us": dom["avg_us"], "launches_per_step": dom["launches_per_step"],
                    "share_of_kernel_time": dom["us_per_step"] / max(1e-9, sum(r["us_per_step"] for r in by_kernel)),
                    "traffic": dom.get("traffic"), "traffic_per": dom.get("traffic_per"),
                    # frac: the class timed alone (eager, one stream); frac_in_step: inside the captured step, beside the
                    # other stream's kernels (profiles/r05_in_step_<mode>_<config>.json, rocprofv3 of this command)
                    "frac_in_step": dom.get("frac_in_step"), "in_step_source": in_step.get("_source"),
                    # the tables above were made by profiler runs of a tree with this id; stale = not the tree running now
                    "build_id": build_id,
                    "in_step_stale": (in_step.get("_build_id") != build_id) if in_step else None,
                    "traffic_stale": (traffic.get("_build_id") != build_id) if traffic else None,
                    "sum_kernel_us_per_step": sum(r["us_per_step"] for r in by_kernel),
                    "by_kernel": by_kernel,
                    "how": "HIP events on the launch stream around every C-ABI launch of %d eager single-stream steps "
                           "after the timed region; work = algorithmic FLOP (fp32 MFMA peak %.1f TFLOP/s; kernels whose "
                           "products are six bf16 MFMAs: bf16 dense peak 2500 / 6 = %.1f TFLOP/s fp32-equivalent) or bytes "
                           "(HBM peak %.0f GB/s)" % (args.profile_steps, PEAK_FP32_MFMA_TFLOPS, 2500.0 / 6, PEAK_HBM_GBS)}
        if by_kernel:
            # sum over the kernel classes of (algorithmic work / that class's own peak): what the step would take if
            # every kernel ran at its roofline, back to back -- the whole step's fraction against a peak that mixes the
            # exact-fp32 pipe, the bf16 pipe of the split-product kernels and HBM
            # (row chains: their own floor -- the weight bytes each workgroup streams through its CU -- not the matrix pipe's)
            floor_us = sum(r["cu_stream_floor_us"] * r["launches_per_step"] if r.get("cu_stream_floor_us") else
                           r["work_per_launch"] * r["launches_per_step"] / (r["peak"] * (1e12 if r["unit"] == "TFLOP/s" else 1e9)) * 1e6
                           for r in by_kernel if r.get("peak"))
            roof["pass_floor_us"] = floor_us
            roof["pass_frac_of_floor"] = floor_us / (res["ms_per_step"] * 1e3)
        pf = pass_fraction(cfgname, args.mode, res["ms_per_step"], L)
        if pf is not None:
            roof["pass_frac_executed"] = pf["frac"]
            roof["pass_bound"] = pf["bound"]
            roof.update({k: v for k, v in pf.items() if k not in ("frac", "bound")})
        out["roofline"] = roof

    # ---- the other workloads of SURVEY.md 8(d), one GPU only (no collective inside; kept off for N > 1 so that the
    # scaling runs stay short).  Each runs in a child process of its own: several workloads in ONE process slow each
    # other down (measured: greedy 9.3 ms instead of 2.4 ms per decode behind a train workload -- the streams and
    # graphs of the earlier workloads stay with the HIP runtime), and a child's number is the number a user of that
    # mode alone would see.
    if world == 1 and not args.no_modes:
        import subprocess
        modes = {}
        del wl
        gc.collect()
        torch.cuda.empty_cache()
        torch.cuda.synchronize()
        # the *_exact_fp32 legs: the headline workloads with every product on the exact fp32 MFMA (ICK_GEMM_SPLIT=0), kept
        # beside the default (split-bf16 products on the large GEMM tiles; all modes are held to the same reference-made
        # goldens and tolerances by the parity suite, tests/conftest.py gemm_split)
        for name, mode, cname, steps, exact in (("forward_cfg2", "forward", "cfg2", 50, False),
                                                ("greedy_cfg5", "greedy", "cfg5", 20, False),
                                                ("beam5_cfg5", "beam", "cfg5", 10, False),
                                                ("train_cfg4", "train", "cfg4", 20, False),
                                                ("train_cfg2_exact_fp32", "train", "cfg2", 50, True),
                                                ("forward_cfg2_exact_fp32", "forward", "cfg2", 50, True)):
            if (mode, cname) == (args.mode, cfgname) and (not exact or _gemm_mode() == 0):
                continue
            # the CPU port beside the forward and greedy numbers as well (north_star: "in the same run"): bounded legs of
            # ~3 s each inside those children; the headline's own leg runs at the end of this process
            cpu_leg = name in ("forward_cfg2", "greedy_cfg5") and not args.no_cpu_baseline
            cmd = [sys.executable, os.path.abspath(__file__), "--mode", mode, "--config", cname, "--steps", str(steps),
                   "--warmup", "3", "--min-seconds", "0.5", "--no-modes", "--profile-steps",
                   "0" if args.no_profile else "2"] + (["--no-profile"] if args.no_profile else []) + \
                  (["--cpu-seconds", "3"] if cpu_leg else ["--no-cpu-baseline"])
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
            if exact:
                env["ICK_GEMM_SPLIT"] = "0"
            r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not lines:
                modes[name] = {"error": (r.stderr or r.stdout)[-400:]}
                continue
            c = json.loads(lines[-1])
            entry = {"workload": c["config"]["workload"], "mode": mode, "steps": steps, "repeats": c["repeats"],
                     "ms_per_step": c["ms_per_step"], "value": c["value"], "unit": c["unit"], "graph": c["config"]["graph"],
                     "ms_per_step_with_input_copy": c["config"].get("ms_per_step_with_input_copy"),
                     "gemm_products": c["config"].get("gemm_products"), "dtype": c.get("dtype")}
            rf = c.get("roofline", {})
            if "pass_frac_executed" in rf:
                entry["pass_frac"] = rf["pass_frac_executed"]
                entry["pass_bound"] = rf["pass_bound"]
            if "pass_frac_of_floor" in rf:
                entry["pass_frac_of_floor"] = rf["pass_frac_of_floor"]
            if rf.get("kernel"):
                entry["dominant_kernel"] = {"name": rf["kernel"], "avg_us": rf["kernel_avg_us"],
                                            "launches_per_step": rf["launches_per_step"], "frac": rf.get("frac"),
                                            "bound": rf.get("bound"), "share_of_kernel_time": rf["share_of_kernel_time"],
                                            "frac_in_step": rf.get("frac_in_step"),
                                            # PMC bytes per launch (decode step: per token); null where no counter run
                                            # of that workload is committed (profiles/r05_traffic_<mode>_<config>.json)
                                            "traffic": rf.get("traffic"), "traffic_per": rf.get("traffic_per"),
                                            "in_step_stale": rf.get("in_step_stale"), "traffic_stale": rf.get("traffic_stale")}
            if "cpu_baseline" in c:
                cb = c["cpu_baseline"]
                entry["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                                         "sample": cb["sample"], "one_thread": cb["one_thread"]["value"]}
            modes[name] = entry
        out["modes"] = modes

    if world > 1:
        # every rank leaves the job before rank 0 spends its seconds on the host baseline (no rank waits in a collective)
        torch.cuda.synchronize()
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, seed, args.cpu_seconds,
                                               {"train": "train", "greedy": "greedy", "beam": "greedy"}.get(args.mode, "forward"))
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
