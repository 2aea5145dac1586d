"""Deterministic training mode (ICK_DETERMINISTIC=1 / TrainStep(deterministic=True)).

The default step lets float atomics order the terms of its split-K weight gradients, bias / LayerNorm column sums and
embedding scatter-adds, so two runs differ by ~1e-7 in the gradients and Adam(eps = 1e-8) turns noise-level entries
into +-lr steps.  In deterministic mode every such reduction has a fixed order (csrc/backward.hip), GEMMs are not
split, the step stays on one stream:
  * two runs of the same steps (dropout ON) end with bit-identical parameters, for the three variants;
  * its gradients equal the default path's within rounding (the default path is pinned to the reference's goldens);
  * checkpoint -> resume continues bit-identically to the uninterrupted run (the reference's CPU path is deterministic:
    geo-aware/train.py:105-129,282-292);
  * with two ranks (gloo, sharing this box's GPU): the run is bit-reproducible, the split (overlapped) all-reduce
    schedule gives the same bits as the single one, and the result equals the one-rank full-batch step within 2e-6
    (bitwise equality with ONE rank is not attainable: the full batch sums its 2 x B samples in another association).
"""
import os
import socket
import subprocess
import sys

import pytest
import torch

import ick_amd
import ick_amd.synth as synth
from test_forward_gpu import build_decoder

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def det_off_after():
    import ick_amd.ops as ops
    yield
    ops.set_deterministic(False)


def run_steps(variant, P, batches, enc_outs, deterministic, nsteps, V, seed=11, use_graph=True):
    from ick_amd.training import TrainStep
    dec = build_decoder(variant, V, P).train()                  # dropout on, as the reference trains
    ts = TrainStep(dec, lr=4e-4, grad_clip=5.0, seed=seed, deterministic=deterministic, use_graph=use_graph)
    losses = []
    for i in range(nsteps):
        b = batches[i % len(batches)]
        args = [b["captions"].cuda(), enc_outs[i % len(batches)].cuda(), b["caption_masks"].cuda(),
                b["caption_lengths"].cuda(), b["entities"]] + ([b["facts"].cuda()] if "facts" in b else [])
        losses.append(ts(*args).item())
    return ts.flat_p.clone(), ts.flat_g.clone(), losses


@pytest.mark.parametrize("variant,B,L,K,V,Fn", [("geo", 8, 12, 7, 300, 0), ("knowledge", 6, 10, 6, 200, 9),
                                                 ("news", 5, 9, 6, 150, 8)])
def test_deterministic_steps_are_bit_reproducible(variant, B, L, K, V, Fn, det_off_after):
    P = synth.make_params(variant, V, 3)
    batches = [synth.make_batch(variant, B, L, K, V, Fn, s) for s in (1, 2)]
    for b in batches:                                           # repeated tokens / entities: the scatter-adds collide
        b["captions"][:, 2] = b["captions"][0, 2]
    enc_outs = [synth.make_enc_out(B, s) for s in (1, 2)]
    runs = [run_steps(variant, P, batches, enc_outs, True, 3, V) for _ in range(2)]
    (p0, g0, l0), (p1, g1, l1) = runs
    assert l0 == l1
    assert torch.equal(g0, g1), (g0 - g1).abs().max().item()
    assert torch.equal(p0, p1), ((p0 - p1) != 0).sum().item()
    # eager launches give the same bits as the captured graphs
    p2, g2, l2 = run_steps(variant, P, batches, enc_outs, True, 3, V, use_graph=False)
    assert torch.equal(p0, p2) and l0 == l2


@pytest.mark.parametrize("variant,B,L,K,V,Fn", [("geo", 8, 12, 7, 300, 0), ("knowledge", 6, 10, 6, 200, 9),
                                                 ("news", 5, 9, 6, 150, 8)])
def test_deterministic_gradients_match_default_path(variant, B, L, K, V, Fn, det_off_after):
    """One step, dropout on (same counter-based masks): the gradient Adam consumed, default vs deterministic."""
    P = synth.make_params(variant, V, 4)
    batches = [synth.make_batch(variant, B, L, K, V, Fn, 5)]
    batches[0]["captions"][:, 3] = batches[0]["captions"][0, 3]
    enc_outs = [synth.make_enc_out(B, 5)]
    _, g_det, l_det = run_steps(variant, P, batches, enc_outs, True, 1, V)
    import ick_amd.ops as ops
    ops.set_deterministic(False)
    _, g_def, l_def = run_steps(variant, P, batches, enc_outs, False, 1, V)
    assert abs(l_det[0] - l_def[0]) < 1e-5
    scale = g_def[:-2].abs().max().item()
    assert (g_det[:-2] - g_def[:-2]).abs().max().item() < 2e-6 * max(1.0, scale) + 2e-7
    assert torch.equal(g_det[-2:], g_def[-2:])                  # [sum of token losses, token count]


def _weights(dec):
    return torch.cat([p.detach().reshape(-1).cpu() for p in dec.parameters()])


def test_resume_is_bitwise_in_deterministic_mode(tmp_path, monkeypatch, det_off_after):
    """train.main: two epochs in one go == one epoch, checkpoint, resume for the second -- bit for bit (Adam moments,
    step counter = position of the dropout stream, lr, batch order all come back from the checkpoint)."""
    from ick_amd import train as tr, utils as ut
    monkeypatch.setenv("ICK_DETERMINISTIC", "1")
    data_dir = str(tmp_path / "data")
    synth.write_dataset(data_dir, "toy", "knowledge", n_train=24, n_val=8, n_test=4, L=12, K=6, V=60, F=5)
    base = dict(variant="knowledge", data_dir=data_dir, data_name="toy", batch_size=8, workers=0, print_freq=1000,
                fused=True, seed=3)
    out_a, out_b = tmp_path / "a", tmp_path / "b"
    os.makedirs(out_a), os.makedirs(out_b)
    torch.manual_seed(0)
    tr.main(tr.Config(epochs=2, out_dir=str(out_a), **base))
    torch.manual_seed(0)
    tr.main(tr.Config(epochs=1, out_dir=str(out_b), **base))
    tr.main(tr.Config(epochs=2, out_dir=str(out_b), checkpoint=str(out_b / "checkpoint_0_toy.pth.tar"), **base))
    wa = _weights(ut.load_checkpoint(str(out_a / "checkpoint_toy.pth.tar"), map_location="cuda")["decoder"])
    wb = _weights(ut.load_checkpoint(str(out_b / "checkpoint_toy.pth.tar"), map_location="cuda")["decoder"])
    assert torch.equal(wa, wb), ((wa - wb) != 0).sum().item()
    # and a second uninterrupted run reproduces the first
    out_c = tmp_path / "c"
    os.makedirs(out_c)
    torch.manual_seed(0)
    tr.main(tr.Config(epochs=2, out_dir=str(out_c), **base))
    wc = _weights(ut.load_checkpoint(str(out_c / "checkpoint_toy.pth.tar"), map_location="cuda")["decoder"])
    assert torch.equal(wa, wc)


def _two_rank_run(tmp, tag, env_extra):
    data_dir = os.path.join(tmp, "data")
    if not os.path.exists(data_dir):
        synth.write_dataset(data_dir, "toy", "knowledge", n_train=24, n_val=8, n_test=4, L=12, K=6, V=60, F=5)
    out = os.path.join(tmp, tag)
    os.makedirs(out)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, ICK_DP_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", ICK_DETERMINISTIC="1", **env_extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dp_train_worker.py"), data_dir, out]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    f0, f1 = torch.load(os.path.join(out, "flat0.pt")), torch.load(os.path.join(out, "flat1.pt"))
    assert torch.equal(f0, f1)
    return f0


def test_two_ranks_deterministic_and_overlapped_allreduce_same_bits(tmp_path):
    """train.main with two ranks in deterministic mode: run twice -> the same bits; with the split schedule (the early
    half of the gradient bucket travels while the late half is computed, ICK_SPLIT_ALLREDUCE=1) -> still the same bits."""
    a = _two_rank_run(str(tmp_path), "serial_a", {"ICK_SPLIT_ALLREDUCE": "0"})
    b = _two_rank_run(str(tmp_path), "serial_b", {"ICK_SPLIT_ALLREDUCE": "0"})
    c = _two_rank_run(str(tmp_path), "split", {"ICK_SPLIT_ALLREDUCE": "1"})
    assert torch.equal(a, b), ((a - b) != 0).sum().item()
    assert torch.equal(a, c), ((a - c) != 0).sum().item()


def test_two_shards_equal_full_batch_in_deterministic_mode(det_off_after):
    """The data-parallel identity at the bucket level: the sum of two ranks' unnormalised gradient buckets (3 samples
    each) against the one-rank bucket of all 6 samples.  Deterministic mode removes the run-to-run noise, what remains is
    the different association of the sample sums: <= 2e-6 of the largest gradient."""
    import ick_amd.ops as ops
    from ick_amd.training import TrainStep
    from test_training_gpu import zero_dropout
    variant, B, L, K, V, Fn, seed = "knowledge", 6, 9, 5, 120, 4, 7
    P = synth.make_params(variant, V, seed)
    batch = synth.make_batch(variant, B, L, K, V, Fn, seed)
    enc_out = synth.make_enc_out(B, seed)

    def bucket(lo, hi):
        dec = zero_dropout(build_decoder(variant, V, P).train())
        ts = TrainStep(dec, lr=0.0, grad_clip=0.0, deterministic=True, use_graph=False)
        ts._part_a(batch["captions"][lo:hi].cuda(), batch["caption_masks"][lo:hi].cuda(),
                   batch["entities"][lo:hi].cuda().float(), batch["facts"][lo:hi].cuda(),
                   dec._token_major(enc_out[lo:hi].cuda()), None, batch["caption_lengths"][lo:hi].cuda())
        return ts.flat_g.clone(), ts.n

    (g0, n), (g1, _), (gf, _) = bucket(0, 3), bucket(3, 6), bucket(0, 6)
    (g0b, _) = bucket(0, 3)
    assert torch.equal(g0, g0b)                                  # bit-reproducible
    summed = g0 + g1
    assert summed[n + 1].item() == gf[n + 1].item()              # token counts add up exactly
    err = (summed[:n] - gf[:n]).abs().max().item()
    assert err < 2e-6 * max(1.0, gf[:n].abs().max().item()), err
    assert ops.is_deterministic()
