"""Per-op Python wrappers over the C ABI.  Tensors are torch CUDA(ROCm) tensors used purely as
device memory + stream plumbing; all arithmetic happens in libick_amd.so."""
import contextlib
import ctypes as C
import gc
import math
import os

import torch

from . import lib as L


def _p(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _f32c(t, name):
    if t.dtype != torch.float32 or not t.is_cuda:
        raise L.IckError("%s must be a float32 device tensor (got %s on %s)" % (name, t.dtype, t.device))
    return t


# bench.py hook: when set to {"shape": (M, N, K), "events": []}, every ick_gemm launch of that shape is
# bracketed by HIP events recorded on the launch stream (torch's current stream).
TIMED = None


def _extent(t):
    """Floats addressable from t.data_ptr() inside t's storage (bounds the kernel's buffer descriptor)."""
    return (t.untyped_storage().nbytes() - t.storage_offset() * t.element_size()) // 4


def _drop(a, drop):
    """drop = (p, seed, site[, epoch_tensor]) or None -> fields of an args struct."""
    if drop is not None and drop[0] > 0.0:
        a.drop_p, a.drop_seed, a.drop_site = float(drop[0]), int(drop[1]) & 0xFFFFFFFF, int(drop[2]) & 0xFFFFFFFF
        a.drop_epoch = _p(drop[3]) if len(drop) > 3 else None


def _dargs(drop):
    if drop is not None and drop[0] > 0.0:
        return (float(drop[0]), int(drop[1]) & 0xFFFFFFFF, int(drop[2]) & 0xFFFFFFFF,
                _p(drop[3]) if len(drop) > 3 else None)
    return 0.0, 0, 0, None


def gemm_args(A, B, Cout, M, N, K, a_rs, a_ks, b_rs, b_ks, c_rs, bias=None, a_grp=0, a_gs=0, a_gmap=None,
              c_grp=0, c_gs=0, c_gmap=None, relu=False, accumulate=False, atomic=False, split_k=1, alpha=1.0,
              drop=None, colsum_a=None, gate=None, gate_scale=1.0, b_ps=None):
    """ick_gemm_args for C[m,n] = act(alpha * sum_k A(m,k) B(n,k) + bias[n]) with explicit element strides; A/B/Cout
    are tensors (only their data pointers are used -- the caller guarantees the strides stay in bounds).
    colsum_a (k-major A only): colsum_a[m] += sum_k A(m,k).  b_ps: the pre-split copy of the (N, K) matrix B
    (presplit_weights): large problems then run on the LDS-DMA kernel of csrc/gemm_ps.hip."""
    for t in (A, B, Cout):
        if t.dtype != torch.float32:
            raise L.IckError("ick_gemm operands must be float32, got %s" % t.dtype)
    a = L.GemmArgs()
    a.A, a.B, a.C, a.bias = _p(A), _p(B), _p(Cout), _p(bias)
    a.M, a.N, a.K = M, N, K
    a.a_rs, a.a_ks, a.a_grp, a.a_gs, a.a_gmap = a_rs, a_ks, a_grp, a_gs, _p(a_gmap)
    a.b_rs, a.b_ks = b_rs, b_ks
    a.c_rs, a.c_grp, a.c_gs, a.c_gmap = c_rs, c_grp, c_gs, _p(c_gmap)
    a.flags = (L.GEMM_RELU if relu else 0) | (L.GEMM_ACCUM if accumulate else 0) | (L.GEMM_ATOMIC if atomic else 0)
    a.split_k = split_k
    a.alpha = alpha
    a.a_extent, a.b_extent = _extent(A), _extent(B)
    a.colsum_a = _p(colsum_a)
    if gate is not None:
        a.gate, a.gate_rs, a.gate_scale = _p(gate), gate.stride(0), gate_scale
    if b_ps is not None:
        assert b_ps.numel() * b_ps.element_size() == presplit_bytes(N, K), "b_ps is not the pre-split copy of an (N, K) matrix"
        a.b_ps = _p(b_ps)
    _drop(a, drop)
    return a


DEFAULT_GEMM_SPLIT = 1      # the library's product mode when ICK_GEMM_SPLIT is not set (csrc/gemm.hip)


def presplit_bytes(N, K):
    n = L.i64()
    L.check(L.load_raw().ick_presplit_bytes(int(N), int(K), C.byref(n)), "ick_presplit_bytes")
    return int(n.value)


def presplit_weights(pairs):
    """[(w (N, K) 2-D view with one unit stride -- a weight or its transposed view, dst uint8 buffer of
    presplit_bytes(N, K))]: the exact three-way bf16 split of every matrix in the image ick_gemm's b_ps wants
    (include/ick_amd.h, ick_presplit_weights); up to 16 matrices per launch."""
    items = (L.PresplitItem * len(pairs))()
    for it, (src, dst) in zip(items, pairs):
        assert src.dim() == 2 and src.dtype == torch.float32 and (src.stride(1) == 1 or src.stride(0) == 1)
        assert dst.numel() * dst.element_size() == presplit_bytes(src.shape[0], src.shape[1])
        it.src, it.dst, it.N, it.K = _p(src), _p(dst), src.shape[0], src.shape[1]
        it.src_rs, it.src_cs = src.stride(0), src.stride(1)
    L.check(L.load().ick_presplit_weights(items, len(pairs), _stream()), "ick_presplit_weights")


def presplit_buffer(N, K, device):
    return torch.empty(presplit_bytes(N, K), device=device, dtype=torch.uint8)


def presplit_cached(holder, name, w2d, key):
    """Persistent pre-split copy of the 2-D weight view `w2d`, kept in holder.__dict__ and refreshed IN PLACE (captured
    graphs keep reading the same buffer) when `key` (parameter versions / data pointers) changes.  None when the
    library's product mode is the exact fp32 MFMA: ick_gemm then never looks at it."""
    if gemm_split_mode() == 0:
        return None
    cache = holder.__dict__.setdefault("_ps_cache", {})
    ent = cache.get(name)
    nbytes = presplit_bytes(w2d.shape[0], w2d.shape[1])
    if ent is None or ent[1].numel() != nbytes or ent[1].device != w2d.device:
        ent = cache[name] = [None, torch.empty(nbytes, device=w2d.device, dtype=torch.uint8)]
    if ent[0] != key:
        presplit_weights([(w2d, ent[1])])
        ent[0] = key
    return ent[1]


def colsum_problem(a2d, out, split_k=1):
    """ick_gemm_args of a pure column sum out[n] += sum_m a2d[m, n] (ICK_GEMM_COLSUM_ONLY): no kernel of its own,
    it rides in a grouped launch (gemm_grouped) with the weight-gradient GEMMs.  split_k: row slices (float atomics)."""
    rows, cols = a2d.shape
    a = L.GemmArgs()
    a.A, a.colsum_a = _p(a2d), _p(out)
    a.M, a.N, a.K = cols, 1, rows
    a.a_rs, a.a_ks = 1, a2d.stride(0)
    a.flags, a.split_k, a.alpha = L.GEMM_COLSUM_ONLY | (L.GEMM_ATOMIC if split_k > 1 else 0), split_k, 1.0
    a.a_extent = _extent(a2d)
    return a


def set_deterministic(on=True):
    """Deterministic mode of the library (ick_set_deterministic): scatter-adds and column sums of the backward pass in a
    fixed order instead of float atomics; every split below collapses to 1.  training.TrainStep(deterministic=True) and
    the autograd bridge also keep the step on one stream.  Slower, bit-reproducible run to run."""
    L.check(L.load().ick_set_deterministic(1 if on else 0), "ick_set_deterministic")


def set_gemm_split(mode):
    """Product mode of the large GEMM tiles (ick_set_gemm_split; ICK_GEMM_SPLIT in the environment; DEFAULT 1 since round
    4): 0 = every product on the exact fp32 MFMA; 1 = six bf16 x bf16 partial products of the exact three-way bf16 split
    of both fp32 operands, accumulated in fp32, where that is faster (B operand k-contiguous or pre-split); 2 = on every
    large-tile problem.  Non-finite operands (documented difference, tests/test_gemm_split_gpu.py): the residual planes
    of a +-inf element -- and of a finite one within 2^-8 of FLT_MAX, whose bf16 hi plane rounds to infinity -- are NaN
    (inf - inf), so its dot products are NaN in the split modes where the exact mode gives +-inf / a finite sum."""
    L.check(L.load().ick_set_gemm_split(int(mode)), "ick_set_gemm_split")


def gemm_split_mode():
    return int(L.load().ick_get_gemm_split())


def is_deterministic():
    return bool(L.load().ick_get_deterministic())


def wgrad_split(rows, n_out, k_in, grouped=False):
    """K split of a weight-gradient GEMM (reduction over `rows`): enough workgroups to fill the GPU (~1600), slices
    of at least 256 rows.  Measured (probe_ops): vocabulary 10000x300 over 1280 rows 121 us at 5 slices, 91 us at 2.
    grouped: the problem goes out in a layer's grouped launch (~1000 workgroups together): two slices are enough, more
    only add atomics and prologues (train step 2.01 -> 1.98 ms)."""
    if is_deterministic():
        return 1       # slices of one reduction meet in float atomics: their order would decide the rounding
    tiles = ((n_out + 63) // 64) * ((k_in + 63) // 64)
    cap = 2 if grouped else 16
    return max(1, min(cap, rows // 256, (1600 + tiles // 2) // tiles))


CONV1_TILE = (128, 64)   # workgroup tile ick_gemm picks for Encoder.conv1 at bench size on the exact fp32 MFMA


def conv1_tile():
    """The workgroup tile Encoder.conv1 runs on at bench size in the library's current product mode (asserted by the
    parity tests): 128 x 80 of the pre-split kernel (csrc/gemm_ps.hip; two workgroups per CU, small enough to share the
    CU with the other stream's row chains) with split products, else CONV1_TILE."""
    return (128, 80) if gemm_split_mode() >= 1 else CONV1_TILE

# test hook: a list that receives (M, N, K, plan dict) of every single-problem ick_gemm launch while it is set
PLAN_LOG = None


def gemm_plan(a):
    """Kernel configuration ick_gemm would choose for the ick_gemm_args `a` (ick_gemm_plan)."""
    info = L.GemmPlanInfo()
    L.check(L.load().ick_gemm_plan(C.byref(a), C.byref(info)), "ick_gemm_plan")
    return {n: getattr(info, n) for n, _ in L.GemmPlanInfo._fields_}


def _log_plan(a):
    if PLAN_LOG is not None:
        PLAN_LOG.append((a.M, a.N, a.K, gemm_plan(a)))


def gemm_raw(A, B, Cout, M, N, K, *args, **kwargs):
    """Launch one GEMM (see gemm_args)."""
    a = gemm_args(A, B, Cout, M, N, K, *args, **kwargs)
    _log_plan(a)
    timed = TIMED is not None and TIMED["shape"] == (M, N, K)
    if timed:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    L.check(L.load().ick_gemm(C.byref(a), _stream()), "ick_gemm")
    if timed:
        e1.record()
        TIMED["events"].append((e0, e1))
    return Cout


def gemm_grouped(problems):
    """Launch a list of ick_gemm_args; problems of one kernel configuration share a launch."""
    for i in range(0, len(problems), 64):
        chunk = problems[i:i + 64]
        arr = (L.GemmArgs * len(chunk))(*chunk)
        L.check(L.load().ick_gemm_grouped(arr, len(chunk), _stream()), "ick_gemm_grouped")


def linear(x, w, bias=None, out=None, relu=False, drop=None, w_ps=None):
    """y = x @ w.T + bias for x (..., K) with contiguous last dim and uniform row stride, w (N, K)."""
    _f32c(x, "x"); _f32c(w, "w")
    K = x.shape[-1]
    N = w.shape[0]
    x2 = x.reshape(-1, K)
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    M = x2.shape[0]
    if out is None:
        out = torch.empty(x.shape[:-1] + (N,), device=x.device, dtype=torch.float32)
    o2 = out.view(-1, N) if out.is_contiguous() else out
    gemm_raw(x2, w, o2, M, N, K, x2.stride(0), 1, w.stride(0), 1, o2.stride(0), bias=bias, relu=relu, drop=drop,
             b_ps=w_ps)
    return out


def add_layernorm(x, res, gamma, beta, eps=1e-5, out=None, save_stats=False, drop=None):
    """LayerNorm(dropout(x) + res) * gamma + beta (dropout only in training: drop = (p, seed, site))."""
    d = x.shape[-1]
    x2 = x.reshape(-1, d)
    r2 = None if res is None else res.reshape(-1, d)
    rows = x2.shape[0]
    if out is None:
        out = torch.empty_like(x2)
    o2 = out.view(-1, d)
    mean = rstd = None
    if save_stats:
        mean = torch.empty(rows, device=x.device, dtype=torch.float32)
        rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
    L.check(L.load().ick_add_layernorm(_p(x2), _p(r2), _p(gamma), _p(beta), _p(o2), rows, d, eps, x2.stride(0),
                                       0 if r2 is None else r2.stride(0), o2.stride(0), _p(mean), _p(rstd),
                                       *_dargs(drop), _stream()), "ick_add_layernorm")
    out = out.view(x.shape)
    return (out, mean, rstd) if save_stats else out


def rowchain_supported(K1, d, N2=0):
    """Can ick_rowchain_fwd run a chain with these widths (GEMM 1 input, LayerNorm width, GEMM 2 output)?"""
    return bool(L.load_raw().ick_rowchain_supported(int(K1), int(d), int(N2)))


def packed_weight_floats(N, K):
    n = L.i64()
    L.check(L.load_raw().ick_packed_weight_floats(int(N), int(K), C.byref(n)), "ick_packed_weight_floats")
    return int(n.value)


def pack_weights(pairs, copies=()):
    """[(m (N, K) 2-D view with positive strides, dst flat float buffer of packed_weight_floats(N, K))]: the packed
    copies the row-chain kernels read (include/ick_amd.h), up to 48 matrices per launch.  A transposed view (w.t())
    gives the operand of the data-gradient chains.  copies: [(src (N, K) view, dst (N, K) view with unit column
    stride)] plain copies that ride in the same launch."""
    todo = [(s_, d_, 0) for s_, d_ in pairs] + [(s_, d_, 1) for s_, d_ in copies]
    for i in range(0, len(todo), 48):
        chunk = todo[i:i + 48]
        items = (L.PackItem * len(chunk))()
        for it, (src, dst, plain) in zip(items, chunk):
            assert src.dim() == 2
            it.src, it.dst, it.N, it.K = _p(src), _p(dst), src.shape[0], src.shape[1]
            it.src_rs, it.src_cs = max(1, src.stride(0)), src.stride(1)
            if plain:
                assert dst.shape == src.shape and (dst.stride(1) == 1 or dst.shape[1] == 1)
                it.dst_rs = max(dst.stride(0), src.shape[1])
            else:
                assert dst.is_contiguous() and dst.numel() == packed_weight_floats(src.shape[0], src.shape[1])
                it.dst_rs = 0
        L.check(L.load().ick_pack_weights(items, len(chunk), _stream()), "ick_pack_weights")


def pack_weight(w):
    """Packed copy of one (N, K) weight (a new tensor)."""
    dst = torch.empty(packed_weight_floats(w.shape[0], w.shape[1]), device=w.device, dtype=torch.float32)
    pack_weights([(w, dst)])
    return dst


def rowchain_fwd(a, w1p, b1, res, gamma, beta, eps, x_out, drop1=None, o_out=None, save_stats=False,
                 w2p=None, b2=None, y2=None, relu=False, drop2=None, heads=None, slim=False):
    """One launch for  o = a @ W1.T + b1;  x = LayerNorm(res + dropout1(o)) * gamma + beta;  y2 = act(x @ W2.T + b2)
    (dropout2 on y2).  a (..., K1) rows with a uniform row stride; w1p / w2p are packed copies (pack_weights) of the
    (d, K1) and (N2, d) nn.Linear weights; b2 gives N2; x_out (..., d) may be a (B, T, d) view with a sample stride
    (rows inside the memory buffer); heads = (nseg, H, S, s0, grp): y2 is the head-major buffer (B, nseg, H, S, DHP)
    of project_heads.  slim: 8-wave workgroups (ICK_CHAIN_SLIM), for chains that run beside bulk GEMMs on another
    stream.  Returns (mean, rstd) when save_stats."""
    K1, d = a.shape[-1], gamma.shape[0]
    a2 = a.reshape(-1, K1)
    if a2.stride(1) != 1:
        a2 = a2.contiguous()
    M = a2.shape[0]
    r2 = res.reshape(-1, d)
    g = L.RowChainArgs()
    g.A, g.a_rs, g.a_grp, g.a_gs = _p(a2), a2.stride(0), 0, 0
    g.M, g.K1, g.d = M, K1, d
    g.w1p, g.b1 = _p(w1p), _p(b1)
    g.res, g.res_rs = _p(r2), r2.stride(0)
    g.gamma, g.beta, g.eps = _p(gamma), _p(beta), eps
    g.drop1_p, g.drop_seed, g.drop1_site, g.drop_epoch = _dargs(drop1)
    if o_out is not None:
        g.o, g.o_rs = _p(o_out), o_out.reshape(-1, d).stride(0)
    if x_out.dim() == 3 and not x_out.is_contiguous():
        assert x_out.stride(2) == 1
        g.x, g.x_rs, g.x_grp, g.x_gs = _p(x_out), x_out.stride(1), x_out.shape[1], x_out.stride(0)
    else:
        g.x, g.x_rs, g.x_grp, g.x_gs = _p(x_out), x_out.reshape(-1, d).stride(0), 0, 0
    mean = rstd = None
    if save_stats:
        mean = torch.empty(M, device=a.device, dtype=torch.float32)
        rstd = torch.empty(M, device=a.device, dtype=torch.float32)
        g.mean, g.rstd = _p(mean), _p(rstd)
    if w2p is not None:
        N2 = b2.shape[0]
        g.w2p, g.b2, g.N2, g.flags = _p(w2p), _p(b2), N2, (1 if relu else 0)
        p2, seed2, site2, ep2 = _dargs(drop2)
        g.drop2_p, g.drop2_site = p2, site2
        if p2 > 0.0:
            if g.drop1_p > 0.0:
                assert seed2 == g.drop_seed
            else:
                g.drop_seed, g.drop_epoch = seed2, ep2
        g.y2 = _p(y2)
        if heads is not None:
            nseg, H, S, s0, grp = heads
            g.y2_rs, g.y2_grp, g.y2_gs = 0, grp, y2.stride(0)
            g.hs_dh, g.hs_dhp, g.hs_H, g.hs_S, g.hs_s0 = (N2 // nseg) // H, DHP, H, S, s0
        else:
            g.y2_rs, g.y2_grp, g.y2_gs = y2.reshape(-1, N2).stride(0), 0, 0
    if slim:
        g.flags |= 256
    L.check(L.load().ick_rowchain_fwd(C.byref(g), _stream()), "ick_rowchain_fwd")
    return (mean, rstd) if save_stats else None


def rowchain_bwd_supported(K0, d, N1=0):
    return bool(L.load_raw().ick_rowchain_bwd_supported(int(K0), int(d), int(N1)))


def ln_partials(rows, d, device):
    """Buffer for the per-workgroup gamma / beta partial sums of a LayerNorm backward (8 rows per workgroup)."""
    rpb = L.load_raw().ick_layernorm_bwd_rows_per_block()
    return torch.empty((rows + rpb - 1) // rpb, 2 * d, device=device, dtype=torch.float32)


def rowchain_bwd(M, d, norm1, w3p, out3, dz_out, g0=None, w0p=None, dzin=None, ffn=None, norm2=None):
    """One launch for the data-gradient path between two attention-backward kernels (include/ick_amd.h,
    ick_rowchain_bwd).  norm1 / norm2 = dict(o, res, mean, rstd, gamma, drop, do, part): saved forward tensors of an
    add & norm, its dropout triple, and the outputs do (M, d) / part (ln_partials).  ffn = dict(w1p, w2p, act,
    gate_scale, t_out) with packed linear2.weight.T / linear1.weight.T.  g0 (M, K0) @ W0 (packed W0.T = w0p) and dzin
    (M, d) are the two addends of the incoming gradient.  (An 8-wave form of this kernel was built in round 4 -- the 16-wave
    one takes 480 of a SIMD's 512 registers, so its workgroups only start on CUs that hold nothing of the other stream --
    and measured slower inside the step all the same, 1.746 against 1.700 ms, profiles/r04_y_ab_slim_bwd.txt; removed.)"""
    a = L.RowChainBwdArgs()
    a.M, a.d = M, d
    a.flags = 0
    seeds = []

    def norm(prefix, n):
        setattr(a, "o" + prefix, _p(n["o"])); setattr(a, "res" + prefix, _p(n["res"]))
        setattr(a, "mean" + prefix, _p(n["mean"])); setattr(a, "rstd" + prefix, _p(n["rstd"]))
        setattr(a, "gamma" + prefix, _p(n["gamma"]))
        pp, seed, site, ep = _dargs(n.get("drop"))
        setattr(a, "drop%s_p" % prefix, pp); setattr(a, "drop%s_site" % prefix, site)
        if pp > 0.0:
            seeds.append((seed, ep))
        setattr(a, "do" + prefix, _p(n["do"])); setattr(a, "part" + prefix, _p(n["part"]))

    norm("1", norm1)
    if g0 is not None:
        if g0.dim() == 3 and not g0.is_contiguous():
            # rows of a (B, T, K0) view with a sample stride (the context rows of the cross K/V gradient buffer)
            assert g0.stride(2) == 1 and g0.shape[0] * g0.shape[1] == M
            a.g0, a.g0_rs, a.K0, a.w0p = _p(g0), g0.stride(1), g0.shape[2], _p(w0p)
            a.g0_grp, a.g0_gs = g0.shape[1], g0.stride(0)
        else:
            g2 = g0.reshape(M, -1)
            assert g2.stride(1) == 1
            a.g0, a.g0_rs, a.K0, a.w0p = _p(g2), g2.stride(0), g2.shape[1], _p(w0p)
    if dzin is not None:
        z2 = dzin.reshape(M, d)
        a.dzin, a.dzin_rs = _p(z2), z2.stride(0)
    if ffn is not None:
        a.w1p, a.N1, a.act, a.gate_scale = _p(ffn["w1p"]), ffn["act"].shape[-1], _p(ffn["act"]), float(ffn["gate_scale"])
        a.t_out, a.w2p = _p(ffn["t_out"]), _p(ffn["w2p"])
        norm("2", norm2)
    if seeds:
        assert all(s == seeds[0] for s in seeds)
        a.drop_seed, a.drop_epoch = seeds[0]
    a.w3p, a.out3, a.dz_out = _p(w3p), _p(out3), _p(dz_out)
    L.check(L.load().ick_rowchain_bwd(C.byref(a), _stream()), "ick_rowchain_bwd")


def attention_raw(Q, K, V, O, B, H, T, S, dh, q_bs, q_hs, q_ts, k_bs, k_hs, k_ss, v_bs, v_hs, v_ss, o_bs, o_ts,
                  causal=False, q_pos0=0, kv_len=None, lse=None, q_off=0, k_off=0, v_off=0, drop=None):
    """Strides in elements: batch / head / row for Q, K, V (see include/ick_amd.h); q_off/k_off/v_off are
    element offsets of the first (segment) inside a packed buffer."""
    a = L.AttnArgs()
    a.Q = Q.data_ptr() + 4 * q_off
    a.K = K.data_ptr() + 4 * k_off
    a.V = V.data_ptr() + 4 * v_off
    a.O, a.lse = _p(O), _p(lse)
    a.B, a.H, a.T, a.S, a.dh = B, H, T, S, dh
    a.q_bs, a.q_hs, a.q_ts = q_bs, q_hs, q_ts
    a.k_bs, a.k_hs, a.k_ss = k_bs, k_hs, k_ss
    a.v_bs, a.v_hs, a.v_ss = v_bs, v_hs, v_ss
    a.o_bs, a.o_ts = o_bs, o_ts
    a.scale = 1.0 / math.sqrt(dh)
    a.causal, a.q_pos0, a.kv_len = int(causal), q_pos0, _p(kv_len)
    _drop(a, drop)
    L.check(L.load().ick_attention(C.byref(a), _stream()), "ick_attention")
    return O


def attention(q, k, v, H, causal=False):
    """q (B,T,d), k/v (B,S,d) row-major -> (B,T,d)."""
    B, T, d = q.shape
    S = k.shape[1]
    dh = d // H
    out = torch.empty_like(q)
    attention_raw(q, k, v, out, B, H, T, S, dh, q.stride(0), dh, q.stride(1), k.stride(0), dh, k.stride(1),
                  v.stride(0), dh, v.stride(1), out.stride(0), out.stride(1), causal=causal)
    return out


DHP = 32  # padded head width of the head-major projection buffers


def project_heads(x, w, bias, nseg, H, S, out=None, s0=0, grp=None, a_gmap=None, a_gs=None, w_ps=None):
    """Packed projection x (B*grp rows, K) @ w.T (nseg*d, K) scattered into the head-major layout
    out (B, nseg, H, S, DHP) at positions s0 .. s0+grp-1 (ick_gemm head-split epilogue)."""
    d = w.shape[0] // nseg
    K = w.shape[1]
    strided = x.dim() == 3 and not x.is_contiguous() and x.stride(2) == 1 and a_gmap is None
    if strided:
        # rows of a (B, T, K) view with a sample stride (the image rows inside the (B, S, d) memory buffer): no copy,
        # the GEMM walks the groups itself
        x2 = x
        M = x.shape[0] * x.shape[1]
    else:
        x2 = x.reshape(-1, K)
        M = x2.shape[0]
    grp = grp if grp is not None else (x.shape[1] if x.dim() == 3 else M)
    Bn = M // grp
    if out is None:
        out = torch.empty(Bn, nseg, H, S, DHP, device=x.device, dtype=torch.float32)
    a = L.GemmArgs()
    a.A, a.B, a.C, a.bias = _p(x2), _p(w), _p(out), _p(bias)
    a.M, a.N, a.K = M, w.shape[0], K
    a.a_rs, a.a_ks = (x.stride(1) if strided else x2.stride(0)), 1
    if strided:
        a.a_grp, a.a_gs = x.shape[1], x.stride(0)
    if a_gmap is not None:
        a.a_grp, a.a_gs, a.a_gmap = grp, a_gs, _p(a_gmap)
    a.b_rs, a.b_ks = w.stride(0), 1
    a.c_rs, a.c_grp, a.c_gs = 0, grp, out.stride(0)
    a.split_k, a.alpha = 1, 1.0
    a.a_extent, a.b_extent = _extent(x2), _extent(w)
    a.hs_dh, a.hs_dhp, a.hs_H, a.hs_S, a.hs_s0 = d // H, DHP, H, S, s0
    if w_ps is not None:
        assert w_ps.numel() == presplit_bytes(w.shape[0], K)
        a.b_ps = _p(w_ps)
    _log_plan(a)
    L.check(L.load().ick_gemm(C.byref(a), _stream()), "ick_gemm(head-split)")
    return out


def attention_heads(q, kv, O, H, dh, T, S, q_seg=0, k_seg=0, v_seg=1, causal=False, kv_len=None, q_pos0=0, q_t0=0,
                    lse=None, drop=None):
    """Attention over head-major buffers: q (B, nq, H, Tq_alloc, DHP), kv (B, nkv, H, S_alloc, DHP);
    the first T query rows starting at q_t0 attend to the first S key rows."""
    B = q.shape[0]
    Tq, Sa = q.shape[3], kv.shape[3]
    attention_raw(q, kv, kv, O, B, H, T, S, dh,
                  q.stride(0), Tq * DHP, DHP, kv.stride(0), Sa * DHP, DHP, kv.stride(0), Sa * DHP, DHP,
                  O.stride(0), O.stride(1), causal=causal, q_pos0=q_pos0, kv_len=kv_len, lse=lse, drop=drop,
                  q_off=q_seg * H * Tq * DHP + q_t0 * DHP, k_off=k_seg * H * Sa * DHP, v_off=v_seg * H * Sa * DHP)
    return O


def entity_encode(variant, entities, type_emb, d, facts=None, word_emb=None):
    B, K, cols = entities.shape
    out = torch.empty(B, K, d, device=entities.device, dtype=torch.float32)
    F = 0 if facts is None else facts.shape[1]
    L.check(L.load().ick_entity_encode(L.VARIANT_ID[variant], _p(entities), cols, _p(facts), _p(type_emb),
                                       type_emb.shape[0], _p(word_emb), 0 if word_emb is None else word_emb.shape[0],
                                       _p(out), B, K, F, d, _stream()), "ick_entity_encode")
    return out


def fact_encode(facts, entities_encoded, pred_emb):
    B, F, _ = facts.shape
    K, d = entities_encoded.shape[1], entities_encoded.shape[2]
    out = torch.empty(B, F, d, device=facts.device, dtype=torch.float32)
    L.check(L.load().ick_fact_encode(_p(facts), _p(entities_encoded), _p(pred_emb), pred_emb.shape[0], _p(out), B, K,
                                     F, d, _stream()), "ick_fact_encode")
    return out


def caption_embed(captions, masks, word_emb, entities_encoded, facts_encoded, pe, V, pad_token, scale, pos0=0,
                  want_emb=False, drop=None):
    B, Lc = captions.shape
    K, d = entities_encoded.shape[1], entities_encoded.shape[2]
    F = 0 if facts_encoded is None else facts_encoded.shape[1]
    out = torch.empty(B, Lc, d, device=captions.device, dtype=torch.float32)
    emb = torch.empty_like(out) if want_emb else None
    L.check(L.load().ick_caption_embed(_p(captions), _p(masks), _p(word_emb), _p(entities_encoded),
                                       _p(facts_encoded), _p(pe), _p(out), _p(emb), B, Lc, K, F, V, d, pad_token,
                                       scale, pos0, *_dargs(drop), _stream()), "ick_caption_embed")
    return (out, emb) if want_emb else out


def context_indicators(captions, facts, K, V, fc_pred_wt=None, fc_pred_b=None, mode=0, eib=None, gate=None,
                       dense_pred=0):
    """dense_pred = number of predicates: `gate` becomes the dense (B, T, num_pred) 0/1 predicate indicator instead of
    fc_predicate applied to it."""
    B, Lc = captions.shape
    F = facts.shape[1]
    T = Lc if mode == 0 else 1
    if eib is None:
        eib = torch.empty(B, T, F, device=captions.device, dtype=torch.float32)
    num_pred = d = 0
    if fc_pred_wt is not None:
        num_pred, d = fc_pred_wt.shape
        if gate is None:
            gate = torch.empty(B, T, d, device=captions.device, dtype=torch.float32)
    elif dense_pred:
        num_pred = d = dense_pred
        if gate is None:
            gate = torch.empty(B, T, d, device=captions.device, dtype=torch.float32)
    L.check(L.load().ick_context_indicators(_p(captions), _p(facts), _p(fc_pred_wt), _p(fc_pred_b), _p(eib),
                                            _p(gate), B, Lc, T, K, F, V, num_pred, d, mode, _stream()),
            "ick_context_indicators")
    return eib, gate


def pointer_scores(h, ctx, w, bias, out, col0, ind=None, out_gmap=None):
    B, T, d = h.shape
    Kc = ctx.shape[1]
    L.check(L.load().ick_pointer_scores(_p(h), _p(ctx), _p(w), _p(bias), _p(ind), _p(out), B, T, Kc, d,
                                        out.stride(-2), col0, _p(out_gmap), _stream()), "ick_pointer_scores")
    return out


def mul(a, b, out=None):
    if out is None:
        out = torch.empty_like(a)
    L.check(L.load().ick_mul(_p(a), _p(b), _p(out), a.numel(), _stream()), "ick_mul")
    return out


def top2(scores):
    B, Vx = scores.shape
    best = torch.empty(B, device=scores.device, dtype=torch.int32)
    second = torch.empty_like(best)
    L.check(L.load().ick_top2(_p(scores), scores.stride(0), B, Vx, _p(best), _p(second), _stream()), "ick_top2")
    return best, second


def greedy_select(scores, output, hist, finished, next_token, next_mask, step, V, K, has_facts, end_token):
    """top2 + greedy_update of one decode step in one launch; scores (B, Vx) rows."""
    B, Vx = scores.shape
    L.check(L.load().ick_greedy_select(_p(scores), scores.stride(0), B, Vx, _p(output), _p(hist), _p(finished),
                                       _p(next_token), _p(next_mask), step, output.shape[1], V, K, int(has_facts),
                                       end_token, _stream()), "ick_greedy_select")


def greedy_update(best, second, output, hist, finished, next_token, next_mask, step, V, K, has_facts, end_token):
    B, max_len = output.shape
    L.check(L.load().ick_greedy_update(_p(best), _p(second), _p(output), _p(hist), _p(finished), _p(next_token),
                                       _p(next_mask), B, step, max_len, V, K, int(has_facts), end_token, _stream()),
            "ick_greedy_update")


def decode_supported(d, H, FF, S, max_len):
    return bool(L.load().ick_decode_supported(d, H, FF, S, max_len))


def decode_beam_supported(Vx, beam):
    return bool(L.load().ick_decode_beam_supported(Vx, beam))


def decode_layers(ctx, pos):
    """Decoder stack + score head of one KV-cached decode step (ick_decode_layers); ctx: lib.DecodeCtx."""
    L.check(L.load().ick_decode_layers(C.byref(ctx), pos, _stream()), "ick_decode_layers")


def decode_layers_part(ctx, pos, part):
    """part 1: the first self-attention block (with ctx.sel_state: the previous step's token selection inside it);
    part 2: the rest of the step."""
    L.check(L.load().ick_decode_layers_part(C.byref(ctx), pos, part, _stream()), "ick_decode_layers_part")


def decode_init(ctx, start_token, n_done_init=0):
    L.check(L.load().ick_decode_init(C.byref(ctx), start_token, n_done_init, _stream()), "ick_decode_init")


def decode_select_greedy(ctx, pos):
    L.check(L.load().ick_decode_select_greedy(C.byref(ctx), pos, _stream()), "ick_decode_select_greedy")


def decode_select_beam(ctx, beam_state, pos):
    L.check(L.load().ick_decode_select_beam(C.byref(ctx), C.byref(beam_state), pos, _stream()), "ick_decode_select_beam")


def packed_ce(scores, captions_sorted, decode_len, pad_token, want_grad=False, out_sum=None, out_count=None):
    """Returns (loss_sum (1,), count (1,), dscores or None): token-mean loss = loss_sum / count.  out_sum / out_count:
    one-element float tensors to receive the two scalars (the tail of TrainStep's gradient bucket)."""
    B, Lc, Vx = scores.shape
    dev = scores.device
    row_loss = torch.empty(B * Lc, device=dev, dtype=torch.float32)
    loss_sum = out_sum if out_sum is not None else torch.empty(1, device=dev, dtype=torch.float32)
    count = out_count if out_count is not None else torch.empty(1, device=dev, dtype=torch.float32)
    dscores = None
    if want_grad:     # same (possibly padded) row stride as the scores: the kernel takes one leading dimension
        dscores = torch.empty(B, Lc, scores.stride(1), device=dev, dtype=torch.float32)[:, :, :Vx]
        assert scores.stride(0) == Lc * scores.stride(1) and scores.stride(2) == 1
    L.check(L.load().ick_packed_ce(_p(scores), scores.stride(1), _p(captions_sorted), _p(decode_len), B, Lc, Vx,
                                   pad_token, _p(row_loss), _p(loss_sum), _p(count), _p(dscores), _stream()),
            "ick_packed_ce")
    return loss_sum, count, dscores


# ------------------------------------------------------------------------------------------------
# Backward / training-step wrappers
# ------------------------------------------------------------------------------------------------
def attention_bwd_buffer(shape, T, S, dh, device):
    """Gradient buffer for attention_heads_bwd: uninitialised when the kernel chosen for (T, S, dh) writes every
    element, zeroed when it accumulates query chunks with atomics."""
    if L.load().ick_attention_bwd_overwrites(T, S, dh):
        return torch.empty(shape, device=device, dtype=torch.float32)
    return torch.zeros(shape, device=device, dtype=torch.float32)


def attention_heads_bwd(q, kv, O, dO, lse, dQ, dK, dV, H, dh, T, S, q_seg=0, k_seg=0, v_seg=1, causal=False,
                        drop=None):
    """Backward of attention_heads.  q (B,nq,H,Tq,DHP), kv (B,nkv,H,Sa,DHP); O/dO (B,T,d) row-major;
    dQ (B,T,*) / dK, dV (B,S,*) row-major views (last-dim stride 1; their column offset selects the
    segment), written as [h*dh + j]."""
    B = q.shape[0]
    Tq, Sa = q.shape[3], kv.shape[3]
    a = L.AttnBwdArgs()
    a.Q = q.data_ptr() + 4 * (q_seg * H * Tq * DHP)
    a.K = kv.data_ptr() + 4 * (k_seg * H * Sa * DHP)
    a.V = kv.data_ptr() + 4 * (v_seg * H * Sa * DHP)
    a.O, a.dO, a.lse = _p(O), _p(dO), _p(lse)
    a.dQ, a.dK, a.dV = _p(dQ), _p(dK), _p(dV)
    a.B, a.H, a.T, a.S, a.dh = B, H, T, S, dh
    a.q_bs, a.q_hs, a.q_ts = q.stride(0), Tq * DHP, DHP
    a.k_bs, a.k_hs, a.k_ss = kv.stride(0), Sa * DHP, DHP
    a.v_bs, a.v_hs, a.v_ss = kv.stride(0), Sa * DHP, DHP
    a.o_bs, a.o_ts = O.stride(0), O.stride(1)
    a.dq_bs, a.dq_ts = dQ.stride(0), dQ.stride(1)
    a.dk_bs, a.dk_ss = dK.stride(0), dK.stride(1)
    a.dv_bs, a.dv_ss = dV.stride(0), dV.stride(1)
    a.scale = 1.0 / math.sqrt(dh)
    a.causal, a.q_pos0 = int(causal), 0
    _drop(a, drop)
    L.check(L.load().ick_attention_bwd(C.byref(a), _stream()), "ick_attention_bwd")


def layernorm_bwd(dy, x, res, gamma, mean, rstd, dgamma, dbeta, drop=None):
    """Returns (dz, dx): dz = gradient of the residual operand (and of z), dx = gradient of the operand the
    forward applied dropout to.  dx is always its own tensor (a copy of dz without dropout): dz is accumulated
    into in place by the next data-gradient GEMM while dx is still being read by weight-gradient kernels on the
    side stream.  The per-workgroup dgamma / dbeta partial sums are reduced by two column sums (side stream)."""
    d = x.shape[-1]
    rows = x.numel() // d
    dz = torch.empty_like(x)
    dxd = torch.empty_like(x)
    rpb = L.load().ick_layernorm_bwd_rows_per_block()
    part = torch.empty((rows + rpb - 1) // rpb, 2 * d, device=x.device, dtype=torch.float32)
    L.check(L.load().ick_layernorm_bwd(_p(dy), _p(x), _p(res), _p(gamma), _p(mean), _p(rstd), _p(dz), None, None,
                                       rows, d, _p(dxd), *_dargs(drop), _p(part), _stream()), "ick_layernorm_bwd")

    if SIDE is not None:
        SIDE.flush()     # side work marked earlier goes out now that the main chain has its next kernel
    ln_partials_reduce(part, dgamma, dbeta)
    return dz, dxd


def ln_partials_reduce(part, dgamma, dbeta):
    """dgamma / dbeta += column sums of the per-workgroup partials (rows x 2d) of a LayerNorm backward: inside the layer's
    grouped weight-gradient launch when a SideStream is installed (no kernel of their own), else two column sums."""
    d = part.shape[1] // 2
    adjacent = dbeta.data_ptr() == dgamma.data_ptr() + 4 * d      # adjacent in a flat gradient bucket: one problem
    if SIDE is not None:
        if adjacent:
            SIDE.add_problem(colsum_problem(part, dgamma), part)
        else:
            SIDE.add_problem(colsum_problem(part[:, :d], dgamma), part)
            SIDE.add_problem(colsum_problem(part[:, d:], dbeta), part)
    elif adjacent:
        colsum(part, dgamma, n_out=2 * d)
    else:
        colsum(part[:, :d], dgamma)
        colsum(part[:, d:], dbeta)


def relu_bwd(dy, act, out=None, scale=1.0):
    if out is None:
        out = torch.empty_like(dy)
    L.check(L.load().ick_relu_bwd(_p(dy), _p(act), _p(out), dy.numel(), scale, _stream()), "ick_relu_bwd")
    return out


def dropout_mask(rows, cols, p, seed, site, device="cuda"):
    out = torch.empty(rows, cols, device=device, dtype=torch.float32)
    L.check(L.load().ick_dropout_mask(_p(out), rows, cols, p, seed & 0xFFFFFFFF, site & 0xFFFFFFFF, _stream()),
            "ick_dropout_mask")
    return out


def colsum(a2d, out, n_out=None):
    """out[n] += sum_m a2d[m, n] for a row-major 2-D view (n_out: out really has that many elements from its
    data pointer on, e.g. two adjacent gradient buffers)."""
    M, N = a2d.shape
    assert (n_out or out.numel()) >= N
    L.check(L.load().ick_colsum(_p(a2d), M, N, a2d.stride(0), _p(out), _stream()), "ick_colsum")
    return out


# ------------------------------------------------------------------------------------------------
# Diagnostic time stamps (ICK_TIMESTAMPS=1): device wall-clock marks on whatever stream is current
# ------------------------------------------------------------------------------------------------
STAMPS = None   # {"buf": int64 tensor, "names": [...]} while enabled


def stamps_enable(n=512):
    global STAMPS
    STAMPS = {"buf": torch.zeros(n, device="cuda", dtype=torch.int64), "names": []}


def stamp(name):
    """Mark the current point of the current stream (no-op unless stamps_enable() was called)."""
    if STAMPS is None or len(STAMPS["names"]) >= STAMPS["buf"].numel():
        return
    i = len(STAMPS["names"])
    STAMPS["names"].append(name)
    L.check(L.load().ick_timestamp(STAMPS["buf"][i:].data_ptr(), _stream()), "ick_timestamp")


def stamps_report():
    """[(name, microseconds since the first stamp)] sorted by time."""
    t = STAMPS["buf"][:len(STAMPS["names"])].cpu().tolist()
    t0 = min(x for x in t if x > 0)
    return sorted(((n, (x - t0) / 100.0) for n, x in zip(STAMPS["names"], t)), key=lambda p: p[1])


def copy_batch(dst, src):
    """dst[i].copy_(src[i]) for up to 8 pairs of contiguous same-sized device tensors per launch (ick_copy_batch)."""
    for i in range(0, len(dst), 8):
        d, s_ = dst[i:i + 8], src[i:i + 8]
        n = len(d)
        sp = (C.c_void_p * n)(*[t.data_ptr() for t in s_])
        dp = (C.c_void_p * n)(*[t.data_ptr() for t in d])
        nb = (C.c_longlong * n)(*[t.numel() * t.element_size() for t in d])
        L.check(L.load().ick_copy_batch(sp, dp, nb, n, _stream()), "ick_copy_batch")


@contextlib.contextmanager
def capture(graph):
    """torch.cuda.graph(graph) for the package's hipGraph captures, with Python's cyclic garbage collector held off until
    the capture has ended.  A collection that starts inside the capture region may finalise objects of EARLIER work that
    sit in reference cycles -- a finished TrainStep's CUDAGraph, its streams -- and destroying those while the thread's
    stream is capturing aborts the process (seen once in the full GPU suite: `Fatal Python error: Aborted`,
    `Garbage-collecting`, inside a captured backward pass).  torch.cuda.graph itself collects right before it begins the
    capture, so nothing is left waiting; whatever the captured function leaves behind is collected after capture_end.
    thread_local: other threads (the RCCL watchdog) may touch the HIP runtime while this one captures."""
    was = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            yield
    finally:
        if was:
            gc.enable()


class SideStream:
    """Second HIP stream for work that is off the critical path: in the backward pass the weight and bias
    gradients of a Linear only feed the optimizer, while the data gradient feeds the next layer's backward.

    fork(): the side stream waits for everything enqueued so far on the main stream (or for an earlier mark());
    join(): the main stream waits for the side stream (before the all-reduce / optimizer).
    submit()/flush(): deferred form -- the dependency point is marked now, the side work is enqueued at the next
    flush().  Callers flush right *after* enqueuing the next main-stream kernel, so that in a captured graph the
    main chain is the first child of every node: hipGraph keeps a node's first child on the parent's queue,
    and a hop to another queue costs ~15 us of idle time on the critical path."""

    def __init__(self, priority=0):
        # priority -1: the chain of small kernels on the side stream competes with a large GEMM on the main stream
        # for workgroup slots; a high-priority queue gets its workgroups dispatched first
        # ICK_GROUP_SAME_STREAM=1 (bench.py's per-kernel pass): same grouping and launch order as the captured step, but
        # everything on the caller's stream, so that HIP events around a launch time that launch alone
        self.stream = torch.cuda.current_stream() if os.environ.get("ICK_GROUP_SAME_STREAM") else \
            torch.cuda.Stream(priority=priority)
        self.pending = False
        self.keep = []   # tensors the side stream reads stay referenced until the pass ends, so the caching
                         # allocator cannot hand their memory to the main stream meanwhile (also under capture)
        self.deferred = []
        self.group = []  # weight-gradient problems (ick_gemm_args) waiting for the next flush_group()
        self.signals = {}

    def add_problem(self, args, *tensors):
        """Queue a GEMM whose operands are complete on the main stream by the next flush_group()."""
        self.group.append(args)
        self.keep.extend(t for t in tensors if t is not None)

    def flush_group(self):
        """One grouped launch (ick_gemm_grouped) of the queued problems on the side stream, ordered after
        everything enqueued so far on the main stream; like submit() it is enqueued at the next flush()."""
        if self.group:
            problems, self.group = self.group, []

            def launch():
                stamp("side: group of %d starts" % len(problems))
                gemm_grouped(problems)
                stamp("side: group of %d done" % len(problems))

            self.deferred.append((self.mark(), launch, ()))

    def flush_group_here(self):
        """The queued problems as one grouped launch on the CALLER's stream, right now: for the tail of a pass, where the
        main stream has run out of work and the side stream is what finishes last."""
        if self.group:
            problems, self.group = self.group, []
            gemm_grouped(problems)

    def mark(self):
        """Event at the current point of the main stream, for a later fork(..., after=event)."""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        return ev

    def fork(self, *tensors, after=None):
        ev = after if after is not None else self.mark()
        self.stream.wait_event(ev)
        self.keep.extend(t for t in tensors if t is not None)
        self.pending = True
        return torch.cuda.stream(self.stream)

    def signal(self, key):
        """Called from work running on the side stream: marks `key` as done at this point of it."""
        ev = torch.cuda.Event()
        ev.record(self.stream)
        self.signals[key] = ev

    def wait(self, key):
        """The main stream waits for the point signal(key) marked (no-op for keys never signalled)."""
        ev = self.signals.get(key)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def wait_or_join(self, key):
        """The main stream waits for signal(key) if deferred work signalled it (after enqueuing that work), else for
        the whole side stream."""
        self.flush_group()
        self.flush()
        if key in self.signals:
            self.wait(key)
        else:
            self.join()

    def submit(self, fn, *tensors):
        self.deferred.append((self.mark(), fn, tensors))

    def flush(self):
        work, self.deferred = self.deferred, []
        for ev, fn, tensors in work:
            with self.fork(*tensors, after=ev):
                fn()

    def join(self):
        self.flush_group()
        self.flush()
        if self.pending:
            torch.cuda.current_stream().wait_stream(self.stream)
            self.pending = False


SIDE = None   # set by training.TrainStep / backward_from_tape for the duration of a backward pass


def linear_bwd(dy, x, w, dw, db, need_dx=True, dx=None, accumulate_dx=False, group_now=False, gate=None,
               gate_scale=1.0, wt_ps=None, xt_ps=None):
    """Backward of y = x @ w.T + b for row-major 2-D views dy (M,N), x (M,K), w (N,K):
    dw += dy.T @ x (split-K over M, float atomics), db += colsum(dy), dx = dy @ w.
    With a SideStream installed the two parameter gradients run beside the data gradient.
    wt_ps: the pre-split copy of w.t() (presplit_weights) -- the data gradient's B operand.
    xt_ps: the pre-split copy of x.t() -- the weight gradient's B operand: dw then runs on csrc/gemm_ps.hip's kernel (the
    vocabulary: 10 000 x 300 outputs over 1 280 rows) and db becomes a column-sum problem of the same group."""
    M, N = dy.shape
    K = x.shape[1]

    # dw += dy.T @ x with db += colsum(dy) riding on the first tile column of the same kernel
    wg = None
    extra = []
    if dw is not None and xt_ps is not None and gemm_split_mode() >= 1 and not is_deterministic():
        # 128 x 128 tiles, two workgroups per CU: K slices so that the ~512 slots of the chip are taken once
        tiles = ((N + 127) // 128) * ((K + 127) // 128)
        split = max(1, min(8, 480 // tiles, M // 512))
        wg = gemm_args(dy, x, dw, N, K, M, 1, dy.stride(0), 1, x.stride(0), dw.stride(0), atomic=True,
                       split_k=split, b_ps=xt_ps)
        if db is not None:
            extra.append(colsum_problem(dy, db, split_k=max(1, min(16, M // 256))))
    elif dw is not None:
        wg = gemm_args(dy, x, dw, N, K, M, 1, dy.stride(0), 1, x.stride(0), dw.stride(0), atomic=True,
                       split_k=wgrad_split(M, N, K, grouped=SIDE is not None), colsum_a=db)

    def param_grads():
        if wg is not None and extra:
            gemm_grouped([wg] + extra)
        elif wg is not None:
            _log_plan(wg)
            L.check(L.load().ick_gemm(C.byref(wg), _stream()), "ick_gemm(wgrad)")
        elif db is not None:
            colsum(dy, db)

    # With a SideStream the weight gradients of a whole layer are queued and go out as one grouped launch at the
    # layer's end (SideStream.flush_group): each alone is a ~15 us latency-bound kernel, and every fork point of
    # the captured graph costs the main chain ~4 us.  The main stream never waits for the side stream until the
    # pass ends.
    overlap = SIDE is not None and (dw is not None or db is not None)
    if overlap:
        if wg is not None:
            SIDE.add_problem(wg, dy, x, xt_ps)
            for e in extra:
                SIDE.add_problem(e, dy)
            if group_now:      # a large problem of its own (the vocabulary): runs beside its data gradient
                SIDE.flush_group()
        else:
            SIDE.submit(param_grads, dy, x)
    if need_dx:
        # a long reduction (the vocabulary: N = 10k..50k) over few output tiles is split over workgroups
        split = max(1, min(16, N // 1024)) if (M * K) <= 1280 * 512 else 1
        if is_deterministic():
            split = 1
        elif accumulate_dx and split == 1 and N >= 512 and (M * K) <= 1280 * 512:
            # the output already holds the residual-path gradient: K slices of ~300 can simply add to it
            # (1280 x 300 x 900: 22 -> ~10 us; the kernel is bound by the latency of its K loop)
            split = max(1, min(8, (N + 150) // 300))
        if wt_ps is not None and split > 1 and gemm_split_mode() >= 1 and K <= 320:
            # the pre-split kernel (the plan picks its 128 x 80 tile for split-K problems at most 320 columns wide): K slices
            # so that row tiles x 4 column tiles x slices take the chip's ~512 two-per-CU slots once (cfg2: 10 x 4 x 12)
            split = max(1, min(16, N // 512, 256 // ((M + 63) // 64)))
        if split > 1:
            if dx is None:
                dx = torch.zeros(M, K, device=dy.device, dtype=torch.float32)
            elif not accumulate_dx:
                dx.zero_()
            gemm_raw(dy, w, dx, M, K, N, dy.stride(0), 1, 1, w.stride(0), dx.stride(0), atomic=True, split_k=split,
                     b_ps=wt_ps)
        else:
            if dx is None:
                dx = torch.empty(M, K, device=dy.device, dtype=torch.float32)
            # gate: the consumer wants ReLU'(act) * dx (FFN inner activation): applied in the epilogue
            gemm_raw(dy, w, dx, M, K, N, dy.stride(0), 1, 1, w.stride(0), dx.stride(0), accumulate=accumulate_dx,
                     gate=gate, gate_scale=gate_scale)
            gate = None
    if not overlap:
        param_grads()
    if gate is not None:        # split-K path: the gate cannot ride on partial sums
        relu_bwd(dx, gate, out=dx, scale=gate_scale)
    return dx


def caption_embed_bwd(dx, captions, masks, dword, dee, dfe, V, pad_token, scale, drop=None):
    B, Lc = captions.shape
    K, d = dee.shape[1], dee.shape[2]
    F = 0 if dfe is None else dfe.shape[1]
    L.check(L.load().ick_caption_embed_bwd(_p(dx), _p(captions), _p(masks), _p(dword), _p(dee), _p(dfe), B, Lc, K, F,
                                           V, d, pad_token, scale, *_dargs(drop), _stream()), "ick_caption_embed_bwd")


def pointer_scores_bwd(dscores, col0, h, ctx, w, ind, dh, dctx, dw, dbias):
    B, T, d = h.shape
    Kc = ctx.shape[1]
    L.check(L.load().ick_pointer_scores_bwd(_p(dscores), dscores.stride(-2), col0, _p(h), _p(ctx), _p(w), _p(ind),
                                            _p(dh), _p(dctx), _p(dw), _p(dbias), B, T, Kc, d, _stream()),
            "ick_pointer_scores_bwd")


def entity_encode_bwd(variant, dee, entities, ee, dtype_emb, word_emb=None, dword=None):
    B, K, cols = entities.shape
    d = dee.shape[2]
    L.check(L.load().ick_entity_encode_bwd(L.VARIANT_ID[variant], _p(dee), _p(entities), cols, _p(ee), _p(word_emb),
                                           0 if word_emb is None else word_emb.shape[0], _p(dtype_emb),
                                           dtype_emb.shape[0], _p(dword), B, K, d, _stream()),
            "ick_entity_encode_bwd")


def fact_encode_bwd(dfe, facts, dee, dpred):
    B, F, d = dfe.shape
    K = dee.shape[1]
    L.check(L.load().ick_fact_encode_bwd(_p(dfe), _p(facts), _p(dee), _p(dpred), dpred.shape[0], B, K, F, d,
                                         _stream()), "ick_fact_encode_bwd")


def context_gate_bwd(captions, facts, dgate, dw, dbias, K, V, mode=0):
    B, Lc = captions.shape
    F = facts.shape[1]
    T, d = dgate.shape[1], dgate.shape[2]
    L.check(L.load().ick_context_gate_bwd(_p(captions), _p(facts), _p(dgate), _p(dw), _p(dbias), B, Lc, T, K, F, V,
                                          dw.shape[1], d, mode, _stream()), "ick_context_gate_bwd")


def adam_clamp(p, g, m, v, step, lr, clip=5.0, gscale=1.0, beta1=0.9, beta2=0.999, eps=1e-8, step_tensor=None,
               gscale_den=None):
    """step (+ the uint32 device counter step_tensor, if given) is Adam's 1-based step count; the gradient is
    scaled by gscale (/ the device scalar gscale_den, if given) before the clamp."""
    L.check(L.load().ick_adam_clamp(_p(p), _p(g), _p(m), _p(v), p.numel(), gscale, clip, lr, beta1, beta2, eps, step,
                                    _p(step_tensor), _p(gscale_den), _stream()), "ick_adam_clamp")


def adam_clamp_derive(p, g, m, v, items_dev, blocks_dev, n_blocks, step, lr, clip=5.0, gscale=1.0, beta1=0.9, beta2=0.999,
                      eps=1e-8, step_tensor=None, gscale_den=None, nbytes=0):
    """adam_clamp over the whole bucket + the re-laid-out copies of the updated weights in the same pass
    (ick_adam_clamp_derive; items_dev / blocks_dev: device arrays of ick_adam_item / ick_adam_block built by
    weights.DerivedWeights).  nbytes: the launch's algorithmic bytes, for profiling.py."""
    L.ADAM_DERIVE_BYTES = nbytes
    L.check(L.load().ick_adam_clamp_derive(_p(p), _p(g), _p(m), _p(v), _p(items_dev), _p(blocks_dev), n_blocks, gscale,
                                           clip, lr, beta1, beta2, eps, step, _p(step_tensor), _p(gscale_den), _stream()),
            "ick_adam_clamp_derive")


def counter_add(counter, inc=1):
    L.check(L.load().ick_counter_add(_p(counter), inc, _stream()), "ick_counter_add")


def counter_add_if(counter, inc, flag):
    """counter += inc iff flag[0] > 0 (device scalar), see ick_counter_add_if."""
    L.check(L.load().ick_counter_add_if(_p(counter), inc, _p(flag), _stream()), "ick_counter_add_if")


def scale_by_ratio(x, num, den):
    L.check(L.load().ick_scale_by_ratio(_p(x), x.numel(), _p(num), _p(den), _stream()), "ick_scale_by_ratio")
