"""Re-laid-out copies of the decoder's weights that the optimizer kernel keeps current (DESIGN.md
§3.1d).  The reference's optimizer.step() (geo-aware/train.py:292) is the only writer of the parameters, so the packed
row-chain images, the gathered cross K/V weight and the bf16 planes of the large GEMMs' weights are written in the same
pass (ick_adam_clamp_derive, csrc/adam_derive.hip) instead of by packing launches in front of every forward pass."""
import torch

from . import ops


def _p(x):
    return x.detach()


class DerivedWeights:
    """The re-laid-out copies of the decoder's weights that a training step's kernels read, as persistent buffers which
    the optimizer kernel itself keeps current (ops.adam_clamp_derive / ick_adam_clamp_derive): nothing re-packs a weight
    between optimizer.step() (geo-aware/train.py:292) and the next forward pass.

      pk / pkb        packed row-chain images of every nn.Linear the chains multiply with, forward and transposed
                      (decoder._chain_pack's persistent buffers), incl. the transposed all-layer cross K/V weight
      wkv, bkv        rows [d:3d] of every decoder layer's cross-attention in_proj gathered into one (2 * layers * d, d)
                      weight and bias (one GEMM projects the memory for all layers)
      wkv_ps          bf16 hi / mid / lo planes of wkv (csrc/gemm_ps.hip's B operand)
      vocab_ps / vocab_t_ps   the planes of fc_vocab.weight and of its transpose (forward, data gradient)
      pred_wt         fc_predicate.weight transposed (knowledge / news variants)

    build() returns None when the widths do not meet the kernel's alignment rules (include/ick_amd.h); the step then keeps
    the per-step packing launches.  refresh() fills every image from the live parameters with the stand-alone packing
    kernels (first step, after load_state_dict, after a capture's rewound warm-up step, after outside writes)."""

    @staticmethod
    def build(ts):
        dec = ts.dec
        if not (dec.chain_supported() and dec.chain_bwd_supported()):
            return None
        try:
            return DerivedWeights(ts)
        except _Unsupported:
            return None

    def __init__(self, ts):
        import ctypes as C
        from . import lib as L
        dec = self.dec = ts.dec
        self.ts = ts
        d, V = dec.emb_dim, dec.vocab_size
        layers = list(dec.transformer_decoder.layers)
        nseg = 2 * len(layers)
        dev = ts.flat_p.device
        base, nfl = ts.flat_p.data_ptr(), ts.n
        self.wkv = torch.empty(nseg * d, d, device=dev, dtype=torch.float32)
        self.bkv = torch.empty(nseg * d, device=dev, dtype=torch.float32)
        self.wkv_ps = ops.presplit_buffer(nseg * d, d, dev)
        self.vocab_ps = ops.presplit_buffer(V, d, dev)
        self.vocab_t_ps = ops.presplit_buffer(d, V, dev)
        self.pred_wt = None
        if dec.has_facts:
            w = dec.fc_predicate.weight
            self.pred_wt = torch.empty(w.shape[1], w.shape[0], device=dev, dtype=torch.float32)
        self.pk = dec._chain_pack()
        self.pkb = dec._chain_pack(bwd=True, extra=[(("kv", "T"), self.wkv.t())])
        self.stale = True
        self._seen = None

        def where(w):
            """Float offset in the bucket of a (row slice of a) trainable parameter; None for frozen ones (their images are
            filled by refresh() and never change)."""
            off = (w.data_ptr() - base) // 4
            if not (0 <= off and off + w.numel() <= nfl):
                return None
            if w.dim() != 2 or w.stride() != (w.shape[1], 1) or w.shape[1] % 4 or off % 4:
                raise _Unsupported()
            return off

        items, nbytes = [], 0

        def item(w, drow0=0, Nd=None, pack=None, pack_t=None, copy=None, ps=None, ps_t=None, tr=None):
            nonlocal nbytes
            off = where(w)
            if off is None:
                return
            rows, K = w.shape
            if (pack_t is not None and (drow0 % 4 or rows % 4)) or (ps_t is not None and (drow0 % 8 or rows % 8)):
                raise _Unsupported()
            it = L.AdamItem()
            it.off, it.rows, it.K, it.drow0, it.Nd = off, rows, K, drow0, Nd if Nd is not None else rows
            for name, t in (("pack", pack), ("pack_t", pack_t), ("copy", copy), ("ps", ps), ("ps_t", ps_t), ("tr", tr)):
                if t is not None:
                    setattr(it, name, t.data_ptr())
                    nbytes += rows * K * (6 if name in ("ps", "ps_t") else 4)
            if copy is not None:
                it.copy_ld = copy.stride(0)
            if tr is not None:
                it.tr_ld = tr.stride(0)
            items.append(it)

        for key, w in dec._chain_items():
            item(w.detach(), pack=self.pk[key], pack_t=self.pkb[(key[0], key[1], key[2] + "T")])
        for i, l in enumerate(layers):
            item(l.multihead_attn.in_proj_weight.detach()[d:], drow0=2 * d * i, Nd=nseg * d, copy=self.wkv, ps=self.wkv_ps,
                 pack_t=self.pkb[("kv", "T")])
        item(dec.fc_vocab.weight.detach(), ps=self.vocab_ps, ps_t=self.vocab_t_ps)
        if dec.has_facts:
            item(dec.fc_predicate.weight.detach(), tr=self.pred_wt)
        # flat runs that are mirrored into a plain copy: rows [d:3d] of the cross-attention in_proj biases -> bkv
        mirrors = []
        for i, l in enumerate(layers):
            b = l.multihead_attn.in_proj_bias.detach()[d:]
            off = (b.data_ptr() - base) // 4
            if 0 <= off and off + b.numel() <= nfl:
                if off % 4 or b.numel() % 4:
                    raise _Unsupported()
                mirrors.append((off, off + b.numel(), self.bkv[2 * d * i:].data_ptr()))
        # ---- the cover of [0, n): tiles of the items, flat runs of <= 1024 float4 everywhere else
        blocks = []

        def flat(lo, hi, copy=0):
            while lo < hi:
                c = min(4096, hi - lo)
                bl = L.AdamBlock()
                bl.item, bl.cnt4, bl.off4, bl.copy = -1, c // 4, lo // 4, copy
                blocks.append(bl)
                lo += c
                if copy:
                    copy += 4 * c

        cuts = sorted([(it.off, it.off + it.rows * it.K, "item", i) for i, it in enumerate(items)] +
                      [(lo, hi, "mirror", ptr) for lo, hi, ptr in mirrors])
        pos = 0
        for lo, hi, kind, what in cuts:
            if lo < pos:
                raise _Unsupported()       # overlapping views of one parameter
            flat(pos, lo)
            if kind == "mirror":
                flat(lo, hi, what)
            else:
                it = items[what]
                for tn in range(it.drow0 // 64, (it.drow0 + it.rows - 1) // 64 + 1):
                    for tk in range((it.K + 63) // 64):
                        bl = L.AdamBlock()
                        bl.item, bl.tn, bl.tk = what, tn, tk
                        blocks.append(bl)
            pos = hi
        flat(pos, nfl)
        assert nfl % 4 == 0
        self.n_blocks = len(blocks)
        self.nbytes = 28 * nfl + nbytes      # seven streams of the update + the images' bytes (profiling)

        def upload(structs, typ):
            arr = (typ * max(1, len(structs)))(*structs)
            return torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)

        self.items_dev = upload(items, L.AdamItem)
        self.blocks_dev = upload(blocks, L.AdamBlock)
        self.n_items = len(items)

    def _key(self):
        return tuple(p._version for p in self.ts.params) + (self.dec.__dict__.get("_param_epoch", 0),)

    def mark_current(self):
        """The optimizer kernel has just written every image from the weights it updated."""
        self.stale = False
        self._seen = self._key()

    def ensure_current(self):
        if self.stale or self._seen != self._key():
            self.refresh()

    def refresh(self):
        """Every image from the live parameters, with the stand-alone packing kernels (eager launches on the current stream)."""
        dec, d = self.dec, self.dec.emb_dim
        copies = []
        for i, l in enumerate(dec.transformer_decoder.layers):
            copies.append((_p(l.multihead_attn.in_proj_weight)[d:], self.wkv[2 * d * i:2 * d * (i + 1)]))
            copies.append((_p(l.multihead_attn.in_proj_bias)[d:].view(1, -1), self.bkv[2 * d * i:2 * d * (i + 1)].view(1, -1)))
        pk = dec._chain_pack(fresh=True, copies=copies)
        pkb = dec._chain_pack(fresh=True, bwd=True, extra=[(("kv", "T"), self.wkv.t())])
        # the images live in the decoder's persistent pack buffers: the views this object handed to the item table
        assert all(pk[k].data_ptr() == v.data_ptr() for k, v in self.pk.items())
        assert all(pkb[k].data_ptr() == v.data_ptr() for k, v in self.pkb.items())
        w = _p(dec.fc_vocab.weight)
        ops.presplit_weights([(self.wkv, self.wkv_ps), (w, self.vocab_ps), (w.t(), self.vocab_t_ps)])
        if self.pred_wt is not None:
            self.pred_wt.copy_(_p(dec.fc_predicate.weight).t())
        self.mark_current()


class _Unsupported(Exception):
    pass
