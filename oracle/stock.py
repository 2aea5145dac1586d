"""
TEST INFRASTRUCTURE ONLY -- the reference's PyTorch-CPU path re-stated with the same stock
torch modules the reference composes (nn.TransformerDecoder / nn.TransformerEncoder / nn.Linear /
nn.Embedding / nn.Conv2d; geo-aware/models.py:32,241-252, knowledge-aware/models.py:319-337), used
  (a) as the timed `cpu_baseline` ("port") of bench.py, and
  (b) as a second, independently written oracle (tests/test_oracle_golden.py checks it against the
      golden vectors too).
The per-sample Python loops of the reference (CaptionEmbedder, EntityEncoder fact counts,
get_context_indicators) are vectorised here, so this port is never slower than the reference --
the GPU/CPU ratio reported by bench.py is therefore conservative.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math

import torch
from torch import nn

from . import restatement as R


class StockDecoder(nn.Module):
    def __init__(self, variant, word_map, emb_dim=300, decoder_dim=512, encoder_dim=512, num_heads=10,
                 num_layers=3, dropout=0.0):
        super().__init__()
        self.cfg = R.config_from_word_map(variant, word_map, emb_dim, num_heads, num_layers)
        d = emb_dim
        self.transformer_decoder = nn.TransformerDecoder(
            nn.TransformerDecoderLayer(d, num_heads, decoder_dim, dropout), num_layers)
        self.transformer_encoder_entities = nn.TransformerEncoder(
            nn.TransformerEncoderLayer(d, num_heads, encoder_dim, dropout), num_layers, enable_nested_tensor=False)
        if variant != "geo":
            self.transformer_encoder_facts = nn.TransformerEncoder(
                nn.TransformerEncoderLayer(d, num_heads, encoder_dim, dropout), num_layers,
                enable_nested_tensor=False)
            self.predicate_embedding = nn.Embedding(self.cfg.num_predicates, d)
            self.fc_fact = nn.Linear(d, 1)
            self.fc_predicate = nn.Linear(self.cfg.num_predicates, d)
        self.word_embedding = nn.Embedding(self.cfg.vocab_size, d)
        self.type_embedding = nn.Embedding({"geo": 1000, "knowledge": 1000, "news": 20}[variant],
                                           d - self.cfg.type_offset)
        self.fc_vocab = nn.Linear(d, self.cfg.vocab_size)
        self.fc_entity = nn.Linear(d, 1)
        self.conv1 = nn.Conv2d(2048, d, 1)
        self.register_buffer("pe", R.pe_table(5000, d))

    def load_reference_params(self, P, conv_w=None, conv_b=None):
        sd = {k: v for k, v in P.items() if not k.startswith("fact_encoder.")}
        sd["type_embedding.weight"] = sd.pop("entity_encoder.type_embedding.weight")
        sd["pe"] = self.pe
        sd["conv1.weight"] = conv_w if conv_w is not None else self.conv1.weight.detach()
        sd["conv1.bias"] = conv_b if conv_b is not None else self.conv1.bias.detach()
        self.load_state_dict(sd, strict=True)
        return self

    def _params(self):
        P = {"entity_encoder.type_embedding.weight": self.type_embedding.weight,
             "word_embedding.weight": self.word_embedding.weight}
        if self.cfg.has_facts:
            P["predicate_embedding.weight"] = self.predicate_embedding.weight
        return P

    def encode_image(self, feats):
        return self.conv1(feats).view(feats.shape[0], self.cfg.emb_dim, -1)

    def forward(self, captions, enc_out, caption_masks, caption_lengths, entities, facts=None):
        cfg = self.cfg
        lengths, sort_ind = caption_lengths.squeeze(1).sort(dim=0, descending=True)
        enc_out, captions, caption_masks, entities = enc_out[sort_ind], captions[sort_ind], caption_masks[sort_ind], \
            entities[sort_ind]
        if facts is not None:
            facts = facts[sort_ind]
        decode_lengths = (lengths - 1).tolist()
        P = self._params()
        ee = R.entity_encode(cfg, P, entities, facts)
        fe = R.fact_encode(P, facts, ee) if cfg.has_facts else None
        emb = R.caption_embed(cfg, P, captions, caption_masks, ee, fe)
        parts = [enc_out.permute(2, 0, 1), self.transformer_encoder_entities(ee.permute(1, 0, 2))]
        if cfg.has_facts:
            parts.append(self.transformer_encoder_facts(fe.permute(1, 0, 2)))
        mem = torch.cat(parts)
        L = captions.shape[1]
        x = emb.permute(1, 0, 2) * math.sqrt(cfg.emb_dim) + self.pe[:L].unsqueeze(1)
        mask = torch.full((L, L), float("-inf")).triu(1)
        h = self.transformer_decoder(x, mem, mask)  # (L,B,d)
        if cfg.has_facts:
            gate_in = R.context_indicators(cfg, captions, facts, entities.shape[1], L)
            eib, pi = gate_in
            vocab = self.fc_vocab(h * self.fc_predicate(pi).permute(1, 0, 2))
        else:
            vocab = self.fc_vocab(h)
        ent = self.fc_entity(h.unsqueeze(2) * ee.unsqueeze(0)).squeeze(3)
        outs = [vocab, ent]
        if cfg.has_facts:
            outs.append(self.fc_fact(h.unsqueeze(2) * fe.unsqueeze(0) * eib.permute(1, 0, 2).unsqueeze(3)).squeeze(3))
        return torch.cat(outs, dim=2).permute(1, 0, 2), captions, decode_lengths

    @torch.no_grad()
    def predict(self, enc_out, max_pred_len, entities, facts=None):
        """Greedy decode with the reference's structure (geo-aware/models.py:363-443): batch 1, NO KV cache -- the
        whole decoder stack runs over all max_pred_len positions at every step -- softmax, argmax, runner-up,
        n-gram clean-up.  This is what bench.py times as the CPU baseline of the greedy mode."""
        cfg = self.cfg
        assert enc_out.shape[0] == 1
        V, K = cfg.vocab_size, entities.shape[1]
        P = self._params()
        ee = R.entity_encode(cfg, P, entities, facts)
        fe = R.fact_encode(P, facts, ee) if cfg.has_facts else None
        parts = [enc_out.permute(2, 0, 1), self.transformer_encoder_entities(ee.permute(1, 0, 2))]
        if cfg.has_facts:
            parts.append(self.transformer_encoder_facts(fe.permute(1, 0, 2)))
        mem = torch.cat(parts)
        captions = torch.full((1, max_pred_len), cfg.start, dtype=torch.long)
        masks = torch.zeros(1, max_pred_len, dtype=torch.long)
        output = [cfg.pad] * max_pred_len
        prev_top_two = []
        mask = torch.full((max_pred_len, max_pred_len), float("-inf")).triu(1)
        for i in range(max_pred_len):
            emb = R.caption_embed(cfg, P, captions, masks, ee, fe)
            x = emb.permute(1, 0, 2) * math.sqrt(cfg.emb_dim) + self.pe[:max_pred_len].unsqueeze(1)
            h = self.transformer_decoder(x, mem, mask)[i:i + 1]           # (1, 1, d)
            if cfg.has_facts:
                eib, pi = R.context_indicators(cfg, captions, facts, K, 1)
                vocab = self.fc_vocab(h * self.fc_predicate(pi).permute(1, 0, 2))
            else:
                vocab = self.fc_vocab(h)
            outs = [vocab, self.fc_entity(h.unsqueeze(2) * ee.unsqueeze(0)).squeeze(3)]
            if cfg.has_facts:
                outs.append(self.fc_fact(h.unsqueeze(2) * fe.unsqueeze(0) * eib.permute(1, 0, 2).unsqueeze(3)).squeeze(3))
            prob = torch.cat(outs, dim=2)[0, 0].softmax(dim=-1)
            out = int(prob.argmax())
            output[i] = out
            if out == cfg.end:
                break
            prev_top_two.append(int(prob.topk(2).indices[1]))
            R.loop_cleanup(output, prev_top_two, i)
            out = output[i]
            if i < max_pred_len - 1:
                captions[0, i + 1] = out
                masks[0, i + 1] = 2 if (cfg.has_facts and out >= V + K) else (1 if out >= V else 0)
        return torch.tensor(output, dtype=torch.long).view(max_pred_len, 1)
