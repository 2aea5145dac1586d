"""Greedy caption generation over a test split, mirroring evaluate() of the reference's eval.py
(geo-aware/eval.py:46-125; knowledge-aware/eval.py:46-200 for the fact tokens): encoder -> decoder.predict ->
token ids -> text (vocabulary words, entity names and fact objects decoded from their integer encodings) ->
generated_captions.csv.  The domain metrics that follow in the reference (Jensen-Shannon, fact accuracy, BLEU...)
are CPU text statistics outside this path."""
import os

import pandas as pd
import torch

from . import utils as ut


def detokenize(seq, word_map, rev_word_map, entity_names, fact_names=None):
    """seq: iterable of token ids of ONE caption; entity_names (K, 2+50) / fact_names (F, 2+50) integer-encoded
    names [idx, length, chars...] as produced by the reference's preprocessing."""
    V = len(word_map)
    K = entity_names.shape[0]
    special = {word_map["<start>"], word_map["<end>"], word_map["<pad>"]}
    words = []
    for tok in seq:
        tok = int(tok)
        if tok < V:
            if tok not in special:
                words.append(rev_word_map[tok])
        elif tok < V + K or fact_names is None:
            k = tok - V
            words.append("<unk_ent>" if k >= K else ut.int_to_str(entity_names[k][2:].tolist(), int(entity_names[k][1])))
        else:
            j = tok - V - K
            words.append("<unk_fact>" if j >= fact_names.shape[0]
                         else ut.int_to_str(fact_names[j][2:].tolist(), int(fact_names[j][1])))
    text = " ".join(words)
    if not text.endswith(".") and text.count(".") > 1:       # drop a trailing unfinished sentence
        text = ".".join(text.split(".")[:-1]) + "."
    return text


@torch.no_grad()
def evaluate(encoder, decoder, loader, word_map, max_caption_len=30, out_csv="generated_captions.csv", device="cuda"):
    decoder.eval()
    encoder.eval()
    rev = {v: k for k, v in word_map.items()}
    captions, sequences = [], []
    # precomputed feature maps go to predict() as they are: Encoder.conv1 then runs inside the captured decode graph
    # beside the context encoders (decoder.attach_encoder); raw images go through the encoder's trunk first
    decoder.attach_encoder(encoder)
    img_buf = None
    for batch in loader:                                      # any batch size: captions decode independently
        ent, names = batch[4], batch[5]
        has_facts = len(batch) > 6
        extra = (batch[6].to(device),) if has_facts else ()
        feature_map = batch[0].dim() == 4 and batch[0].shape[1] == encoder.encoder_dim
        if feature_map and img_buf is not None and img_buf.shape == batch[0].shape:
            # the host-to-device copy lands straight in the decode graph's own input buffer (widening a float16 feature
            # file on the way): predict() then skips its device-to-device copy of the feature map (51 MB at batch 32)
            img_buf.copy_(batch[0], non_blocking=True)
            image = img_buf
        else:
            image = batch[0].to(device)
        seq = decoder.predict(image if feature_map else encoder(image), max_caption_len, ent, *extra)   # (max_len, B)
        bufs = decoder.input_buffers() if feature_map else None
        img_buf = bufs[0] if bufs is not None and bufs[0] is not None and bufs[0].dim() == 4 else None
        for b in range(seq.shape[1]):
            ids = seq[:, b].tolist()
            sequences.append(ids)
            captions.append(detokenize(ids, word_map, rev, names[b], batch[7][b] if has_facts else None))
    if out_csv:
        pd.DataFrame({"generated_caption": captions}).to_csv(out_csv, index=False)
    return captions, sequences
