"""Drop-in for the reference's knowledge-aware/models.py (entity + fact contexts, predicate
gate, fact pointer scores; knowledge-aware/models.py:290-609).  See geo_aware/models.py for
how to install it in front of train.py / eval.py."""
import torch

from ick_amd.decoder import (CaptionEmbedder, Encoder, EntityEncoder, FactEncoder, PositionEncoder,  # noqa: F401
                             DecoderTransformer as _Engine)

device = torch.device("cuda" if torch.cuda.is_available() else "cpu")


class DecoderTransformer(_Engine):
    variant = "knowledge"

    def forward(self, captions, encoder_out, caption_masks, caption_lengths, entities, facts):
        return super().forward(captions, encoder_out, caption_masks, caption_lengths, entities, facts)

    def predict(self, encoder_out, max_pred_len, entities, facts):
        return super().predict(encoder_out, max_pred_len, entities, facts)
