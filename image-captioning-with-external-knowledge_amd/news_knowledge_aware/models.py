"""Drop-in for the reference's news-knowledge-aware/models.py: as knowledge-aware, with news
entity features (count / in-headline / in-first-paragraph, 20 entity types, 3500 predicates)
and the entity encoding scaled by the mean embedding of the entity's name words
(news-knowledge-aware/models.py:79-134,273-592)."""
import torch

from ick_amd.decoder import (CaptionEmbedder, Encoder, EntityEncoder, FactEncoder, PositionEncoder,  # noqa: F401
                             DecoderTransformer as _Engine)

device = torch.device("cuda" if torch.cuda.is_available() else "cpu")


class DecoderTransformer(_Engine):
    variant = "news"

    def forward(self, captions, encoder_out, caption_masks, caption_lengths, entities, facts):
        return super().forward(captions, encoder_out, caption_masks, caption_lengths, entities, facts)

    def predict(self, encoder_out, max_pred_len, entities, facts):
        return super().predict(encoder_out, max_pred_len, entities, facts)
