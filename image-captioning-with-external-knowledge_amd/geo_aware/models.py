"""Drop-in for the reference's geo-aware/models.py (same class names, constructor and call
signatures, parameter names and checkpoint layout).  Put this directory first
on sys.path (or copy the file next to train.py / eval.py) so `import models` resolves here; the
reference's whole-object checkpoints then unpickle into these classes.  Computation: HIP kernels
in libick_amd.so, see ick_amd.decoder."""
import torch

from ick_amd.decoder import (CaptionEmbedder, Encoder, EntityEncoder, PositionEncoder,  # noqa: F401
                             DecoderTransformer as _Engine)

device = torch.device("cuda" if torch.cuda.is_available() else "cpu")


class DecoderTransformer(_Engine):
    """geo-aware/models.py:212-443: entity context only (E+1 geographic entities)."""
    variant = "geo"

    def forward(self, captions, encoder_out, caption_masks, caption_lengths, entities):
        return super().forward(captions, encoder_out, caption_masks, caption_lengths, entities)

    def predict(self, encoder_out, max_pred_len, entities):
        return super().predict(encoder_out, max_pred_len, entities)
