"""av_separation -- MI355X-native drop-in for the forward path of danieleschmidt/AV-Separation-Transformer.

Same import surface as the reference package (``/root/reference/src/av_separation/__init__.py:6-22``):

    from av_separation import AVSeparationTransformer, SyntheticAVDataset
    from av_separation.model import AudioEncoder, VisualEncoder, CrossModalFusion, SeparationDecoder
    from av_separation.losses import SeparationLoss, si_snr

Put ``av-separation-transformer_amd/`` on ``sys.path`` where the reference's ``src/`` used to be.
"""
from .model import (
    AudioEncoder,
    VisualEncoder,
    CrossModalFusion,
    SeparationDecoder,
    AVSeparationTransformer,
)
from .dataset import SyntheticAVDataset

__all__ = [
    "AudioEncoder",
    "VisualEncoder",
    "CrossModalFusion",
    "SeparationDecoder",
    "AVSeparationTransformer",
    "SyntheticAVDataset",
]
