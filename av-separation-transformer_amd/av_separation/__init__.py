"""av_separation -- MI355X-native drop-in for the forward path of danieleschmidt/AV-Separation-Transformer.

Import surface of the reference package (``/root/reference/src/av_separation/__init__.py:6-22``): put
``av-separation-transformer_amd/`` on ``sys.path`` where the reference's ``src/`` used to be, then

    from av_separation import AVSeparationTransformer, SyntheticAVDataset
    from av_separation.model import AudioEncoder, VisualEncoder, CrossModalFusion, SeparationDecoder
    from av_separation.losses import SeparationLoss, si_snr
    from av_separation.evaluate import evaluate_separation          # demo.py's SNR harness
"""
from . import model as _model
from .dataset import SyntheticAVDataset

_MODULES = ("AudioEncoder", "VisualEncoder", "CrossModalFusion", "SeparationDecoder", "AVSeparationTransformer")
globals().update({name: getattr(_model, name) for name in _MODULES})

__all__ = [*_MODULES, "SyntheticAVDataset"]
