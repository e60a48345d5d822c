"""MI355X-native VGGHeads forward path behind the reference's HeadDetector API (head_detector/__init__.py)."""
from .head_info import FLAME_CONSTS, Bbox, FlameParams, HeadMetadata, RPY  # noqa: F401

name = "head_detector_amd"
__version__ = "0.1.0"
__all__ = ["HeadDetector", "VGHeadsEngine", "FLAMELayer", "reproject_spatial_vertices", "FlameParams", "HeadMetadata", "Bbox", "RPY", "FLAME_CONSTS"]


def __getattr__(attr):  # lazy: importing the package must work where libvgh.so / a GPU is absent
    if attr == "HeadDetector":
        from .detector import HeadDetector

        return HeadDetector
    if attr == "VGHeadsEngine":
        from .engine import VGHeadsEngine

        return VGHeadsEngine
    if attr in ("FLAMELayer", "reproject_spatial_vertices"):
        from . import flame

        return getattr(flame, attr)
    raise AttributeError(attr)
