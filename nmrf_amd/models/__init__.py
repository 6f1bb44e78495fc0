from .nmrf import NMRF, build


def build_model(cfg):
    """nmrf.models.build_model(cfg) -> (NMRF, criterion)   (nmrf/models/__init__.py:9-10)"""
    return build(cfg)
