from ..utils.registry import Registry

MODELS = Registry('models')


def build_model_from_cfg(cfg, **kwargs):
    """Same entry as the reference's models/vcn/models/build.py:7-15."""
    return MODELS.build(cfg, **kwargs)
