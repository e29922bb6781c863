from .build import MODELS, build_model_from_cfg  # noqa: F401
from . import VCN_CN  # noqa: F401
from . import VCN_VC  # noqa: F401
