"""Mirror of the reference's VCN API (see/surface_completion/models/): MODELS registry, VCN_VC, VCN_CN, VCN."""
from .models import MODELS, build_model_from_cfg  # noqa: F401
