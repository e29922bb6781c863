"""Same helpers as the reference's pcdet/utils/spconv_utils.py, bound to the seevcn spconv-shaped API."""
from typing import Set

import torch.nn as nn

from ... import spconv  # noqa: F401  (re-exported: `from ...utils.spconv_utils import replace_feature, spconv`)


def find_all_spconv_keys(model: nn.Module, prefix="") -> Set[str]:
    """Names of all sparse-conv weights (the ones whose layout differs between spconv versions)."""
    found: Set[str] = set()
    for name, child in model.named_children():
        new_prefix = f"{prefix}.{name}" if prefix != "" else name
        if isinstance(child, spconv.conv.SparseConvolution):
            found.add(f"{new_prefix}.weight")
        found.update(find_all_spconv_keys(child, prefix=new_prefix))
    return found


def replace_feature(out, new_features):
    if "replace_feature" in out.__dir__():
        return out.replace_feature(new_features)
    out.features = new_features
    return out
