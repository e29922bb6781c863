from .anchor_head_single import AnchorHeadSingle
from .anchor_head_template import AnchorHeadTemplate
from .center_head import CenterHead
from .point_head_simple import PointHeadSimple

# same registry shape as the reference (dense_heads/__init__.py:9-17)
__all__ = {
    'AnchorHeadTemplate': AnchorHeadTemplate,
    'AnchorHeadSingle': AnchorHeadSingle,
    'PointHeadSimple': PointHeadSimple,
    'CenterHead': CenterHead,
}
