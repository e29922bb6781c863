from .pvrcnn_head import PVRCNNHead
from .roi_head_template import RoIHeadTemplate
from .second_head import SECONDHead

__all__ = {
    'RoIHeadTemplate': RoIHeadTemplate,
    'PVRCNNHead': PVRCNNHead,
    'SECONDHead': SECONDHead,
}
