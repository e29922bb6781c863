from .centerpoint import CenterPoint
from .detector3d_template import Detector3DTemplate
from .pointpillar import PointPillar
from .pv_rcnn import PVRCNN
from .second_net import SECONDNet
from .second_net_iou import SECONDNetIoU

# same registry shape as the reference (detectors/__init__.py:13-26)
__all__ = {
    'Detector3DTemplate': Detector3DTemplate,
    'SECONDNet': SECONDNet,
    'SECONDNetIoU': SECONDNetIoU,
    'PointPillar': PointPillar,
    'PVRCNN': PVRCNN,
    'CenterPoint': CenterPoint,
}


def build_detector(model_cfg, num_class, dataset):
    name = model_cfg['NAME'] if isinstance(model_cfg, dict) else model_cfg.NAME
    return __all__[name](model_cfg=model_cfg, num_class=num_class, dataset=dataset)
