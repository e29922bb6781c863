from .second_net import SECONDNet


class PointPillar(SECONDNet):
    """PillarVFE -> PointPillarScatter -> BaseBEVBackbone -> AnchorHeadSingle; identical control flow to SECONDNet
    (reference detectors/pointpillar.py:4-37)."""
