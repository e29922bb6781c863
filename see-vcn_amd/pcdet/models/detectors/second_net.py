from .detector3d_template import Detector3DTemplate


class SECONDNet(Detector3DTemplate):
    """VFE -> sparse 3-D backbone -> HeightCompression -> BEV backbone -> anchor head (reference detectors/second_net.py:4-37).
    One loss head (the template's default): tb_dict carries 'loss_rpn'."""

    def __init__(self, model_cfg, num_class, dataset):
        super().__init__(model_cfg=model_cfg, num_class=num_class, dataset=dataset)
        self.module_list = self.build_networks()
