from .detector3d_template import Detector3DTemplate


class PVRCNN(Detector3DTemplate):
    """VFE -> VoxelBackBone8x -> HeightCompression -> VoxelSetAbstraction -> BaseBEVBackbone -> AnchorHeadSingle -> PointHeadSimple ->
    PVRCNNHead; training loss = rpn + point + rcnn (reference detectors/pv_rcnn.py:4-36).  Module loop, train / eval branching and
    the loss sum live in Detector3DTemplate."""
    LOSS_HEADS = ('dense_head', 'point_head', 'roi_head')

    def __init__(self, model_cfg, num_class, dataset):
        super().__init__(model_cfg=model_cfg, num_class=num_class, dataset=dataset)
        self.module_list = self.build_networks()
