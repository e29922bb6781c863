from .detector3d_template import Detector3DTemplate


class PVRCNN(Detector3DTemplate):
    """VFE -> VoxelBackBone8x -> HeightCompression -> VoxelSetAbstraction -> BaseBEVBackbone -> AnchorHeadSingle -> PointHeadSimple ->
    PVRCNNHead; loss = rpn + point + rcnn (reference detectors/pv_rcnn.py:4-36)."""

    def __init__(self, model_cfg, num_class, dataset):
        super().__init__(model_cfg=model_cfg, num_class=num_class, dataset=dataset)
        self.module_list = self.build_networks()

    def forward(self, batch_dict):
        for cur_module in self.module_list:
            batch_dict = cur_module(batch_dict)
        if self.training:
            loss, tb_dict, disp_dict = self.get_training_loss()
            return {'loss': loss}, tb_dict, disp_dict
        return self.post_processing(batch_dict)

    def get_training_loss(self):
        loss_rpn, tb_dict = self.dense_head.get_loss()
        loss_point, tb_dict = self.point_head.get_loss(tb_dict)
        loss_rcnn, tb_dict = self.roi_head.get_loss(tb_dict)
        return loss_rpn + loss_point + loss_rcnn, tb_dict, {}
