import torch

from .detector3d_template import Detector3DTemplate


class CenterPoint(Detector3DTemplate):
    """VFE -> sparse backbone -> HeightCompression -> BEV backbone -> CenterHead (reference detectors/centerpoint.py:4-50)."""

    def __init__(self, model_cfg, num_class, dataset):
        super().__init__(model_cfg=model_cfg, num_class=num_class, dataset=dataset)
        self.module_list = self.build_networks()

    def post_processing(self, batch_dict):
        final = batch_dict['final_box_dicts']
        recall_dict = {}
        thresh = (self.model_cfg['POST_PROCESSING'] if isinstance(self.model_cfg, dict) else self.model_cfg.POST_PROCESSING)['RECALL_THRESH_LIST']
        for index in range(batch_dict['batch_size']):
            recall_dict = self.generate_recall_record(box_preds=final[index]['pred_boxes'], recall_dict=recall_dict, batch_index=index,
                                                      data_dict=batch_dict, thresh_list=thresh)
        return final, recall_dict
