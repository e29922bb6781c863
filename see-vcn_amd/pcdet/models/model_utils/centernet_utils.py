"""Top-K heat-map decoding (reference models/model_utils/centernet_utils.py:107-216)."""
import torch

from ...utils.loss_utils import _transpose_and_gather_feat


def _topk(scores, K=40):
    batch, num_class, height, width = scores.size()
    topk_scores, topk_inds = torch.topk(scores.flatten(2, 3), K)
    topk_inds = topk_inds % (height * width)
    topk_ys = (topk_inds // width).float()
    topk_xs = (topk_inds % width).int().float()
    topk_score, topk_ind = torch.topk(topk_scores.view(batch, -1), K)
    topk_classes = (topk_ind // K).int()
    g = lambda t: t.view(batch, -1).gather(1, topk_ind)
    return topk_score, g(topk_inds), topk_classes, g(topk_ys), g(topk_xs)


def decode_bbox_from_heatmap(heatmap, rot_cos, rot_sin, center, center_z, dim, point_cloud_range=None, voxel_size=None,
                             feature_map_stride=None, vel=None, K=100, circle_nms=False, score_thresh=None, post_center_limit_range=None):
    assert not circle_nms, 'circle_nms is dead code in the reference (assert False, centernet_utils.py:161)'
    batch_size = heatmap.size(0)
    scores, inds, class_ids, ys, xs = _topk(heatmap, K=K)
    center = _transpose_and_gather_feat(center, inds).view(batch_size, K, 2)
    rot_sin = _transpose_and_gather_feat(rot_sin, inds).view(batch_size, K, 1)
    rot_cos = _transpose_and_gather_feat(rot_cos, inds).view(batch_size, K, 1)
    center_z = _transpose_and_gather_feat(center_z, inds).view(batch_size, K, 1)
    dim = _transpose_and_gather_feat(dim, inds).view(batch_size, K, 3)
    angle = torch.atan2(rot_sin, rot_cos)
    xs = (xs.view(batch_size, K, 1) + center[:, :, 0:1]) * feature_map_stride * voxel_size[0] + point_cloud_range[0]
    ys = (ys.view(batch_size, K, 1) + center[:, :, 1:2]) * feature_map_stride * voxel_size[1] + point_cloud_range[1]
    parts = [xs, ys, center_z, dim, angle]
    if vel is not None:
        parts.append(_transpose_and_gather_feat(vel, inds).view(batch_size, K, 2))
    boxes = torch.cat(parts, dim=-1)
    final_scores, final_ids = scores.view(batch_size, K), class_ids.view(batch_size, K)
    assert post_center_limit_range is not None
    mask = (boxes[..., :3] >= post_center_limit_range[:3]).all(2) & (boxes[..., :3] <= post_center_limit_range[3:]).all(2)
    if score_thresh is not None:
        mask &= (final_scores > score_thresh)
    return [{'pred_boxes': boxes[k, mask[k]], 'pred_scores': final_scores[k, mask[k]], 'pred_labels': final_ids[k, mask[k]]}
            for k in range(batch_size)]
