import copy
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.init import kaiming_normal_

from .... import _lib
from ...utils import common_utils, loss_utils
from ...utils.common_utils import cfg_get
from ..model_utils import centernet_utils, model_nms_utils


class SeparateHead(nn.Module):
    """One 3x3-conv branch per regression target + heat map; branch names become attributes so the state_dict keys are the
    reference's (`heads_list.0.center.0.0.weight`, ...; center_head.py:11-45)."""

    def __init__(self, input_channels, sep_head_dict, init_bias=-2.19, use_bias=False):
        super().__init__()
        self.sep_head_dict = sep_head_dict
        for name in self.sep_head_dict:
            out_ch, num_conv = self.sep_head_dict[name]['out_channels'], self.sep_head_dict[name]['num_conv']
            layers = [nn.Sequential(nn.Conv2d(input_channels, input_channels, kernel_size=3, stride=1, padding=1, bias=use_bias),
                                    nn.BatchNorm2d(input_channels), nn.ReLU()) for _ in range(num_conv - 1)]
            layers.append(nn.Conv2d(input_channels, out_ch, kernel_size=3, stride=1, padding=1, bias=True))
            fc = nn.Sequential(*layers)
            if 'hm' in name:
                fc[-1].bias.data.fill_(init_bias)
            else:
                for m in fc.modules():
                    if isinstance(m, nn.Conv2d):
                        kaiming_normal_(m.weight.data)
                        if getattr(m, 'bias', None) is not None:
                            nn.init.constant_(m.bias, 0)
            self.__setattr__(name, fc)

    def forward(self, x):
        return {name: self.__getattr__(name)(x) for name in self.sep_head_dict}


# All branches of all heads as three convolutions instead of 6 x 6 x 2 (seevcn: same parameters, same state_dict, same sums in another order).  One
# 180 x 180 nuScenes map: 1 997 -> 1 455 launches per CenterPoint train step, step 23.6 -> 23.2 ms; GPU time is unchanged (the block-diagonal last
# convolution does 36x the needed arithmetic, which eats what the fewer BatchNorm / transpose / weight-gradient launches save -- in isolation the
# merged form looks 2x faster only because a chain of small launches on an idle GPU is latency; profiles/r05_head_conv_probe.txt).
MERGE_BRANCHES = os.environ.get("SEEVCN_CENTERHEAD_MERGED", "1") != "0"


def _plain(m):
    return not (m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or m._backward_pre_hooks)


def merged_branches(heads, x):
    """[head(x) for head in heads] with every branch's first conv + BatchNorm + ReLU run as ONE convolution / ONE BatchNorm over the concatenated
    channels and every last conv as ONE block-diagonal convolution.  Returns None when the branches are not all `conv-BN-ReLU, conv` of one width
    on plain modules (the caller then runs them one by one, as center_head.py:43-45 does)."""
    branches = [(hi, name, getattr(head, name)) for hi, head in enumerate(heads) for name in head.sep_head_dict]
    if len(branches) < 2 or not x.is_cuda:
        return None
    if not all(_plain(head) and type(head).forward is SeparateHead.forward for head in heads):
        return None                                              # the merged form never calls head.forward: hooks on a head, or a subclass's forward, need the modules
    c = x.shape[1]
    for _, _, fc in branches:
        if len(fc) != 2 or not isinstance(fc[0], nn.Sequential) or len(fc[0]) != 3:
            return None
        conv1, bn, act, conv2 = fc[0][0], fc[0][1], fc[0][2], fc[1]
        # exact types (as spconv/chain.py checks its blocks): a subclass -- a frozen BatchNorm with its own forward, say -- must run as itself
        ok = (type(conv1) is nn.Conv2d and type(bn) is nn.BatchNorm2d and type(act) is nn.ReLU and type(conv2) is nn.Conv2d
              and bn.momentum is not None                        # momentum None = cumulative average over num_batches_tracked: F.batch_norm has no such mode
              and conv1.in_channels == c and conv1.out_channels == c and conv2.in_channels == c and conv2.bias is not None
              and all(m.kernel_size == (3, 3) and m.stride == (1, 1) and m.padding == (1, 1) and m.dilation == (1, 1) and m.groups == 1 for m in (conv1, conv2))
              and (conv1.bias is None) == (branches[0][2][0][0].bias is None)
              and bn.affine and bn.track_running_stats and bn.momentum == branches[0][2][0][1].momentum and bn.eps == branches[0][2][0][1].eps
              and bn.training == branches[0][2][0][1].training and all(_plain(m) for m in (fc, fc[0], conv1, bn, act, conv2)))
        if not ok:
            return None
    nb = len(branches)
    conv1s, bns, conv2s = [b[2][0][0] for b in branches], [b[2][0][1] for b in branches], [b[2][1] for b in branches]
    # first layers: (nb c, c, 3, 3)
    y = F.conv2d(x, torch.cat([m.weight for m in conv1s], 0), None if conv1s[0].bias is None else torch.cat([m.bias for m in conv1s], 0), padding=1)
    bn0 = bns[0]
    gamma, beta = torch.cat([m.weight for m in bns], 0), torch.cat([m.bias for m in bns], 0)
    mean, var = torch.cat([m.running_mean for m in bns], 0), torch.cat([m.running_var for m in bns], 0)
    y = F.relu(F.batch_norm(y, mean, var, gamma, beta, bn0.training, bn0.momentum, bn0.eps))
    if bn0.training:
        with torch.no_grad():                                    # the running statistics back into the modules that own them
            torch._foreach_copy_([m.running_mean for m in bns], list(mean.split(c)))
            torch._foreach_copy_([m.running_var for m in bns], list(var.split(c)))
            torch._foreach_add_([m.num_batches_tracked for m in bns], 1)
    # last layers: branch b reads channels [b c, (b + 1) c) only -- a (nb kmax, nb c, 3, 3) weight that is zero off the diagonal blocks
    kmax = max(m.out_channels for m in conv2s)
    wp = torch.stack([F.pad(m.weight, (0, 0, 0, 0, 0, 0, 0, kmax - m.out_channels)) for m in conv2s], 0)          # (nb, kmax, c, 3, 3)
    eye = torch.eye(nb, device=x.device, dtype=x.dtype)
    wd = (wp[:, :, None] * eye[:, None, :, None, None, None]).reshape(nb * kmax, nb * c, 3, 3)
    bp = torch.cat([F.pad(m.bias, (0, kmax - m.out_channels)) for m in conv2s], 0)
    z = F.conv2d(y, wd, bp, padding=1)
    out = [dict() for _ in heads]
    for b, (hi, name, _) in enumerate(branches):
        out[hi][name] = z[:, b * kmax:b * kmax + conv2s[b].out_channels]
    return out


class CenterHead(nn.Module):
    """Drop-in for the reference CenterHead (dense_heads/center_head.py:48-355).  Heat-map / regression targets for all heads
    and scenes are produced by one HIP launch (sv_center_assign_targets) instead of a python loop on CPU tensors."""

    def __init__(self, model_cfg, input_channels, num_class, class_names, grid_size, point_cloud_range, voxel_size,
                 predict_boxes_when_training=True):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_class = num_class
        self.grid_size = grid_size
        self.point_cloud_range = [float(v) for v in point_cloud_range]
        self.voxel_size = [float(v) for v in voxel_size]
        tcfg = cfg_get(model_cfg, 'TARGET_ASSIGNER_CONFIG')
        self.feature_map_stride = cfg_get(tcfg, 'FEATURE_MAP_STRIDE', None)
        self.class_names = list(class_names)
        self.class_names_each_head = [[x for x in names if x in class_names] for names in cfg_get(model_cfg, 'CLASS_NAMES_EACH_HEAD')]
        self.class_id_mapping_each_head = [torch.tensor([self.class_names.index(x) for x in names], dtype=torch.long)
                                           for names in self.class_names_each_head]
        assert sum(len(x) for x in self.class_names_each_head) == len(self.class_names), f'class_names_each_head={self.class_names_each_head}'
        use_bias = cfg_get(model_cfg, 'USE_BIAS_BEFORE_NORM', False)
        ch = cfg_get(model_cfg, 'SHARED_CONV_CHANNEL')
        self.shared_conv = nn.Sequential(nn.Conv2d(input_channels, ch, 3, stride=1, padding=1, bias=use_bias), nn.BatchNorm2d(ch), nn.ReLU())
        self.heads_list = nn.ModuleList()
        self.separate_head_cfg = cfg_get(model_cfg, 'SEPARATE_HEAD_CFG')
        for names in self.class_names_each_head:
            head_dict = copy.deepcopy(dict(cfg_get(self.separate_head_cfg, 'HEAD_DICT')))
            head_dict['hm'] = dict(out_channels=len(names), num_conv=cfg_get(model_cfg, 'NUM_HM_CONV'))
            self.heads_list.append(SeparateHead(input_channels=ch, sep_head_dict=head_dict, init_bias=-2.19, use_bias=use_bias))
        self.predict_boxes_when_training = predict_boxes_when_training
        self.forward_ret_dict = {}
        self.add_module('hm_loss_func', loss_utils.FocalLossCenterNet())
        self.add_module('reg_loss_func', loss_utils.RegLossCenterNet())
        self._tables = {}

    def _device_tables(self, dev):
        if dev not in self._tables:
            nh, nc = len(self.class_names_each_head), len(self.class_names)
            tab = np.zeros((nh, nc + 1), np.int32)
            for h, names in enumerate(self.class_names_each_head):
                for li, n in enumerate(names):
                    tab[h, self.class_names.index(n) + 1] = li + 1
            ncls = np.array([len(n) for n in self.class_names_each_head], np.int32)
            off = (np.cumsum(ncls) - ncls).astype(np.int32)
            self._tables[dev] = tuple(torch.from_numpy(x).to(dev) for x in (tab, ncls, off)) + (int(ncls.sum()), off.tolist())
        return self._tables[dev]

    def assign_targets(self, gt_boxes, feature_map_size=None, **kwargs):
        """gt_boxes (B,M,8+) -> dict of per-head lists: heatmaps (B,ncls,H,W), target_boxes (B,500,dim), inds, masks."""
        lib = _lib.load()
        _lib.require_cuda(gt_boxes)
        H, W = int(feature_map_size[0]), int(feature_map_size[1])
        tcfg = cfg_get(self.model_cfg, 'TARGET_ASSIGNER_CONFIG')
        gt = gt_boxes.contiguous().float()
        B, G, D = gt.shape
        dev = gt.device
        tab, ncls, off, total, off_host = self._device_tables(dev)
        nh, nmax = len(self.class_names_each_head), cfg_get(tcfg, 'NUM_MAX_OBJS')
        heat = torch.empty((B, total, H, W), dtype=torch.float32, device=dev)
        tbox = torch.empty((nh, B, nmax, D), dtype=torch.float32, device=dev)
        inds = torch.empty((nh, B, nmax), dtype=torch.int64, device=dev)
        masks = torch.empty((nh, B, nmax), dtype=torch.int64, device=dev)
        rc = lib.sv_center_assign_targets(_lib.ptr(gt) if G else None, B, G, D, nh, len(self.class_names), _lib.ptr(tab), _lib.ptr(ncls), _lib.ptr(off),
                                          total, W, H, self.point_cloud_range[0], self.point_cloud_range[1], self.voxel_size[0], self.voxel_size[1],
                                          float(cfg_get(tcfg, 'FEATURE_MAP_STRIDE')), nmax, float(cfg_get(tcfg, 'GAUSSIAN_OVERLAP')),
                                          int(cfg_get(tcfg, 'MIN_RADIUS')), _lib.ptr(heat), _lib.ptr(tbox), _lib.ptr(inds), _lib.ptr(masks), _lib.stream())
        _lib.check(rc, "sv_center_assign_targets")
        offs = off_host + [total]
        return {'heatmaps': [heat[:, offs[h]:offs[h + 1]] for h in range(nh)], 'target_boxes': [tbox[h] for h in range(nh)],
                'inds': [inds[h] for h in range(nh)], 'masks': [masks[h] for h in range(nh)], 'heatmap_masks': []}

    @staticmethod
    def sigmoid(x):
        return torch.clamp(x.sigmoid(), min=1e-4, max=1 - 1e-4)

    def get_loss(self):
        pred_dicts, target_dicts = self.forward_ret_dict['pred_dicts'], self.forward_ret_dict['target_dicts']
        lw = cfg_get(self.model_cfg, 'LOSS_CONFIG')['LOSS_WEIGHTS']
        tb_dict, loss = {}, 0
        for idx, pred in enumerate(pred_dicts):
            pred['hm'] = self.sigmoid(pred['hm'])
            hm_loss = self.hm_loss_func(pred['hm'], target_dicts['heatmaps'][idx]) * lw['cls_weight']
            pred_boxes = torch.cat([pred[name] for name in cfg_get(self.separate_head_cfg, 'HEAD_ORDER')], dim=1)
            reg_loss = self.reg_loss_func(pred_boxes, target_dicts['masks'][idx], target_dicts['inds'][idx], target_dicts['target_boxes'][idx])
            loc_loss = (reg_loss * common_utils.const_tensor(lw['code_weights'], reg_loss.device, reg_loss.dtype)).sum() * lw['loc_weight']
            loss = loss + hm_loss + loc_loss
            tb_dict['hm_loss_head_%d' % idx] = common_utils.tb_value(hm_loss)
            tb_dict['loc_loss_head_%d' % idx] = common_utils.tb_value(loc_loss)
        tb_dict['rpn_loss'] = common_utils.tb_value(loss)
        return loss, tb_dict

    @torch.no_grad()
    def generate_predicted_boxes(self, batch_size, pred_dicts):
        pp = cfg_get(self.model_cfg, 'POST_PROCESSING')
        nms_cfg = cfg_get(pp, 'NMS_CONFIG')
        dev = pred_dicts[0]['hm'].device
        limit = torch.tensor(cfg_get(pp, 'POST_CENTER_LIMIT_RANGE'), device=dev).float()
        ret = [{'pred_boxes': [], 'pred_scores': [], 'pred_labels': []} for _ in range(batch_size)]
        order = cfg_get(self.separate_head_cfg, 'HEAD_ORDER')
        for idx, pred in enumerate(pred_dicts):
            finals = centernet_utils.decode_bbox_from_heatmap(
                heatmap=pred['hm'].sigmoid(), rot_cos=pred['rot'][:, 0].unsqueeze(1), rot_sin=pred['rot'][:, 1].unsqueeze(1), center=pred['center'],
                center_z=pred['center_z'], dim=pred['dim'].exp(), vel=pred['vel'] if 'vel' in order else None,
                point_cloud_range=self.point_cloud_range, voxel_size=self.voxel_size, feature_map_stride=self.feature_map_stride,
                K=cfg_get(pp, 'MAX_OBJ_PER_SAMPLE'), circle_nms=(cfg_get(nms_cfg, 'NMS_TYPE') == 'circle_nms'), score_thresh=cfg_get(pp, 'SCORE_THRESH'),
                post_center_limit_range=limit)
            mapping = self.class_id_mapping_each_head[idx].to(dev)
            for k, fd in enumerate(finals):
                fd['pred_labels'] = mapping[fd['pred_labels'].long()]
                if cfg_get(nms_cfg, 'NMS_TYPE') != 'circle_nms':
                    selected, selected_scores = model_nms_utils.class_agnostic_nms(box_scores=fd['pred_scores'], box_preds=fd['pred_boxes'],
                                                                                   nms_config=nms_cfg, score_thresh=None)
                    fd['pred_boxes'], fd['pred_scores'], fd['pred_labels'] = fd['pred_boxes'][selected], selected_scores, fd['pred_labels'][selected]
                for key in ('pred_boxes', 'pred_scores', 'pred_labels'):
                    ret[k][key].append(fd[key])
        for k in range(batch_size):
            ret[k]['pred_boxes'] = torch.cat(ret[k]['pred_boxes'], dim=0)
            ret[k]['pred_scores'] = torch.cat(ret[k]['pred_scores'], dim=0)
            ret[k]['pred_labels'] = torch.cat(ret[k]['pred_labels'], dim=0) + 1
        return ret

    @staticmethod
    def reorder_rois_for_refining(batch_size, pred_dicts):
        n = max(1, max(len(d['pred_boxes']) for d in pred_dicts))
        pb = pred_dicts[0]['pred_boxes']
        rois, scores, labels = pb.new_zeros((batch_size, n, pb.shape[-1])), pb.new_zeros((batch_size, n)), pb.new_zeros((batch_size, n)).long()
        for b in range(batch_size):
            m = len(pred_dicts[b]['pred_boxes'])
            rois[b, :m], scores[b, :m], labels[b, :m] = pred_dicts[b]['pred_boxes'], pred_dicts[b]['pred_scores'], pred_dicts[b]['pred_labels']
        return rois, scores, labels

    def forward(self, data_dict):
        x = self.shared_conv(data_dict['spatial_features_2d'])
        pred_dicts = merged_branches(self.heads_list, x) if MERGE_BRANCHES else None
        if pred_dicts is None:
            pred_dicts = [head(x) for head in self.heads_list]
        if self.training:
            self.forward_ret_dict['target_dicts'] = self.assign_targets(data_dict['gt_boxes'], feature_map_size=data_dict['spatial_features_2d'].size()[2:])
        self.forward_ret_dict['pred_dicts'] = pred_dicts
        if not self.training or self.predict_boxes_when_training:
            preds = self.generate_predicted_boxes(data_dict['batch_size'], pred_dicts)
            if self.predict_boxes_when_training:
                data_dict['rois'], data_dict['roi_scores'], data_dict['roi_labels'] = self.reorder_rois_for_refining(data_dict['batch_size'], preds)
                data_dict['has_class_labels'] = True
            else:
                data_dict['final_box_dicts'] = preds
        return data_dict
