"""Batch assembly and host->device hand-off, same semantics as the reference's
DatasetTemplate.collate_batch (detector3d/pcdet/datasets/dataset.py:175-257) and load_data_to_gpu / model_fn_decorator
(detector3d/pcdet/models/__init__.py:23-52).  Host glue: numpy in, numpy out; the device copy is one `torch.from_numpy(...).to()`
per array on the current stream (pinned when the caller hands pinned arrays)."""
from collections import defaultdict, namedtuple

import numpy as np
import torch


def _pad_params(desired_size, cur_size):
    """common_utils.get_pad_params (utils/common_utils.py:120-135): all padding goes after the data."""
    assert desired_size >= cur_size
    return 0, desired_size - cur_size


def collate_batch(batch_list, _unused=False):
    data_dict = defaultdict(list)
    for cur_sample in batch_list:
        for key, val in cur_sample.items():
            data_dict[key].append(val)
    batch_size = len(batch_list)
    ret = {}
    for key, val in data_dict.items():
        try:
            if key in ['voxels', 'voxel_num_points']:
                ret[key] = np.concatenate(val, axis=0)
            elif key in ['points', 'voxel_coords']:
                # prepend the sample index as column 0: points (P, 1+C) [b,x,y,z,...], voxel_coords (V,4) [b,z,y,x]
                ret[key] = np.concatenate([np.pad(c, ((0, 0), (1, 0)), mode='constant', constant_values=i) for i, c in enumerate(val)], axis=0)
                if key == 'points':
                    # seevcn extension: the per-scene counts are known here for free; with them on the host the detector needs no device -> host
                    # read to size its keypoint sampling (VoxelSetAbstraction.prefetch_keypoints)
                    ret['points_per_scene'] = [len(c) for c in val]
            elif key in ['gt_boxes', 'gt_boxes2d']:
                max_gt = max(len(x) for x in val)
                out = np.zeros((batch_size, max_gt, val[0].shape[-1]), dtype=np.float32)
                for k in range(batch_size):
                    if val[k].size > 0:
                        out[k, :len(val[k]), :] = val[k]
                ret[key] = out
            elif key in ['images', 'depth_maps']:
                max_h = max(im.shape[0] for im in val)
                max_w = max(im.shape[1] for im in val)
                images = []
                for im in val:
                    pad = (_pad_params(max_h, im.shape[0]), _pad_params(max_w, im.shape[1]))
                    images.append(np.pad(im, pad_width=pad + ((0, 0),) if key == 'images' else pad, mode='constant', constant_values=0))
                ret[key] = np.stack(images, axis=0)
            elif key in ['calib']:
                ret[key] = val
            elif key in ['points_2d']:
                max_len = max(len(v) for v in val)
                ret[key] = np.stack([np.pad(v, ((0, max_len - len(v)), (0, 0)), mode='constant', constant_values=0) for v in val], axis=0)
            else:
                ret[key] = np.stack(val, axis=0)
        except Exception as e:
            print('Error in collate_batch: key=%s' % key)
            print('e: ', e)
            raise TypeError
    ret['batch_size'] = batch_size
    return ret


def load_data_to_gpu(batch_dict, device=None):
    """In place: every ndarray except frame_id / metadata / calib becomes a float32 CUDA tensor (image_shape int32;
    images HWC -> CHW like kornia.image_to_tensor)."""
    device = torch.device('cuda', torch.cuda.current_device()) if device is None else device
    for key, val in batch_dict.items():
        if not isinstance(val, np.ndarray):
            continue
        elif key in ['frame_id', 'metadata', 'calib']:
            continue
        elif key in ['images']:
            t = torch.from_numpy(val)
            batch_dict[key] = t.permute(0, 3, 1, 2).float().to(device).contiguous() if t.dim() == 4 else t.float().to(device).contiguous()
        elif key in ['image_shape']:
            batch_dict[key] = torch.from_numpy(val).int().to(device)
        else:
            batch_dict[key] = torch.from_numpy(val).float().to(device)


def model_fn_decorator():
    ModelReturn = namedtuple('ModelReturn', ['loss', 'tb_dict', 'disp_dict'])

    def model_func(model, batch_dict):
        load_data_to_gpu(batch_dict)
        ret_dict, tb_dict, disp_dict = model(batch_dict)
        loss = ret_dict['loss'].mean()
        if hasattr(model, 'update_global_step'):
            model.update_global_step()
        else:
            model.module.update_global_step()
        return ModelReturn(loss, tb_dict, disp_dict)

    return model_func
