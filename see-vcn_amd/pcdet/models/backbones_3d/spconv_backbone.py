from functools import partial

import torch
import torch.nn as nn

from ....spconv import chain, norm
from ...utils.spconv_utils import replace_feature, spconv


def post_act_block(in_channels, out_channels, kernel_size, indice_key=None, stride=1, padding=0,
                   conv_type='subm', norm_fn=None):
    """conv -> norm -> ReLU block (same composition as the reference, spconv_backbone.py:8-27)."""
    if conv_type == 'subm':
        conv = spconv.SubMConv3d(in_channels, out_channels, kernel_size, bias=False, indice_key=indice_key)
    elif conv_type == 'spconv':
        conv = spconv.SparseConv3d(in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                                   bias=False, indice_key=indice_key)
    elif conv_type == 'inverseconv':
        conv = spconv.SparseInverseConv3d(in_channels, out_channels, kernel_size, indice_key=indice_key, bias=False)
    else:
        raise NotImplementedError
    return spconv.SparseSequential(conv, norm_fn(out_channels), nn.ReLU())


class SparseBasicBlock(spconv.SparseModule):
    """Residual block of VoxelResBackBone8x (reference spconv_backbone.py:30-66)."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, norm_fn=None, downsample=None, indice_key=None):
        super().__init__()
        assert norm_fn is not None
        bias = norm_fn is not None
        self.conv1 = spconv.SubMConv3d(inplanes, planes, kernel_size=3, stride=stride, padding=1, bias=bias, indice_key=indice_key)
        self.bn1 = norm_fn(planes)
        self.relu = nn.ReLU()
        self.conv2 = spconv.SubMConv3d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=bias, indice_key=indice_key)
        self.bn2 = norm_fn(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x
        out = self.conv1(x)
        if norm.fusable(self.bn1, out.features):
            out = replace_feature(out, norm.batch_norm_relu(self.bn1, out.features, True))
        else:
            out = replace_feature(out, self.relu(self.bn1(out.features)))
        out = self.conv2(out)
        if norm.fusable(self.bn2, out.features):
            out = replace_feature(out, norm.batch_norm_relu(self.bn2, out.features, False))
        else:
            out = replace_feature(out, self.bn2(out.features))
        if self.downsample is not None:
            identity = self.downsample(x)
        out = replace_feature(out, self.relu(out.features + identity.features))
        return out


class _BackBone8xBase(nn.Module):
    def _finish(self, batch_dict, x_conv1, x_conv2, x_conv3, x_conv4, out):
        # same keys as the reference (spconv_backbone.py:159-178)
        batch_dict.update({'encoded_spconv_tensor': out, 'encoded_spconv_tensor_stride': 8})
        batch_dict.update({'multi_scale_3d_features': {'x_conv1': x_conv1, 'x_conv2': x_conv2, 'x_conv3': x_conv3, 'x_conv4': x_conv4}})
        batch_dict.update({'multi_scale_3d_strides': {'x_conv1': 1, 'x_conv2': 2, 'x_conv3': 4, 'x_conv4': 8}})
        return batch_dict

    def _chain_blocks(self):
        """The stages as one list of conv -> norm -> ReLU blocks (taps after conv1..conv4 and conv_out), or None; cached until a module is registered
        anywhere (the registration epoch of seevcn_amd.spconv.conv)."""
        from ....spconv import conv as sconv
        epoch = sconv._registration_epoch[0]
        hit = self.__dict__.get('_seevcn_chain')
        if hit is None or hit[0] != epoch:
            stages = [self.conv_input, self.conv1, self.conv2, self.conv3, self.conv4, self.conv_out]
            blocks = chain.flatten_blocks(stages)
            if blocks is not None:
                for b in blocks[:len(chain.flatten_blocks([self.conv_input]) or [])]:
                    b.tap = False                                     # conv_input feeds conv1 only
            hit = (epoch, blocks)
            self.__dict__['_seevcn_chain'] = hit
        return hit[1]

    def forward_stages(self, batch_dict):
        """forward() as a generator that yields between the backbone's stages (a caller with other work to enqueue in between, bench.py, steps
        through it); the value of the StopIteration is the batch_dict."""
        voxel_features, voxel_coords = batch_dict['voxel_features'], batch_dict['voxel_coords']
        input_sp_tensor = spconv.SparseConvTensor(features=voxel_features, indices=voxel_coords.int(),
                                                  spatial_shape=self.sparse_shape, batch_size=batch_dict['batch_size'],
                                                  indice_dict=batch_dict.get('spconv_indice_dict'))      # seevcn: rulebooks built ahead (pipeline.front)
        # all rulebooks + conv plans first: their host syncs then wait for index kernels only, and the layer loop below is enqueued without one
        spconv.prebuild_rulebooks(self, input_sp_tensor, with_backward=self.training and torch.is_grad_enabled())
        # the MFMA fragment copies of all layer weights in one launch (they follow the weights every forward; spconv/functional.py)
        spconv.refresh_weight_fragments(self)
        # training, every stage a plain run of conv -> BatchNorm1d -> ReLU blocks (VoxelBackBone8x): the whole chain and its backward as two launch
        # lists inside one autograd node (seevcn_amd/spconv/chain.py) instead of ~100 calls from the module tree
        blocks = self._chain_blocks()
        if self.training and chain.applicable(blocks, input_sp_tensor):
            x_conv1, x_conv2, x_conv3, x_conv4, out = chain.run_chain(blocks, input_sp_tensor)
            return self._finish(batch_dict, x_conv1, x_conv2, x_conv3, x_conv4, out)
        x = self.conv_input(input_sp_tensor)
        x_conv1 = self.conv1(x)
        yield
        x_conv2 = self.conv2(x_conv1)
        yield
        x_conv3 = self.conv3(x_conv2)
        yield
        x_conv4 = self.conv4(x_conv3)
        out = self.conv_out(x_conv4)
        return self._finish(batch_dict, x_conv1, x_conv2, x_conv3, x_conv4, out)

    def forward(self, batch_dict):
        g = self.forward_stages(batch_dict)
        try:
            while True:
                next(g)
        except StopIteration as done:
            return done.value


class VoxelBackBone8x(_BackBone8xBase):
    """Drop-in for the reference VoxelBackBone8x (backbones_3d/spconv_backbone.py:69-180): same constructor
    keywords, submodule names (state_dict keys), indice_keys and batch_dict contract; the sparse ops run on
    libseevcn_hip.so through seevcn_amd.spconv."""

    def __init__(self, model_cfg, input_channels, grid_size, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        norm_fn = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        self.sparse_shape = [int(g) for g in list(grid_size)[::-1]]
        self.sparse_shape[0] += 1                     # grid_size[::-1] + [1, 0, 0]  (spconv_backbone.py:75)
        self.conv_input = spconv.SparseSequential(
            spconv.SubMConv3d(input_channels, 16, 3, padding=1, bias=False, indice_key='subm1'), norm_fn(16), nn.ReLU())
        block = post_act_block
        self.conv1 = spconv.SparseSequential(block(16, 16, 3, norm_fn=norm_fn, padding=1, indice_key='subm1'))
        self.conv2 = spconv.SparseSequential(
            block(16, 32, 3, norm_fn=norm_fn, stride=2, padding=1, indice_key='spconv2', conv_type='spconv'),
            block(32, 32, 3, norm_fn=norm_fn, padding=1, indice_key='subm2'),
            block(32, 32, 3, norm_fn=norm_fn, padding=1, indice_key='subm2'))
        self.conv3 = spconv.SparseSequential(
            block(32, 64, 3, norm_fn=norm_fn, stride=2, padding=1, indice_key='spconv3', conv_type='spconv'),
            block(64, 64, 3, norm_fn=norm_fn, padding=1, indice_key='subm3'),
            block(64, 64, 3, norm_fn=norm_fn, padding=1, indice_key='subm3'))
        self.conv4 = spconv.SparseSequential(
            block(64, 64, 3, norm_fn=norm_fn, stride=2, padding=(0, 1, 1), indice_key='spconv4', conv_type='spconv'),
            block(64, 64, 3, norm_fn=norm_fn, padding=1, indice_key='subm4'),
            block(64, 64, 3, norm_fn=norm_fn, padding=1, indice_key='subm4'))
        last_pad = self.model_cfg.get('last_pad', 0) if hasattr(self.model_cfg, 'get') else 0
        self.conv_out = spconv.SparseSequential(
            spconv.SparseConv3d(64, 128, (3, 1, 1), stride=(2, 1, 1), padding=last_pad, bias=False, indice_key='spconv_down2'),
            norm_fn(128), nn.ReLU())
        self.num_point_features = 128
        self.backbone_channels = {'x_conv1': 16, 'x_conv2': 32, 'x_conv3': 64, 'x_conv4': 64}


class VoxelResBackBone8x(_BackBone8xBase):
    """Drop-in for the reference VoxelResBackBone8x (spconv_backbone.py:183-293)."""

    def __init__(self, model_cfg, input_channels, grid_size, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        norm_fn = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        self.sparse_shape = [int(g) for g in list(grid_size)[::-1]]
        self.sparse_shape[0] += 1
        self.conv_input = spconv.SparseSequential(
            spconv.SubMConv3d(input_channels, 16, 3, padding=1, bias=False, indice_key='subm1'), norm_fn(16), nn.ReLU())
        block = post_act_block
        self.conv1 = spconv.SparseSequential(
            SparseBasicBlock(16, 16, norm_fn=norm_fn, indice_key='res1'),
            SparseBasicBlock(16, 16, norm_fn=norm_fn, indice_key='res1'))
        self.conv2 = spconv.SparseSequential(
            block(16, 32, 3, norm_fn=norm_fn, stride=2, padding=1, indice_key='spconv2', conv_type='spconv'),
            SparseBasicBlock(32, 32, norm_fn=norm_fn, indice_key='res2'),
            SparseBasicBlock(32, 32, norm_fn=norm_fn, indice_key='res2'))
        self.conv3 = spconv.SparseSequential(
            block(32, 64, 3, norm_fn=norm_fn, stride=2, padding=1, indice_key='spconv3', conv_type='spconv'),
            SparseBasicBlock(64, 64, norm_fn=norm_fn, indice_key='res3'),
            SparseBasicBlock(64, 64, norm_fn=norm_fn, indice_key='res3'))
        self.conv4 = spconv.SparseSequential(
            block(64, 128, 3, norm_fn=norm_fn, stride=2, padding=(0, 1, 1), indice_key='spconv4', conv_type='spconv'),
            SparseBasicBlock(128, 128, norm_fn=norm_fn, indice_key='res4'),
            SparseBasicBlock(128, 128, norm_fn=norm_fn, indice_key='res4'))
        last_pad = self.model_cfg.get('last_pad', 0) if hasattr(self.model_cfg, 'get') else 0
        self.conv_out = spconv.SparseSequential(
            spconv.SparseConv3d(128, 128, (3, 1, 1), stride=(2, 1, 1), padding=last_pad, bias=False, indice_key='spconv_down2'),
            norm_fn(128), nn.ReLU())
        self.num_point_features = 128
        self.backbone_channels = {'x_conv1': 16, 'x_conv2': 32, 'x_conv3': 64, 'x_conv4': 128}
