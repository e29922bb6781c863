"""The measured hot path as one object: VCN completion of cropped objects (network + surface selection + largest
cluster) -> merge into the scenes (unique + replace the original object points) ->
dynamic voxelisation -> VoxelBackBone8x -> HeightCompression (forward + backward + optimiser step).

This is the composition BASELINE.json's metric names ("VCN + voxel + spconv fwd+bwd"); it only chains modules that
mirror the reference's own classes (VCN_VC, DynMeanVFE, VoxelBackBone8x, HeightCompression) the way
SEE_VCN.complete_*_pts / replace_with_completed_pts (see/surface_completion/SEE_VCN.py:85-115,247-265) and
SECONDNet.forward (detector3d/pcdet/models/detectors/second_net.py:9-22) chain them.
"""
import torch
import torch.nn as nn

from .pcdet.models.backbones_2d import map_to_bev
from .pcdet.models import backbones_3d
from .pcdet.models.backbones_3d import vfe
from . import _lib
from .vcn import MODELS
from .vcn.scene_merge import complete_scene_batch_device
from .vcn.utils.sampling import get_largest_cluster_batch_device, get_partial_mesh_batch_device

KITTI = dict(point_cloud_range=[0, -40, -3, 70.4, 40, 1], voxel_size=[0.05, 0.05, 0.1], grid_size=[1408, 1600, 40])


class SceneStep(nn.Module):
    def __init__(self, geometry=None, input_channels=3, post_process=True, sel_k=30, cluster_eps=0.4):
        super().__init__()
        self.post_process, self.sel_k, self.cluster_eps = post_process, sel_k, cluster_eps
        g = dict(KITTI if geometry is None else geometry)
        self.geometry = g
        self.vcn = MODELS.build({'NAME': 'VCN_VC'}).eval()
        for p in self.vcn.parameters():
            p.requires_grad_(False)
        self.vfe = vfe.__all__['DynMeanVFE'](model_cfg={}, num_point_features=input_channels, voxel_size=g['voxel_size'],
                                             grid_size=g['grid_size'], point_cloud_range=g['point_cloud_range'])
        self.backbone_3d = backbones_3d.__all__['VoxelBackBone8x']({}, input_channels, g['grid_size'])
        self.map_to_bev = map_to_bev.__all__['HeightCompression']({'NUM_BEV_FEATURES': 256})

    def train(self, mode=True):
        super().train(mode)
        self.vcn.eval()   # VCN weights are given (inference only in the SEE pipeline)
        return self

    def complete_and_paste(self, points, objects, object_scene):
        """points (ΣP,4) [b,x,y,z]; objects (B_o,1024,3); object_scene (B_o,) float scene index of each object."""
        coarse = self.vcn({'input': objects})['coarse']                          # (B_o,1024,3)
        if not self.post_process:
            bcol = object_scene.view(-1, 1, 1).expand(-1, coarse.shape[1], 1)
            return torch.cat([points, torch.cat([bcol, coarse], dim=2).view(-1, 4)], dim=0)
        # VCN.inference's post-processing (models/VCN.py:89-93) and the scene merge (SEE_VCN.py:115,247-265), all on the GPU
        surface, n_sel = get_partial_mesh_batch_device(objects, coarse, k=self.sel_k)
        # the empty-cluster check rides on the voxeliser's read below instead of blocking here (one host <-> device round trip less per step)
        clustered, _ = get_largest_cluster_batch_device(surface, eps=self.cluster_eps, min_points=2, total_pts=coarse.shape[1], defer_check=True, period=n_sel)
        return complete_scene_batch_device(points, clustered, object_scene, 0.1, compact=False)   # replaced points: scene id -1, dropped by the VFE

    def front_a(self, points, objects, object_scene):
        """First half of the input side: stage A (frozen VCN, surface selection, largest cluster) and the merge into the scenes -> the pasted
        point rows (SP', 4).  No device -> host read (the empty-cluster check is parked on the stream, _lib.defer_check)."""
        with torch.no_grad():
            return self.complete_and_paste(points, objects, object_scene)

    def front_b(self, pts, batch_size, flush=True):
        """Second half: dynamic voxelisation + mean VFE and every rulebook / convolution plan of the backbone, with the input side's ONE
        device -> host read (voxel count + the strided levels' site counts; checks parked on this stream ride on it)."""
        from . import spconv
        with torch.no_grad():
            # the voxel count stays on the device until the strided levels of the backbone have been counted there too
            # (spconv.prebuild_rulebooks -> Fsp.build_network_index)
            bd = self.vfe({'batch_size': batch_size, 'points': pts}, lazy_count=True)
            sp = spconv.SparseConvTensor(features=bd['voxel_features'], indices=bd['voxel_coords'], spatial_shape=self.backbone_3d.sparse_shape,
                                         batch_size=batch_size)
            spconv.prebuild_rulebooks(self.backbone_3d, sp, with_backward=self.training, n0_dev=bd.pop('voxel_count_device'))
            if flush:
                _lib.flush_checks()                               # nothing parked survives the front (normally taken by the one read above)
            bd['voxel_features'] = sp.features
            bd['voxel_coords'] = sp.indices                       # the very tensor the rulebooks are bound to
            bd['spconv_indice_dict'] = sp.indice_dict
        return bd

    def front(self, points, objects, object_scene, batch_size):
        """The INPUT side of the step -- everything that depends on the scene batch only, not on the trained weights: stage A (frozen VCN,
        surface selection, cluster, merge), dynamic voxelisation + mean VFE, and every rulebook / convolution plan of the backbone.  No
        gradients.  This is what the reference does ahead of the training step (SEE-VCN writes completed clouds offline, SEE_VCN.py:85-115;
        pcdet voxelises in dataloader workers, dataset.py:126-160); here it may run on a side stream for later batches while batch N trains
        (bench.py: front_b of batch N + 1 and front_a of batch N + 2), and its one device -> host read then waits for index kernels only."""
        return self.front_b(self.front_a(points, objects, object_scene), batch_size)

    def compute(self, bd):
        """The trained side: VoxelBackBone8x over the prebuilt rulebooks -> HeightCompression (autograd graph starts here)."""
        bd = self.backbone_3d(bd)
        bd = self.map_to_bev(bd)
        return bd

    def compute_stages(self, bd):
        """compute() as a generator yielding between the backbone's stages (see _BackBone8xBase.forward_stages)."""
        bd = yield from self.backbone_3d.forward_stages(bd)
        return self.map_to_bev(bd)

    def forward(self, points, objects, object_scene, batch_size):
        return self.compute(self.front(points, objects, object_scene, batch_size))


def record_stream_tree(obj, stream, _seen=None):
    """tensor.record_stream(stream) on every CUDA tensor reachable from obj (dicts, sequences, SparseConvTensor / Rulebook / TablePlan
    objects): tensors produced on a side stream and consumed on `stream` must tell the caching allocator so, or a later side-stream allocation
    may reuse their memory while `stream` still reads it."""
    seen = set() if _seen is None else _seen
    if id(obj) in seen:
        return
    seen.add(id(obj))
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, dict):
        for v in obj.values():
            record_stream_tree(v, stream, seen)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            record_stream_tree(v, stream, seen)
    elif hasattr(obj, '__dict__') and type(obj).__module__.startswith(__name__.split('.')[0]):
        record_stream_tree(vars(obj), stream, seen)
