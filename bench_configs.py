"""Side modes of bench.py: `python bench.py --config {stageA,second,pvrcnn,centerpoint}` -- the other BASELINE.json configs as timed
workloads of their own (the default, no --config, stays the headline VCN + voxel + spconv step).  Same protocol as the main mode: W untimed
warm-up steps, K timed steps between synchronisations (barrier + max over ranks for N > 1), one JSON line on rank 0.  No cpu_baseline and
no live kernel timing here: the per-kernel tables of these modes come from `tools/profile.sh` runs of the same command (profiles/).

  stageA       configs[1]  VCN_VC forward + surface selection (kNN, k = 30) + largest DBSCAN cluster, 64 objects x 1024 points, inference
  second       configs[2]  SECONDNet (DynMeanVFE, VoxelBackBone8x, HeightCompression, BaseBEVBackbone, AnchorHeadSingle) train step,
                           16 KITTI-shaped scenes per GPU
  pvrcnn       configs[3]  SEE-VCN PV-RCNN train step in the domain-adaptation geometry ([41,1504,1504], 4096 keypoints), 4 scenes per GPU
                           (bs = 32 over 8 GPUs)
  centerpoint  configs[4]  CenterPoint (VoxelResBackBone8x + CenterHead) train step on one ~300 k-point nuScenes-shaped scene per GPU
                           (bs = 8 over 8 GPUs), hard voxelisation (10 points / voxel, 120 k voxels) inside the step
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = ("stageA", "second", "pvrcnn", "centerpoint")


def _detector(cfg, num_class, ds, device, seed):
    from seevcn_amd.seeding import seeded_state_dict
    from seevcn_amd.pcdet.models import detectors
    net = detectors.build_detector(cfg, num_class=num_class, dataset=ds)
    net.load_state_dict(seeded_state_dict(net, seed=seed))
    return net.to(device).train()


def _train_step(net, batch, opt, params, world, allreduce):
    opt.zero_grad(set_to_none=True)
    ret, _, _ = net(dict(batch))
    ret["loss"].backward()
    allreduce(params, world)
    opt.step()
    return ret["loss"]


def build(config, rank, device, scenes=None):
    """-> (step() callable, units per step on this rank, unit name, metric name, config dict)"""
    # SEEVCN_MIOPEN_FIND=1: torch.backends.cudnn.benchmark -- MIOpen then times its applicable solvers for every new convolution shape (in the warm-up
    # steps) instead of taking the heuristic pick of immediate mode.  A runtime switch of the library the dense 2-D part stays on, not a code path here.
    if os.environ.get("SEEVCN_MIOPEN_FIND") is not None:
        torch.backends.cudnn.benchmark = os.environ["SEEVCN_MIOPEN_FIND"] == "1"
    import seevcn_amd.synth as synth
    from seevcn_amd.pcdet import model_cfgs as C
    if config == "stageA":
        import seevcn_amd.vcn as V
        from seevcn_amd.seeding import seeded_state_dict
        from seevcn_amd.vcn.utils import sampling
        n_obj = scenes or 64
        objs, _ = synth.make_object_batch(n_obj, seed=1000 + 1000 * rank)
        x = torch.from_numpy(objs).to(device)
        vcn = V.MODELS.build({"NAME": "VCN_VC"})
        vcn.load_state_dict(seeded_state_dict(vcn, seed=0))
        vcn = vcn.to(device).eval()

        def step():
            with torch.no_grad():
                coarse = vcn({"input": x})["coarse"]
                surface, n_sel = sampling.get_partial_mesh_batch_device(x, coarse, k=30)
                clustered, _ = sampling.get_largest_cluster_batch_device(surface, eps=0.4, min_points=2, total_pts=coarse.shape[1], period=n_sel)
            return clustered
        return step, n_obj, "objects/sec", "completed objects/sec (VCN_VC fwd + surface select + largest cluster)", {
            "workload": f"BASELINE configs[1]: VCN surface completion only, {n_obj} objects x 1024 pts per GPU, inference", "objects_per_gpu": n_obj}

    import bench
    if config == "second":
        n = scenes or 16
        pts, gt = synth.make_scene_batch(n, seed=2000 + 1000 * rank, n_az=bench.SCENE_N_AZ)
        ds = C.SyntheticDatasetInfo()
        net = _detector(C.second_model_cfg(dynamic_vfe=True), 3, ds, device, 5)
        batch = {"batch_size": n, "points": torch.from_numpy(pts).to(device), "gt_boxes": torch.from_numpy(gt).to(device)}
        workload = (f"BASELINE configs[2]: SECONDNet train step (DynMeanVFE, VoxelBackBone8x, HeightCompression, BaseBEVBackbone, AnchorHeadSingle, "
                    f"losses, backward, SGD), {n} KITTI-shaped scenes per GPU ({len(pts) / n / 1e3:.1f}k returns each)")
    elif config == "pvrcnn":
        import seevcn_amd.config_inputs as ci
        n = scenes or 4
        pts, gt = ci.pvrcnn_scene_batch(n, seed=3000 + 1000 * rank, n_az=350)
        gt = gt.copy()
        gt[:, :, 7] = (gt[:, :, 3] > 0)
        ds = C.SyntheticDatasetInfo(class_names=["car"], point_cloud_range=C.DA_RANGE, voxel_size=C.DA_VOXEL, num_point_features=3)
        net = _detector(C.see_pvrcnn_model_cfg(), 1, ds, device, 6)
        # points_per_scene: what collate_batch hands over with the points (host-side counts: the keypoint sampling then starts without a read-back)
        batch = {"batch_size": n, "points": torch.from_numpy(pts).to(device), "gt_boxes": torch.from_numpy(gt).to(device),
                 "points_per_scene": np.bincount(pts[:, 0].astype(np.int64), minlength=n).tolist()}
        workload = (f"BASELINE configs[3]: SEE-VCN PV-RCNN train step, DA geometry [41,1504,1504], 4096 keypoints, 512 proposals -> 128 RoIs x 216 grid "
                    f"points, {n} 360-degree scenes per GPU ({len(pts) / n / 1e3:.1f}k returns each; bs 32 = 8 GPUs x 4)")
    elif config == "centerpoint":
        import seevcn_amd.config_inputs as ci
        from seevcn_amd.pcdet.ops import voxel_ops
        n = scenes or 1
        clouds, boxes = [], []
        for i in range(n):
            p, g = ci.centerpoint_scene(seed=4000 + 1000 * rank + i, n_az=1200)
            clouds.append(p[np.random.default_rng(i).permutation(len(p))])
            boxes.append(np.concatenate([g[:, :7], np.zeros((len(g), 2), np.float32), g[:, 7:8]], axis=1))
        gmax = max(len(b) for b in boxes)
        gt = np.zeros((n, gmax, 10), np.float32)
        for i, b in enumerate(boxes):
            gt[i, :len(b)] = b
        counts = [len(p) for p in clouds]
        flat = torch.from_numpy(np.concatenate(clouds, axis=0)).to(device)
        grid = np.round((np.array(ci.NUSC_RANGE[3:]) - np.array(ci.NUSC_RANGE[:3])) / np.array(ci.NUSC_VOXEL)).astype(np.int64)
        ds = C.SyntheticDatasetInfo(class_names=C.NUSC_CLASS_NAMES, point_cloud_range=ci.NUSC_RANGE, voxel_size=ci.NUSC_VOXEL, num_point_features=3)
        net = _detector(C.centerpoint_model_cfg(), 10, ds, device, 21)
        gt_dev = torch.from_numpy(gt).to(device)
        params = [p for p in net.parameters() if p.requires_grad]
        opt = torch.optim.SGD(params, lr=1e-4, momentum=0.9, fused=True)
        world = int(os.environ.get("WORLD_SIZE", "1"))

        def step():
            # the hard voxeliser (first-come slots, 10 points / voxel, 120 k voxels) is part of the step: it is the reference's per-frame
            # DataProcessor.transform_points_to_voxels (data_processor.py:156-193), here on the GPU
            vox, crd, nmp, nv = voxel_ops.voxelize_hard(flat, 0, 3, counts, ci.NUSC_RANGE, ci.NUSC_VOXEL, grid, 10, 120000)
            nvl = [int(v) for v in nv]
            voxels = torch.cat([vox[i, :k] for i, k in enumerate(nvl)], dim=0)
            coords = torch.cat([torch.cat([torch.full((k, 1), i, dtype=torch.int32, device=device), crd[i, :k]], dim=1) for i, k in enumerate(nvl)], dim=0)
            num = torch.cat([nmp[i, :k] for i, k in enumerate(nvl)], dim=0)
            batch = {"batch_size": n, "voxels": voxels, "voxel_coords": coords, "voxel_num_points": num, "gt_boxes": gt_dev}
            return _train_step(net, batch, opt, params, world, bench.allreduce_grads)
        return step, n, "scenes/sec", "scenes/sec (CenterPoint voxel backbone train step, 300k-pt scenes)", {
            "workload": (f"BASELINE configs[4]: CenterPoint train step (hard voxelise 10 pts/voxel <= 120k voxels, MeanVFE, VoxelResBackBone8x, HeightCompression, "
                         f"BaseBEVBackbone, CenterHead, losses, backward, SGD), {n} nuScenes-shaped scene(s) of {counts[0] / 1e3:.0f}k points per GPU"),
            "scenes_per_gpu": n, "points_per_scene": counts[0]}
    else:
        raise ValueError(config)

    params = [p for p in net.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=1e-4, momentum=0.9, fused=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))

    def step():
        return _train_step(net, batch, opt, params, world, bench.allreduce_grads)
    name = {"second": "scenes/sec (SECOND train step)", "pvrcnn": "scenes/sec (SEE-VCN PV-RCNN train step)"}[config]
    return step, n, "scenes/sec", name, {"workload": workload, "scenes_per_gpu": n}
