"""Synthetic inputs of BASELINE configs 4 and 5 at their FULL size, shared by the parity tests and bench.py's side modes (package data so that bench.py
does not import from tests/).

config 4: 360-degree 64-beam scenes in the SEE-VCN domain-adaptation geometry, range [-75.2,-75.2,-2,75.2,75.2,4] (z shifted +1.8 like SHIFT_COOR,
          detector3d/tools/cfgs/source-nuscenes/pvrcnn.yaml:7), voxel [0.1,0.1,0.15] -> sparse [41,1504,1504]; 1 class; NUM_KEYPOINTS 4096 (:111-115).
config 5: one nuScenes-shaped scene = 10 superimposed 32-beam sweeps with ego-motion jitter, ~300 k points, range [-54,-54,-5,54,54,3], voxel
          [0.075,0.075,0.2] -> sparse [41,1440,1440], <= 10 points per voxel, <= 120 000 voxels (cbgs_voxel0075_res3d_centerpoint.yaml:6,56-61)."""
import numpy as np

from . import synth

DA_RANGE = [-75.2, -75.2, -2.0, 75.2, 75.2, 4.0]
DA_VOXEL = [0.1, 0.1, 0.15]
NUSC_RANGE = [-54.0, -54.0, -5.0, 54.0, 54.0, 3.0]
NUSC_VOXEL = [0.075, 0.075, 0.2]
NUSC_SIZES = ((4.63, 1.97, 1.74), (6.93, 2.51, 2.84), (6.37, 2.85, 3.19), (10.5, 2.94, 3.47), (12.29, 2.90, 3.87), (0.50, 2.53, 0.98),
              (2.11, 0.77, 1.47), (1.70, 0.60, 1.28), (0.73, 0.67, 1.77), (0.41, 0.41, 1.07))      # mean box sizes of the 10 nuScenes classes


def pvrcnn_scene_batch(batch_size=4, seed=3000, n_az=520):
    """(points (sum P, 4) [b,x,y,z], gt_boxes (B,G,8) class 1): 64 beams x 520 azimuth steps over 360 degrees, ~21 k returns per scene."""
    return synth.make_scene_batch(batch_size, seed=seed, az=(-180.0, 180.0), n_az=n_az, box_area=((-60.0, 60.0), (-60.0, 60.0)), ground_z=-1.73 + 1.8,
                                  z_shift=0.0, max_range=100.0, sizes=((3.9, 1.6, 1.56),), max_boxes=64)


def centerpoint_scene(seed=4000, n_az=1500):
    """(points (P,3) ~300 k, gt_boxes (G,8) with class 1..10 in column 7)."""
    return synth.make_scene(seed, n_beams=32, elev=(-30.0, 10.0), az=(-180.0, 180.0), n_az=n_az, n_sweeps=10, box_area=((-50.0, 50.0), (-50.0, 50.0)),
                            ground_z=-1.84, max_range=76.0, sizes=NUSC_SIZES, max_boxes=64)
