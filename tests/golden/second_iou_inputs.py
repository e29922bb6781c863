"""SECOND-IoU head test configuration shared by the golden generator and the test."""
import numpy as np

HEAD_KW = dict(in_channel=8, shared_fc=(64, 64), iou_fc=(64, 64), roi_per_image=32, train_post=64, test_post=50)
DATASET_CFG = dict(POINT_CLOUD_RANGE=[0, -40, -3, 70.4, 40, 1],
                   DATA_PROCESSOR=[dict(NAME='mask_points_and_boxes_outside_range'), dict(NAME='transform_points_to_voxels', VOXEL_SIZE=[0.05, 0.05, 0.1])])


def bev_map():
    return np.random.default_rng(77).normal(size=(2, 8, 200, 176)).astype(np.float32)
