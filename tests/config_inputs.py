"""Shim: the full-size inputs of BASELINE configs 4 and 5 live in the package (seevcn_amd.config_inputs)."""
from seevcn_amd.config_inputs import *  # noqa: F401,F403
from seevcn_amd.config_inputs import DA_RANGE, DA_VOXEL, NUSC_RANGE, NUSC_VOXEL, NUSC_SIZES, pvrcnn_scene_batch, centerpoint_scene  # noqa: F401
