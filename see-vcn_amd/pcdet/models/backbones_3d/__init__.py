from .spconv_backbone import VoxelBackBone8x, VoxelResBackBone8x

# same registry shape as the reference (backbones_3d/__init__.py:6-13)
__all__ = {
    'VoxelBackBone8x': VoxelBackBone8x,
    'VoxelResBackBone8x': VoxelResBackBone8x,
}
