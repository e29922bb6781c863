"""`spconv.pytorch`-shaped API on the seevcn HIP kernels.

Exposes exactly the surface the reference uses: SparseConvTensor, SubMConv3d, SparseConv3d, SparseInverseConv3d,
SparseSequential, SparseModule, `conv.SparseConvolution` (used by pcdet/utils/spconv_utils.py:11-25) and the `utils` voxel
generators the dataloader binds (datasets/processor/data_processor.py:17-59).
`import seevcn_amd.spconv as spconv` replaces `import spconv.pytorch as spconv` (spconv_utils.py:3-6).
"""
from . import conv  # noqa: F401
from . import utils  # noqa: F401  (spconv.utils.VoxelGeneratorV2 / VoxelGenerator / Point2VoxelCPU3d, data_processor.py:17-26)
from .conv import SparseConv3d, SparseConvolution, SparseInverseConv3d, SubMConv3d, prebuild_rulebooks, refresh_weight_fragments  # noqa: F401
from .core import SparseConvTensor  # noqa: F401
from .modules import SparseModule, SparseSequential  # noqa: F401

__version__ = "2.1.0-seevcn"  # 2.x weight layout (C_out, kz, ky, kx, C_in)
