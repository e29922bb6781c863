"""seevcn_amd — MI355X-native hot path of SEE-VCN (VCN surface completion + voxelize → sparse conv → BEV → head).

Layout:
  csrc/      hand-written HIP kernels (gfx950) + the C-ABI (`include/seevcn_hip.h`) → lib/libseevcn_hip.so
  _lib.py    ctypes binding of that library (fails loudly if it is missing — no CPU fallback)
  spconv/    `spconv.pytorch`-shaped API over the HIP rulebook / sparse-conv kernels
  pcdet/     mirror of the reference's pcdet module registries + op wrappers for this path
  vcn/       VCN_VC / VCN_CN / VCN.inference mirror
  synth.py   deterministic synthetic inputs (tests, bench)
"""
__version__ = "0.1.0"
