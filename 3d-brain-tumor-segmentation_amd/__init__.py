"""MI355X-native (gfx950) 3D U-Net+VAE segmentation engine: drop-in for the model.py / layers.* / util.py call
surface of vliu15/3d-brain-tumor-segmentation, driving hand-written HIP kernels through the C ABI in
include/bts_hip.h.  Import as `bts_amd` (see bts_amd.py at the repository root)."""
__version__ = '0.1'
