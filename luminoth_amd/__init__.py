"""luminoth_amd — MI355X-native rebuild of the Faster R-CNN / SSD train-step hot
path of tryolabs/luminoth, behind luminoth's own model-module API
(`luminoth_amd.models.get_model`, reference: luminoth/models/models.py:6-17).

Python host code -> thin C ABI (`include/luminoth_hip.h`, loaded with ctypes)
-> hand-written gfx950 HIP kernels (`luminoth_amd/csrc`).  PyTorch-ROCm is used
for device memory, streams, autograd plumbing and torch.distributed (RCCL).
"""
__version__ = '0.1.0'
