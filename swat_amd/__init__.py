"""swat_amd — MI355X-native (gfx950) implementation of tian1327/SWAT's two compute paths:
(A) the OpenCLIP ViT finetuning loop and (B) the embedding / cosine-similarity retrieval.

Host code is PyTorch-ROCm Python mirroring the reference's plug-in surface; every hot op is a
hand-written HIP kernel reached through the C ABI in include/swat_hip.h (swat_amd/_lib.py).
There is NO CPU fallback: ops raise if libswat_hip.so is missing or a tensor is not on the GPU.
"""
__version__ = "0.1.0"
