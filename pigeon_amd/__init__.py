import os as _os

# dmabuf IPC for multi-process GPU work (RCCL, CUDA-tensor sharing): must be in the environment before the HIP runtime
# initialises -- see pigeon_amd/distributed.py.
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
