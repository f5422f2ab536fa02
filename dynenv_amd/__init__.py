"""dynenv_amd — MI355X-native batched step() for DynEnv (drop-in for the reference's make_dyn_env path).

Only what the hot path needs lives here: csrc/ (HIP kernels + the C ABI of include/dynenv.h) and the host-side
mirror of the reference's vectorised-environment interface.
"""
from .enums import DynEnvType, NoiseType, ObservationType
from .vec_env import BatchedDynEnv, make_dyn_env
from .arranger import GpuInOutArranger, groups_for

__all__ = ["BatchedDynEnv", "make_dyn_env", "DynEnvType", "NoiseType", "ObservationType", "GpuInOutArranger", "groups_for"]
