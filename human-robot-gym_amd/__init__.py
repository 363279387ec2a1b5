"""human-robot-gym_amd — MI355X-native batched stepper for human-robot-gym's ReachHuman hot path.

The directory name follows the project naming (`human-robot-gym_amd`); import it as `human_robot_gym_amd`
(the sibling shim package points its `__path__` here).
"""
from .model import build_model_desc, DEFAULT_ENV_KWARGS  # noqa: F401
from .animation import ClipSet, synthetic_clips, static_clip  # noqa: F401

__version__ = "0.1.0"


def __getattr__(name):
    # heavy (torch / HIP library) pieces are imported lazily
    if name in ("HipVecEnv", "HipGymEnv"):
        from . import vec_env
        return getattr(vec_env, name)
    if name == "make_vec_env":
        from .env_util import make_vec_env
        return make_vec_env
    if name in ("create_training_vec_env", "compose_environment_kwargs", "wrapper_kwargs_from_config"):
        from . import training_utils
        return getattr(training_utils, name)
    if name in ("MixedBatch", "MixedHipVecEnv", "make_mixed_batch", "make_mixed_vec_env", "ICRA_TASKS"):
        from . import mixed
        return getattr(mixed, name)
    if name == "HipBatch":
        from ._lib import HipBatch
        return HipBatch
    raise AttributeError(name)
