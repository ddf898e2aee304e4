"""genesis.utils.geom helpers used by the Go2Env files (go2_env_walk.py:4: quat_to_xyz, transform_by_quat, inv_quat,
transform_quat_by_quat; genesis/utils/geom.py:717-762,989-1070), re-exported from the shim."""
from go2_sim2real_locomotion_rl_amd.genesis_shim import inv_quat, quat_to_xyz, transform_by_quat, transform_quat_by_quat  # noqa: F401

__all__ = ["inv_quat", "quat_to_xyz", "transform_by_quat", "transform_quat_by_quat"]
