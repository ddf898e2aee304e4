"""go2_sim2real_locomotion_rl_amd -- MI355X-native vectorised Go2 locomotion environment.

Drop-in for the hot path of saifahmadgit/go2-sim2real-locomotion-rl:
``Go2Env.step()/reset()`` over ``gs.Scene.step()``."""
from .capi import C, Go2Sim, Go2SimError, load_hip_lib  # noqa: F401
from .configs import (flatten_base_cfg, flatten_walk_cfg, get_crouch_cfgs, get_jump_cfgs, get_stair_cfgs,  # noqa: F401
                      get_walk_cfgs)
from .eval_io import load_cfgs, read_checkpoint, save_cfgs, save_checkpoint  # noqa: F401
from .go2_env import Go2Env, init  # noqa: F401
from .model_blob import load_model_json, pack_model  # noqa: F401
from .policy import ActorCritic  # noqa: F401
from .rollout import RolloutStorage  # noqa: F401

__all__ = ["Go2Env", "ActorCritic", "RolloutStorage", "init", "C", "Go2Sim", "Go2SimError", "load_hip_lib", "flatten_walk_cfg", "flatten_base_cfg", "get_walk_cfgs", "get_stair_cfgs",
           "get_crouch_cfgs", "get_jump_cfgs", "load_model_json", "pack_model", "load_cfgs", "save_cfgs", "read_checkpoint", "save_checkpoint"]
