"""Files of the eval / teleop surface: the config pickle and the policy checkpoint of a training run.

* ``logs/<exp>/cfgs.pkl`` -- the 5-element list ``[env_cfg, obs_cfg, reward_cfg, command_cfg, train_cfg]`` of plain dicts / lists / strings /
  numbers written by go2_train_walk.py:462-465 and read back by go2_eval_walk.py:500 / go2_eval_stairs.py:486.  ``load_cfgs`` reads it with an
  unpickler that refuses every global, i.e. nothing from the file is imported or executed; ``save_cfgs`` writes the same layout.
* ``logs/<exp>/model_<it>.pt`` -- rsl_rl 2.2.4 ``OnPolicyRunner.save``: ``{"model_state_dict", "optimizer_state_dict", "iter", "infos"}``.
  ``read_checkpoint`` loads it with ``torch.load(weights_only=True)``; ``compatible_state_dict`` is the partial load of
  go2_eval_stairs.py:368-450 (``load_model_compat`` / ``_partial_load``): tensors whose shapes match are taken from the file, the rest (typically
  the first critic layer of a checkpoint trained with another privileged-observation width) keep their current values.
"""
import io
import pickle

import torch

CFG_LAYOUT = ("env_cfg", "obs_cfg", "reward_cfg", "command_cfg", "train_cfg")


class _NoGlobalsUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        raise pickle.UnpicklingError(f"refusing global {module}.{name}: cfgs.pkl holds plain containers only")


def load_cfgs(path):
    """-> (env_cfg, obs_cfg, reward_cfg, command_cfg, train_cfg) of ``logs/<exp>/cfgs.pkl`` (go2_train_walk.py:462-465)."""
    with open(path, "rb") as f:
        cfgs = _NoGlobalsUnpickler(io.BytesIO(f.read())).load()
    if not isinstance(cfgs, (list, tuple)) or len(cfgs) != len(CFG_LAYOUT) or not all(isinstance(c, dict) for c in cfgs):
        raise ValueError(f"{path}: expected the 5-element list {list(CFG_LAYOUT)}")
    return tuple(cfgs)


def _plain(x):
    if isinstance(x, dict):
        return {str(k): _plain(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_plain(v) for v in x]
    if isinstance(x, (bool, int, float, str)) or x is None:
        return x
    if hasattr(x, "item") and getattr(x, "ndim", 1) == 0:      # numpy / torch scalars
        return x.item()
    if hasattr(x, "tolist"):
        return _plain(x.tolist())
    raise TypeError(f"cfgs.pkl holds plain containers only, got {type(x).__name__}")


def save_cfgs(path, env_cfg, obs_cfg, reward_cfg, command_cfg, train_cfg):
    """Write ``cfgs.pkl`` in the reference's layout (plain containers only, so that ``load_cfgs`` and the reference's ``pickle.load`` both read it)."""
    with open(path, "wb") as f:
        pickle.dump([_plain(env_cfg), _plain(obs_cfg), _plain(reward_cfg), _plain(command_cfg), _plain(train_cfg)], f)


def read_checkpoint(path, map_location="cpu"):
    """rsl_rl 2.2.4 checkpoint -> dict with at least ``model_state_dict`` (a bare state dict is accepted too, go2_eval_stairs.py:380)."""
    ckpt = torch.load(path, map_location=map_location, weights_only=True)
    if not isinstance(ckpt, dict):
        raise ValueError(f"{path}: not a checkpoint dictionary")
    if "model_state_dict" not in ckpt:
        ckpt = {"model_state_dict": ckpt, "iter": 0, "infos": None}
    return ckpt


def save_checkpoint(path, model_state_dict, optimizer_state_dict=None, it=0, infos=None):
    """``OnPolicyRunner.save`` layout (rsl_rl 2.2.4 runners/on_policy_runner.py)."""
    torch.save({"model_state_dict": {k: v.detach().cpu() for k, v in model_state_dict.items()},
                "optimizer_state_dict": optimizer_state_dict if optimizer_state_dict is not None else {}, "iter": int(it), "infos": infos}, path)


def compatible_state_dict(current, saved):
    """go2_eval_stairs.py:422-450 ``_partial_load``: -> (merged state dict, loaded keys, skipped {key: reason})."""
    merged, loaded, skipped = dict(current), [], {}
    for k, v in saved.items():
        if k not in current:
            skipped[k] = "not in current model"
        elif tuple(current[k].shape) != tuple(v.shape):
            skipped[k] = f"saved={list(v.shape)} vs current={list(current[k].shape)}"
        else:
            merged[k] = v
            loaded.append(k)
    return merged, loaded, skipped


def critic_input_mismatch(saved, expected_critic_in):
    """go2_eval_stairs.py:382-407: expected minus saved input width of the first critic layer (0 = compatible; 1 = model without terrain_row,
    77 = without height scan, 78 = walking model evaluated in the stair env)."""
    for k, v in saved.items():
        if "critic" in k.lower() and "weight" in k.lower():
            return int(expected_critic_in) - int(v.shape[1])
    return 0
