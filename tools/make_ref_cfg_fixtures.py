#!/usr/bin/env python3
"""Build-container script: turn the reference-held config pickles into JSON fixtures.

    python tools/make_ref_cfg_fixtures.py        # reads /root/reference/logs/<exp>/cfgs.pkl, writes tests/golden/ref_cfgs_<task>.json

`logs/<exp>/cfgs.pkl` is what go2_train_*.py writes next to its checkpoints (go2_train_walk.py:462-465): the 5-element list
[env_cfg, obs_cfg, reward_cfg, command_cfg, train_cfg] of plain dicts / lists / strings / numbers.  They are the only reference-held data
on this path.  The files are read with an unpickler that refuses every global (`find_class` raises), i.e. nothing
from the file is executed or imported; a pickle that needs a global makes the script fail instead of loading it.  The JSON fixtures are
data (inputs of the env), not reference source text.  /root/reference does not exist on the GPU box: tests read only the JSON."""
import io
import json
import os
import pickle
import sys

REF_LOGS = "/root/reference/logs"
OUT_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
TASKS = {"walk": "go2-walk", "stairs": "go2-stairs", "jump": "go2-jump", "crouch": "go2-crouch"}


class NoGlobalsUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        raise pickle.UnpicklingError(f"refusing global {module}.{name}: config pickles hold plain containers only")


def load_plain_pickle(path):
    with open(path, "rb") as f:
        return NoGlobalsUnpickler(io.BytesIO(f.read())).load()


def main():
    os.makedirs(OUT_DIR, exist_ok=True)
    for task, exp in TASKS.items():
        src = os.path.join(REF_LOGS, exp, "cfgs.pkl")
        cfgs = load_plain_pickle(src)
        assert isinstance(cfgs, (list, tuple)) and len(cfgs) == 5, f"{src}: expected [env, obs, reward, command, train]"
        out = {"source": f"logs/{exp}/cfgs.pkl", "layout": ["env_cfg", "obs_cfg", "reward_cfg", "command_cfg", "train_cfg"],
               "env_cfg": cfgs[0], "obs_cfg": cfgs[1], "reward_cfg": cfgs[2], "command_cfg": cfgs[3], "train_cfg": cfgs[4]}
        dst = os.path.join(OUT_DIR, f"ref_cfgs_{task}.json")
        with open(dst, "w") as f:
            json.dump(out, f, indent=1)   # insertion order kept: the order of reward_scales is the evaluation order of the reward terms
            f.write("\n")
        print(f"{src} -> {dst} ({os.path.getsize(dst)} bytes)")


if __name__ == "__main__":
    sys.exit(main())
