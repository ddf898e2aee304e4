"""Scalars of the one TensorBoard event file the reference ships (logs/test1/events.out.tfevents.*): the only measured throughput of this path in
the reference tree.  tensorboard / tensorflow are not importable here, so the TFRecord framing and the Event / Summary protobufs are decoded
directly (records: u64 length, u32 crc, payload, u32 crc; Event{1: wall_time f64, 2: step varint, 5: Summary{1: Value{1: tag, 2: simple_value f32}}}).

    python tools/extract_ref_tb_scalars.py     # build container only -> tests/golden/ref_test1_tb_scalars.json
"""
import glob
import json
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def varint(b, i):
    v = s = 0
    while True:
        x = b[i]; i += 1
        v |= (x & 0x7F) << s
        s += 7
        if x < 0x80:
            return v, i


def fields(b):
    i = 0
    while i < len(b):
        key, i = varint(b, i)
        no, wt = key >> 3, key & 7
        if wt == 0:
            v, i = varint(b, i)
        elif wt == 1:
            v = b[i:i + 8]; i += 8
        elif wt == 2:
            n, i = varint(b, i)
            v = b[i:i + n]; i += n
        elif wt == 5:
            v = b[i:i + 4]; i += 4
        else:
            raise ValueError(f"wire type {wt}")
        yield no, wt, v


def main():
    paths = sorted(glob.glob("/root/reference/logs/test1/events.out.tfevents.*"))
    if not paths:
        raise SystemExit("the reference tree is not present")
    data = open(paths[0], "rb").read()
    i, scalars = 0, {}
    while i + 12 <= len(data):
        (n,) = struct.unpack("<Q", data[i:i + 8])
        payload = data[i + 12:i + 12 + n]
        i += 12 + n + 4
        step, summary = 0, None
        for no, wt, v in fields(payload):
            if no == 2 and wt == 0:
                step = v
            elif no == 5 and wt == 2:
                summary = v
        if summary is None:
            continue
        for no, wt, v in fields(summary):
            if no != 1 or wt != 2:
                continue
            tag, val = None, None
            for n2, w2, v2 in fields(v):
                if n2 == 1 and w2 == 2:
                    tag = v2.decode()
                elif n2 == 2 and w2 == 5:
                    (val,) = struct.unpack("<f", v2)
            if tag is not None and val is not None:
                scalars.setdefault(tag, []).append([int(step), float(val)])
    perf = {k: v for k, v in scalars.items() if k.startswith("Perf/")}
    out = {"source": "logs/test1/" + os.path.basename(paths[0]), "note": "scalars written by rsl_rl's OnPolicyRunner.log of a go2_train_stair.py run (logs/test1/cfgs.pkl)",
           "scalars": scalars}
    # derived: total_fps x (collection + learning) = steps_per_env x num_envs  ->  num_envs; collection-only env-steps/s
    if {"Perf/total_fps", "Perf/collection time", "Perf/learning_time"} <= set(perf):
        rows = []
        for (s, fps), (_, tc), (_, tl) in zip(perf["Perf/total_fps"], perf["Perf/collection time"], perf["Perf/learning_time"]):
            transitions = fps * (tc + tl)
            rows.append({"iteration": s, "total_fps": fps, "collection_s": tc, "learning_s": tl, "transitions_per_iteration": round(transitions),
                         "collection_env_steps_per_s": transitions / tc})
        out["derived"] = {"per_iteration": rows, "num_envs_if_24_steps_per_env": round(rows[0]["transitions_per_iteration"] / 24) if rows else None}
    dst = os.path.join(ROOT, "tests", "golden", "ref_test1_tb_scalars.json")
    json.dump(out, open(dst, "w"), indent=1)
    print(dst, {k: len(v) for k, v in scalars.items() if k.startswith("Perf/")}, out.get("derived"))


if __name__ == "__main__":
    main()
