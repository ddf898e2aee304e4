#!/bin/bash
# env-steps/s and ms per step against the number of envs (steady default run, then the driver's landing window); run from the repo root on the GPU box
for n in 256 1024 4096 16384 32768; do
  python3 bench.py --envs-per-gpu $n --no-extras --no-cpu-baseline --no-profile-pass 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print($n, d['value'], d['ms_per_step'])"
done
for n in 256 1024 4096 16384; do
  python3 bench.py --envs-per-gpu $n --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-profile-pass 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('window', $n, d['value'], d['ms_per_step'])"
done
