#!/bin/bash
# After tools/profile_round.sh <tag> on the GPU box: copy the summaries the docs cite into profiles/ as r03_* and write the bench records
# (run from the repo root on the GPU box; everything lands under gpurun_out/ and is copied to profiles/ by the caller).
set -e
TAG=${1:-r03}
P=gpurun_out/prof_$TAG
D=gpurun_out/final_profiles
mkdir -p $D
cp $P/kernel_window.txt $D/r03_kernel_window.txt; cp $P/kernel_window_rocprof_stats.csv $D/r03_kernel_stats.csv
cp $P/steady_kernel_window.txt $D/r03_steady_kernel_window.txt; cp $P/steady_kernel_window_rocprof_stats.csv $D/r03_steady_kernel_stats.csv
cp $P/pmc_fetch_size.txt $D/r03_pmc_fetch_size.txt; cp $P/pmc_write_size.txt $D/r03_pmc_write_size.txt; cp $P/pmc_sq.txt $D/r03_pmc_sq.txt
cp $P/pmc_traffic.json $D/r03_pmc_traffic.json
cp $P/steady_pmc_fetch_size.txt $D/r03_steady_pmc_fetch_size.txt; cp $P/steady_pmc_write_size.txt $D/r03_steady_pmc_write_size.txt; cp $P/steady_pmc_traffic.json $D/r03_steady_pmc_traffic.json
cp $P/stairs_kernel_window.txt $D/r03_stairs_kernel_window.txt; cp $P/stairs_kernel_window_rocprof_stats.csv $D/r03_stairs_kernel_stats.csv
cp $P/stairs_pmc_fetch_size.txt $D/r03_stairs_pmc_fetch_size.txt; cp $P/stairs_pmc_write_size.txt $D/r03_stairs_pmc_write_size.txt; cp $P/stairs_pmc_traffic.json $D/r03_stairs_pmc_traffic.json
cp $P/policy_kernel_stats.csv $D/r03_policy_kernel_stats.csv
# bench.py reads the traffic files from profiles/
cp $D/r03_pmc_traffic.json profiles/r03_pmc_traffic.json; cp $D/r03_stairs_pmc_traffic.json profiles/r03_stairs_pmc_traffic.json
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 2> gpurun_out/bench_driver_window.err | tail -1 > $D/r03_bench_driver_window.json
timeout -k 10 400 python3 bench.py 2> gpurun_out/bench_default.err | tail -1 > $D/r03_bench_default.json
timeout -k 10 300 python3 bench.py --workload stairs --no-extras --no-cpu-baseline 2> gpurun_out/bench_stairs.err | tail -1 > $D/r03_bench_stairs.json
python3 tools/phase_profile.py 5 20 2>&1 | grep -v amdgpu.ids > $D/r03_phase_profile_driver_window.txt
python3 tools/phase_profile.py 5 20 each 2>&1 | grep -v amdgpu.ids >> $D/r03_phase_profile_driver_window.txt
GO2SIM_PROFILE_WORKLOAD=stairs python3 tools/phase_profile.py 100 100 2>&1 | grep -v amdgpu.ids > $D/r03_phase_profile_stairs.txt
cut -c1-200 $D/r03_bench_driver_window.json
