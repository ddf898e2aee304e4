#!/bin/bash
# After tools/profile_round.sh <tag> on the GPU box: copy the summaries the docs cite into profiles/ as <round>_* and write the bench records   usage: finalize_round.sh <tag> <round, e.g. r04>
# (run from the repo root on the GPU box; everything lands under gpurun_out/ and is copied to profiles/ by the caller).
set -e
TAG=${1:-r04}
R=${2:-r04}
P=gpurun_out/prof_$TAG
D=gpurun_out/final_profiles
mkdir -p $D
cp $P/kernel_window.txt $D/${R}_kernel_window.txt; cp $P/kernel_window_rocprof_stats.csv $D/${R}_kernel_stats.csv
cp $P/steady_kernel_window.txt $D/${R}_steady_kernel_window.txt; cp $P/steady_kernel_window_rocprof_stats.csv $D/${R}_steady_kernel_stats.csv
cp $P/pmc_fetch_size.txt $D/${R}_pmc_fetch_size.txt; cp $P/pmc_write_size.txt $D/${R}_pmc_write_size.txt; cp $P/pmc_sq.txt $D/${R}_pmc_sq.txt
cp $P/pmc_traffic.json $D/${R}_pmc_traffic.json; cp $P/pmc_sq_stamped.json $D/${R}_pmc_sq.json; cp $P/steady_pmc_sq_stamped.json $D/${R}_steady_pmc_sq.json; cp $P/pmc_sq2.txt $D/${R}_pmc_sq2.txt
cp $P/steady_pmc_fetch_size.txt $D/${R}_steady_pmc_fetch_size.txt; cp $P/steady_pmc_write_size.txt $D/${R}_steady_pmc_write_size.txt; cp $P/steady_pmc_traffic.json $D/${R}_steady_pmc_traffic.json
cp $P/stairs_kernel_window.txt $D/${R}_stairs_kernel_window.txt; cp $P/stairs_kernel_window_rocprof_stats.csv $D/${R}_stairs_kernel_stats.csv
cp $P/stairs_pmc_fetch_size.txt $D/${R}_stairs_pmc_fetch_size.txt; cp $P/stairs_pmc_write_size.txt $D/${R}_stairs_pmc_write_size.txt; cp $P/stairs_pmc_traffic.json $D/${R}_stairs_pmc_traffic.json
cp $P/policy_kernel_stats.csv $D/${R}_policy_kernel_stats.csv
# bench.py reads the traffic and SQ files from profiles/
cp $D/${R}_pmc_traffic.json profiles/${R}_pmc_traffic.json; cp $D/${R}_stairs_pmc_traffic.json profiles/${R}_stairs_pmc_traffic.json
cp $D/${R}_pmc_sq.json profiles/${R}_pmc_sq.json; cp $D/${R}_steady_pmc_sq.json profiles/${R}_steady_pmc_sq.json
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 2> gpurun_out/bench_driver_window.err | tail -1 > $D/${R}_bench_driver_window.json
timeout -k 10 400 python3 bench.py 2> gpurun_out/bench_default.err | tail -1 > $D/${R}_bench_default.json
timeout -k 10 300 python3 bench.py --workload stairs --no-extras --no-cpu-baseline 2> gpurun_out/bench_stairs.err | tail -1 > $D/${R}_bench_stairs.json
python3 tools/phase_profile.py 5 20 2>&1 | grep -v amdgpu.ids > $D/${R}_phase_profile_driver_window.txt
python3 tools/phase_profile.py 5 20 each 2>&1 | grep -v amdgpu.ids >> $D/${R}_phase_profile_driver_window.txt
GO2SIM_PROFILE_WORKLOAD=stairs python3 tools/phase_profile.py 100 100 2>&1 | grep -v amdgpu.ids > $D/${R}_phase_profile_stairs.txt
cut -c1-200 $D/${R}_bench_driver_window.json
# what a launch costs (per-workgroup wall-clock stamps) and the decoupling bound on the current kernels
for a in "4096 5 20" "256 5 20" "4096 300 50" "256 300 50"; do python3 tools/launch_overhead.py $a 2>&1 | grep -v amdgpu.ids; echo; done > $D/${R}_launch_overhead.txt
python3 tools/phase_profile.py 5 20 decouple 2>&1 | grep -v amdgpu.ids > $D/${R}_decoupling_bound_window.txt
python3 tools/phase_profile.py 300 100 decouple 2>&1 | grep -v amdgpu.ids > $D/${R}_decoupling_bound_steady.txt
