#!/bin/bash
# One-shot profile of the bench workload on the GPU box (run from the repo root):   bash tools/profile_round.sh <tag> [W K]
#   1. rocprofv3 --kernel-trace --stats of the driver's command (bench.py --warmup W --steps K, default 5 / 20), summarised over the timed window
#   2. --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (rocprofv3 cannot hold both), same command, same window
#        -> <out>/pmc_traffic.json stamped with the source hash (copy to profiles/rNN_pmc_traffic.json: bench.py reads it, and refuses a stale one)
#   3. two SQ passes (issue slots, lane occupancy, waits; instruction mix, LDS) -> pmc_sq_stamped.json (copy to profiles/rNN_pmc_sq.json: bench.py reads it)
#   4. the same three for the steady state (400 steps after 300)
#   5. kernel stats + FETCH / WRITE passes of the stairs workload (-> stairs_pmc_traffic.json)
#   6. kernel stats of the policy kernels (tools/policy_bench.py)
# Everything lands in gpurun_out/prof_<tag>/; counters are never combined with any trace but --kernel-trace.
set -e
TAG=${1:-run}; W=${2:-5}; K=${3:-20}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --no-extras --no-cpu-baseline --no-profile-pass"
csvdir() { dirname $(ls $1/*/*_kernel_trace.csv $1/*_kernel_trace.csv 2>/dev/null | head -1); }
run_pass() {   # name, window fraction, bench args, rocprof args...
  local name=$1 frac=$2 bargs=$3; shift 3
  timeout -k 10 300 rocprofv3 --kernel-trace "$@" --output-format csv -d $OUT/$name -o p -- $BENCH $bargs > $OUT/$name.log 2>&1
  python3 $R/tools/pmc_summary.py $(csvdir $OUT/$name) p --tail $frac --json $OUT/$name.json > $OUT/$name.txt
  if [ -f $(csvdir $OUT/$name)/p_kernel_stats.csv ]; then cp $(csvdir $OUT/$name)/p_kernel_stats.csv $OUT/${name}_rocprof_stats.csv; fi
  rm -rf $OUT/$name
}
FR=$(python3 -c "print($K / ($W + $K))")
CMD="--warmup $W --steps $K"
run_pass kernel_window $FR "$CMD" --stats
run_pass pmc_fetch_size $FR "$CMD" --pmc FETCH_SIZE
run_pass pmc_write_size $FR "$CMD" --pmc WRITE_SIZE
run_pass pmc_sq $FR "$CMD" --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run_pass pmc_sq2 $FR "$CMD" --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_VMEM_RD
python3 $R/tools/make_pmc_sq.py $OUT/pmc_sq.json $OUT/pmc_sq_stamped.json \
  "rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -- python3 bench.py --no-extras --no-cpu-baseline --no-profile-pass $CMD (last $K steps)" > /dev/null
python3 $R/tools/make_pmc_traffic.py $OUT/pmc_fetch_size.json $OUT/pmc_write_size.json $OUT/pmc_traffic.json \
  "rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE> -- python3 bench.py --no-extras --no-cpu-baseline --no-profile-pass $CMD (last $K steps)" > /dev/null
SS="--warmup 300 --steps 400"; SF=$(python3 -c "print(400 / 700)")
run_pass steady_kernel_window $SF "$SS" --stats
run_pass steady_pmc_fetch_size $SF "$SS" --pmc FETCH_SIZE
run_pass steady_pmc_write_size $SF "$SS" --pmc WRITE_SIZE
run_pass steady_pmc_sq $SF "$SS" --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
python3 $R/tools/make_pmc_sq.py $OUT/steady_pmc_sq.json $OUT/steady_pmc_sq_stamped.json "the same SQ pass, bench.py $SS (last 400 steps)" > /dev/null
python3 $R/tools/make_pmc_traffic.py $OUT/steady_pmc_fetch_size.json $OUT/steady_pmc_write_size.json $OUT/steady_pmc_traffic.json \
  "rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE> -- python3 bench.py --no-extras --no-cpu-baseline --no-profile-pass $SS (last 400 steps)" > /dev/null
# stairs (BASELINE configs[2]): bench.py --workload stairs, default window (100 warm-up + 300 timed steps)
ST="--workload stairs --warmup 100 --steps 300"; STF=$(python3 -c "print(300 / 400)")
run_pass stairs_kernel_window $STF "$ST" --stats
run_pass stairs_pmc_fetch_size $STF "$ST" --pmc FETCH_SIZE
run_pass stairs_pmc_write_size $STF "$ST" --pmc WRITE_SIZE
python3 $R/tools/make_pmc_traffic.py $OUT/stairs_pmc_fetch_size.json $OUT/stairs_pmc_write_size.json $OUT/stairs_pmc_traffic.json \
  "rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE> -- python3 bench.py --no-extras --no-cpu-baseline --no-profile-pass $ST (last 300 steps)" > /dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pol -o pol -- python3 $R/tools/policy_bench.py 4096 > $OUT/policy_bench.log 2>&1
cp $(ls $OUT/pol/*/pol_kernel_stats.csv $OUT/pol/pol_kernel_stats.csv 2>/dev/null | head -1) $OUT/policy_kernel_stats.csv
rm -rf $OUT/pol
head -12 $OUT/kernel_window.txt; python3 -c "
import json; d = json.load(open('$OUT/pmc_traffic.json')); print('transient env-step bytes', d['env_step_bytes'] / 1e6, 'MB')
d = json.load(open('$OUT/steady_pmc_traffic.json')); print('steady env-step bytes', d['env_step_bytes'] / 1e6, 'MB')"
