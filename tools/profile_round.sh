#!/bin/bash
# One-shot profile of the bench workload on the GPU box: kernel stats + the two PMC traffic passes (separate runs, as rocprofv3 requires).
# usage (from the repo root, on the box): bash tools/profile_round.sh <tag>     -> gpurun_out/prof_<tag>/{kernel_stats.csv,pmc_fetch_size.txt,pmc_write_size.txt,pmc_sq.txt}
set -e
TAG=${1:-run}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks -o ks -- python3 $R/bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-profile-pass > $OUT/ks.log 2>&1
cp $(ls $OUT/ks/*/ks_kernel_stats.csv $OUT/ks/ks_kernel_stats.csv 2>/dev/null | head -1) $OUT/kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  c=$(echo $C | tr A-Z a-z)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$c -o p -- python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-profile-pass > $OUT/$c.log 2>&1
  D=$(dirname $(ls $OUT/$c/*/p_counter_collection.csv $OUT/$c/p_counter_collection.csv 2>/dev/null | head -1))
  python3 $R/tools/pmc_summary.py $D p > $OUT/pmc_$c.txt
done
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/sq -o p -- python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-profile-pass > $OUT/sq.log 2>&1
D=$(dirname $(ls $OUT/sq/*/p_counter_collection.csv $OUT/sq/p_counter_collection.csv 2>/dev/null | head -1))
python3 $R/tools/pmc_summary.py $D p > $OUT/pmc_sq.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pol -o pol -- python3 $R/tools/policy_bench.py 4096 > $OUT/policy_bench.log 2>&1
cp $(ls $OUT/pol/*/pol_kernel_stats.csv $OUT/pol/pol_kernel_stats.csv 2>/dev/null | head -1) $OUT/policy_kernel_stats.csv
rm -rf $OUT/ks $OUT/fetch_size $OUT/write_size $OUT/sq $OUT/pol
head -8 $OUT/kernel_stats.csv | cut -c1-160; head -6 $OUT/pmc_fetch_size.txt; head -6 $OUT/pmc_write_size.txt
