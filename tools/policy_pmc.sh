#!/bin/bash
# SQ counters of the policy kernels (run from the repo root on the GPU box): two --pmc passes over tools/policy_bench.py, summaries under gpurun_out/pol_pmc/
set -e
R=$PWD
O=$R/gpurun_out/pol_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
pass() {
  local n=$1; shift
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/raw_$n -o p -- python3 $R/tools/policy_bench.py 4096 > $O/$n.log 2>&1
  python3 $R/tools/pmc_summary.py $(dirname $(ls $O/raw_$n/*/*_kernel_trace.csv $O/raw_$n/*_kernel_trace.csv 2>/dev/null | head -1)) p > $O/$n.txt 2>&1
  rm -rf "$R/gpurun_out/pol_pmc/raw_$n"
  head -4 $O/$n.txt | cut -c1-260
}
pass a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD
pass b SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY
