#!/bin/bash
# Collects a round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   kernel-trace --stats of a bench workload, then separate --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ mix / L2 hits) -- never
#   combined with a trace domain other than --kernel-trace.  usage: bash profiles/collect.sh <name> [bench.py workload options]
#   e.g.  bash profiles/collect.sh c4        bash profiles/collect.sh c3 --scene c3 --width 1024 --height 1024
# Raw output goes to gpurun_out/prof_<name> (scratch); profiles/summarize.py turns it into the committed summaries.
NAME=${1:-c4}; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 1 --warmup 1 --no-cpu --no-counters $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/trace.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $B > $OUT/fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $B > $OUT/write.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/sq -- $B > $OUT/sq.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d $OUT/tcc -- $B > $OUT/tcc.log 2>&1 || exit 1
grep -h '"metric"' $OUT/trace.log | tail -1 > $OUT/bench_under_rocprof.json
echo collected $NAME
