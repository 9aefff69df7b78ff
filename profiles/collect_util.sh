#!/bin/bash
# Utilisation counters of the vector-memory path (TA, vector L1 = TCP) and of LDS, next to the VALU counters of collect.sh: what
# k_trace_coop is limited by.  Few counters per pass: these blocks have two to four counter slots per instance.
#   usage: bash profiles/collect_util.sh <name> [bench.py workload options]     (after collect.sh <name>: same output directory)
NAME=${1:-c4}; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 1 --warmup 1 --no-cpu --no-counters $*"
pass() { n=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -- $B > $OUT/$n.log 2>&1 || { echo "pass $n failed"; tail -3 $OUT/$n.log; return 1; }; }
pass ta GRBM_GUI_ACTIVE TA_BUSY_avr TA_BUSY_max || exit 1
pass ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum || exit 1
pass tcp TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum || exit 1
pass tcp2 TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum || exit 1
pass tcp3 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum || exit 1
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD || exit 1
echo collected utilisation counters for $NAME
