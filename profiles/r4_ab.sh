#!/bin/bash
# A/B on one box: per-kernel stats + whole-job rate of library variants on C4 / C3 / C5.  usage: bash profiles/r4_ab.sh "<variants>" [tag]
VARS=${1:-"r3 base"}; TAG=${2:-ab}
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/r4
{
bash $R/profiles/kstats_ab.sh "$VARS"
bash $R/profiles/kstats_ab.sh "$VARS" --scene c3 --width 1024 --height 1024 --vthreads 16
bash $R/profiles/kstats_ab.sh "$VARS" --scene c5 --width 4096 --height 4096 --vthreads 8
} 2>&1 | tee $R/gpurun_out/r4/$TAG.txt
