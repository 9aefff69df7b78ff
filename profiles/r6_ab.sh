#!/bin/bash
# Round 6 A/B (profiles/r5_ab.sh with the round-6 output directory) on one box: per-kernel average durations (rocprofv3 --kernel-trace --stats) + whole-job rate of variants on C4 / C3 / C5 / S4.
# A variant is  tag[:lib-suffix][:opt=val,opt=val]   e.g.  "split  nosplit::shade_split=0  w8:w8"
# usage: bash profiles/r5_ab.sh "<variants>" [out tag] [scenes] [repetitions]
# The variants are run round-robin REPS times (a b c a b c): a stage kernel's average moved by up to 8 % between two processes of the SAME
# library on one box (ab1 / ab2: 4.30 vs 4.66 ms), so one run per variant cannot rank variants that differ by less.
VARS=${1:-"split nosplit::shade_split=0"}; TAG=${2:-ab}; SCENES=${3:-"c4 c3 c5 s4"}; REPS=${4:-1}
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/r6
cd /tmp && export TMPDIR=/tmp
{
for sc in $SCENES; do
  case $sc in
    c4) ARGS="" ;;
    c3) ARGS="--scene c3 --width 1024 --height 1024 --vthreads 16" ;;
    c5) ARGS="--scene c5 --width 4096 --height 4096 --vthreads 8" ;;
    s4) ARGS="--scene s4" ;;
  esac
  for rep in $(seq 1 $REPS); do
  for v in $VARS; do
    IFS=: read tag suf opts <<< "$v"
    lib=$R/ada-ray-tracer_amd/libart_hip.so; [ -n "$suf" ] && lib=$R/ada-ray-tracer_amd/libart_hip_$suf.so
    O=""; for kv in ${opts//,/ }; do O="$O --opt $kv"; done
    d=$R/gpurun_out/r6/ks_${TAG}_${sc}_${tag}_$rep; rm -rf $d; mkdir -p $d
    ART_LIB=$lib timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d/trace -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-counters $ARGS $O > $d/log 2>&1
    python3 - $d $sc $tag <<PY
import csv,glob,sys,json
d,sc,v=sys.argv[1:4]
f=glob.glob(d+"/trace/*/*kernel_stats.csv")[0]
L=[json.loads(l) for l in open(d+"/log") if l.startswith('{"metric"')][-1]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
def nm(r): return r["Name"].split("(")[0].replace("art::","").replace("void ","").replace("k_shade_compact","ksc").replace("k_trace_coop","ktc")
print(sc, v, "whole %.1f ms/step %.1f w/t %s |" % (L["value"], L["ms_per_step"], round(L["trace_ms_per_step"]/L["ms_per_step"],4)), " | ".join("%s %.3f ms x%s (%.1f%%)" % (nm(r)[:22], float(r["AverageNs"])/1e6, r["Calls"], 100*float(r["TotalDurationNs"])/tot) for r in rows[:8]))
PY
    cp $d/trace/*/*kernel_stats.csv $R/gpurun_out/r6/kstats_${TAG}_${sc}_${tag}_$rep.csv 2>/dev/null
    rm -rf $d/trace
  done
  done
done
} 2>&1 | tee $R/gpurun_out/r6/$TAG.txt
