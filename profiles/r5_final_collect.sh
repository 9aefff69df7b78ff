# Round-5 evidence.  bash profiles/r5_final_collect.sh <part>   (raw output under gpurun_out/; summaries made afterwards in the build container
# by profiles/valu_mix.py + profiles/summarize.py)
#   part 1: kernel stats + PMC passes (trace kernel AND stages) of C4, S4 (the review's item 3), C5 and C3
#   part 2: every BASELINE configuration at full size, the structured and the instanced scene, the driver's command line, and the
#           stage's cumulative A/B against the round-4 kernel (libart_hip_r4k.so = the round-4 k_shade_compact: option shade_split=0 of a
#           build of the round-4 source)
set -e
cd $GRAFT_REPO_ROOT
case "$1" in
1)
  bash profiles/collect.sh c4 && bash profiles/collect_util.sh c4
  bash profiles/collect.sh s4 --scene s4 && bash profiles/collect_util.sh s4 --scene s4
  bash profiles/collect.sh c5 --scene c5 --width 4096 --height 4096 --vthreads 8 && bash profiles/collect_util.sh c5 --scene c5 --width 4096 --height 4096 --vthreads 8
  bash profiles/collect.sh c3 --scene c3 --width 1024 --height 1024 --vthreads 16 && bash profiles/collect_util.sh c3 --scene c3 --width 1024 --height 1024 --vthreads 16
  ;;
2)
  mkdir -p gpurun_out/cfg5
  python bench.py > gpurun_out/cfg5/c4.json 2> gpurun_out/cfg5/c4.err
  python bench.py --scene c2 --width 512 --height 512 --vthreads 4 --steps 1 --warmup 1 > gpurun_out/cfg5/c2.json 2> gpurun_out/cfg5/c2.err
  python bench.py --scene c3 --width 1024 --height 1024 --vthreads 16 --steps 4 --warmup 1 > gpurun_out/cfg5/c3.json 2> gpurun_out/cfg5/c3.err
  python bench.py --scene c5 --width 4096 --height 4096 --vthreads 8 --steps 32 --warmup 1 > gpurun_out/cfg5/c5.json 2> gpurun_out/cfg5/c5.err
  python bench.py --scene s4 --no-cpu > gpurun_out/cfg5/s4.json 2> gpurun_out/cfg5/s4.err
  python bench.py --scene i64 --vthreads 16 > gpurun_out/cfg5/i64.json 2> gpurun_out/cfg5/i64.err
  python bench.py --host-buffers --no-cpu > gpurun_out/cfg5/c4_host.json 2> gpurun_out/cfg5/c4_host.err
  python bench.py --no-cpu --contexts 8 > gpurun_out/cfg5/c4_ctx8.json 2> gpurun_out/cfg5/c4_ctx8.err
  python bench.py --steps 20 --warmup 5 > gpurun_out/cfg5/c4_steps20_warmup5.json 2> gpurun_out/cfg5/c4_steps20.err
  for f in gpurun_out/cfg5/*.json; do python -c "
import json,sys
d=json.load(open('$f')); r=d.get('roofline') or {}; w=d.get('whole_job_roofline') or {}; print('$f', d['value'], d['ms_per_step'], r.get('frac'), r.get('whole_over_trace_only'), w.get('frac'), (d.get('cpu_baseline') or {}).get('value'))"; done
  bash profiles/r5_ab.sh "r4k:r4k:shade_split=0 final" ab_final "c4 c3 c5 s4" 2 | cut -c1-260
  ;;
esac
