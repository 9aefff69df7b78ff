# Round-6 evidence.  bash profiles/r6_final_collect.sh <part>   (raw output under gpurun_out/, summaries made afterwards in the build
# container by profiles/valu_mix.py + profiles/summarize.py)
#   part 1: kernel stats + PMC passes of C4 (trace kernel AND the stages) with the utilisation counters; kernel stats + PMC of C3, C5, S4
#   part 2: every BASELINE configuration at full size, the structured and the instanced scene, the driver's command line, host buffers,
#           eight contexts on the one GPU (the multi_gpu object with the per-device busy / idle / start-skew times)
set -e
cd $GRAFT_REPO_ROOT
case "$1" in
1)
  bash profiles/collect.sh c4 && bash profiles/collect_util.sh c4
  bash profiles/collect.sh c3 --scene c3 --width 1024 --height 1024 --vthreads 16
  bash profiles/collect.sh c5 --scene c5 --width 4096 --height 4096 --vthreads 8
  bash profiles/collect.sh s4 --scene s4
  ;;
2)
  mkdir -p gpurun_out/cfg6
  python bench.py > gpurun_out/cfg6/c4.json 2> gpurun_out/cfg6/c4.err
  python bench.py --steps 20 --warmup 5 > gpurun_out/cfg6/c4_steps20_warmup5.json 2> gpurun_out/cfg6/c4_steps20.err
  python bench.py --scene c2 --width 512 --height 512 --vthreads 4 --steps 1 --warmup 1 > gpurun_out/cfg6/c2.json 2> gpurun_out/cfg6/c2.err
  for k in 2 3 4 5; do python bench.py --scene c2 --width 512 --height 512 --vthreads 4 --steps 1 --warmup 1 --no-cpu > gpurun_out/cfg6/c2_rep$k.json 2> gpurun_out/cfg6/c2_rep$k.err; done      # (C2 is a 1.6 ms render: its rate is launch latency, run to run)
  python bench.py --scene c3 --width 1024 --height 1024 --vthreads 16 --steps 4 --warmup 1 > gpurun_out/cfg6/c3.json 2> gpurun_out/cfg6/c3.err
  python bench.py --scene c5 --width 4096 --height 4096 --vthreads 8 --steps 32 --warmup 1 > gpurun_out/cfg6/c5.json 2> gpurun_out/cfg6/c5.err
  python bench.py --scene s4 --no-cpu > gpurun_out/cfg6/s4.json 2> gpurun_out/cfg6/s4.err
  python bench.py --scene i64 --vthreads 16 --steps 4 --warmup 1 > gpurun_out/cfg6/i64.json 2> gpurun_out/cfg6/i64.err
  python bench.py --host-buffers --no-cpu > gpurun_out/cfg6/c4_host.json 2> gpurun_out/cfg6/c4_host.err
  python bench.py --no-cpu --contexts 8 > gpurun_out/cfg6/c4_ctx8.json 2> gpurun_out/cfg6/c4_ctx8.err
  python bench.py --no-cpu --contexts 2 > gpurun_out/cfg6/c4_ctx2.json 2> gpurun_out/cfg6/c4_ctx2.err
  for f in gpurun_out/cfg6/*.json; do python -c "
import json,sys
d=json.load(open('$f')); r=d.get('roofline') or {}; print('$f', d['value'], d['ms_per_step'], r.get('frac'), r.get('trace_Mrays_per_s'), r.get('whole_over_trace_only'), d['config'].get('end_to_end_s'), (d.get('cpu_baseline') or {}).get('value'))"; done
  ;;
esac
