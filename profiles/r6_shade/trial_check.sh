R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r6
for i in 1 2 3; do
for sc in c4 c5 s4 c3; do
  case $sc in
    c3) A="--scene c3 --width 1024 --height 1024 --vthreads 16 --steps 4 --warmup 3" ;;
    c4) A="--steps 2 --warmup 1" ;;
    c5) A="--scene c5 --width 4096 --height 4096 --vthreads 8 --steps 4 --warmup 1" ;;
    s4) A="--scene s4 --steps 2 --warmup 1" ;;
  esac
  for per in 0 4 2; do
    ART_DEBUG_ADDR=1 python3 $R/bench.py $A --no-cpu --no-counters --opt shade_per=$per > /tmp/t.json 2> /tmp/t.err
    python3 - $sc $per $i <<PY
import json,sys
sc,per,i=sys.argv[1:4]
L=[json.loads(l) for l in open("/tmp/t.json") if l.startswith('{"metric"')][-1]
t=[l.strip() for l in open("/tmp/t.err") if "trial" in l]
print(sc, "shade_per", per, "run", i, "shade %.3f ms/batch  %.1f Mrays/s" % (L["stages"]["shade"]["ms_per_batch"], L["value"]), "|", t[-1][15:] if t else "", flush=True)
PY
  done
done
done
