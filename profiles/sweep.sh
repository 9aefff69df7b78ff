#!/bin/bash
# A/B helper for the GPU box (run through gpurun from the repo root): one bench.py run per argument, each argument a
# space-separated list of art_set_option name=value pairs.  usage: profiles/sweep.sh "<opts A>" "<opts B>" ...   each a space-separated list of name=value
i=0
for o in "$@"; do
  args=""; for kv in $o; do args="$args --opt $kv"; done
  timeout -k 10 200 python bench.py --no-cpu --no-counters $args > gpurun_out/sw_$i.json 2>>gpurun_out/sw.err || { echo "FAILED: $o"; exit 1; }
  python - "$o" gpurun_out/sw_$i.json <<PY
import json,sys
d=json.load(open(sys.argv[2])); print("%-70s %8.1f  build %.0f ms nodes %d" % (sys.argv[1], d["value"], d["config"]["bvh_build_ms"], d["config"]["bvh_nodes"]))
PY
  i=$((i+1))
done
