cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu --no-counters --steps 1 --warmup 1 --vthreads 16 "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', j['value'])"; }
run
run --opt ray_chunk=32
run --opt ray_chunk=64
run --opt ray_chunk=96
run --opt ray_chunk=128
run --opt node_min=3
run --opt node_min=5
run --opt refill_min=1
run --opt refill_min=3
run --opt refill_min=4
run --opt queue_segments=4
run
