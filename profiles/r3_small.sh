cd $GRAFT_REPO_ROOT
for T in 16777216 200000000; do
export ART_SHADE_SMALL_MAX=$T
python bench.py --scene c3 --width 1024 --height 1024 --vthreads 16 --steps 4 --warmup 1 --no-cpu --no-counters 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$T c3', j['value'], j['ms_per_step'])"
python bench.py --no-cpu --no-counters --steps 1 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$T c4', j['value'], j['ms_per_step'])"
python bench.py --scene c5 --width 4096 --height 4096 --vthreads 8 --steps 4 --warmup 1 --no-cpu --no-counters 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$T c5', j['value'], j['ms_per_step'])"
done
