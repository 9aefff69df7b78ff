cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu --steps 1 --warmup 1 --vthreads 16 "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('$*', j['value'], r['trace_Mrays_per_s'], r['node_visits_per_ray'], r['leaf_visits_per_ray'], r['tri_tests_per_ray'], j['config']['bvh_nodes'])"; }
run --opt bvh_builder=0
run --opt bvh_builder=0 --opt bvh_spatial_splits=1
run --opt bvh_builder=0
run --opt bvh_builder=0 --opt bvh_spatial_splits=1
