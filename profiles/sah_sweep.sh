cd $GRAFT_REPO_ROOT
for tc in 200 100 50 0; do for nc in 400 700 1000; do
  python bench.py --steps 1 --warmup 1 --no-cpu --opt bvh_tri_cost_milli=$tc --opt bvh_node_cost_milli=$nc > gpurun_out/sah_${tc}_${nc}.json 2>>gpurun_out/sah.err && python -c "
import json; d=json.load(open('gpurun_out/sah_${tc}_${nc}.json')); r=d['roofline']; print('tri_cost $tc node_cost $nc: whole %.1f trace-only %.1f launch %.2f ms NV %.2f LV %.2f T %.2f nodes %d tris/leaf %.2f' % (d['value'], r['trace_Mrays_per_s'], r['avg_launch_ms'], r['node_visits_per_ray'], r['leaf_visits_per_ray'], r['tri_tests_per_ray'], d['config']['bvh_nodes'], r['wave_occupancy']['tris_per_leaf_visit']))"
done; done
