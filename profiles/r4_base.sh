# Round-4 baseline, one gpurun call: the new stated-spp tests, then per-kernel stats of C3 / C4 / C5 steps on the round-3 kernels.
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_gpu_stated_spp.py -m gpu -x -q > gpurun_out/r4/t_stated.log 2>&1 || { tail -30 gpurun_out/r4/t_stated.log; exit 1; }
tail -3 gpurun_out/r4/t_stated.log
bash profiles/kstats_ab.sh base > gpurun_out/r4/ks_c4.txt 2>&1; cat gpurun_out/r4/ks_c4.txt
cp gpurun_out/ks_base/trace/*/*kernel_stats.csv gpurun_out/r4/ks_c4.csv
bash profiles/kstats_ab.sh base --scene c3 --width 1024 --height 1024 --vthreads 16 > gpurun_out/r4/ks_c3.txt 2>&1; cat gpurun_out/r4/ks_c3.txt
cp gpurun_out/ks_base/trace/*/*kernel_stats.csv gpurun_out/r4/ks_c3.csv
bash profiles/kstats_ab.sh base --scene c5 --width 4096 --height 4096 --vthreads 8 > gpurun_out/r4/ks_c5.txt 2>&1; cat gpurun_out/r4/ks_c5.txt
cp gpurun_out/ks_base/trace/*/*kernel_stats.csv gpurun_out/r4/ks_c5.csv
