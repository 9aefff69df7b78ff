#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on the kernels' own access shapes (one gpurun call).  usage: bash profiles/r5_calib.sh
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r5/calib; rm -rf $O; mkdir -p $O $R/profiles/bin
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $R/profiles/calib_shapes.hip -o $R/profiles/bin/calib_shapes || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 $R/profiles/bin/calib_shapes > $O/known.txt || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $R/profiles/bin/calib_shapes > /dev/null || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- $R/profiles/bin/calib_shapes > /dev/null || exit 1
python3 $R/profiles/calib_shapes.py $O $O/calib_shapes.json
