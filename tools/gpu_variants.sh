#!/bin/bash
# time the pre-filter kernel of several library builds (rocprofv3 kernel trace): bash tools/gpu_variants.sh lib lib_vA ...
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for l in "$@"; do
  rm -rf gpurun_out/pv; MREFSR_HIP_LIB=$GRAFT_REPO_ROOT/mrefsr_amd/$l/libmrefsr_hip.so timeout 300 rocprofv3 --kernel-trace -d gpurun_out/pv -o v -- python3 tools/corr_time.py 3 > gpurun_out/pv.log 2>&1
  echo "$l: $(grep 'corr call' gpurun_out/pv.log | sed 's/.*corr call/corr call/')"
  python3 tools/rocpd_stats.py gpurun_out/pv/v_results.db --top 8 | grep -E "corr_" | cut -c1-130
done
