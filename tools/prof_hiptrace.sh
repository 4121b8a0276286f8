cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -s INT 150 rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $R/gpurun_out/prof_hip -- python3 $R/bench.py --steps 30 --warmup 5 --cpu-rows 0 --plan ${1:-fused} > $R/gpurun_out/prof_hip.log 2>&1
tail -1 $R/gpurun_out/prof_hip.log | cut -c1-200
ls $R/gpurun_out/prof_hip/*/
