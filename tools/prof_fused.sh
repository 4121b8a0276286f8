cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -s INT 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_fused -- python3 $R/bench.py --steps 60 --warmup 6 --cpu-rows 0 --no-variants --spinup-ms 0 --plan ${1:-fused} > $R/gpurun_out/prof_fused.log 2>&1
tail -1 $R/gpurun_out/prof_fused.log | cut -c1-300
python3 $R/tools/trace_gaps.py $R/gpurun_out/prof_fused
