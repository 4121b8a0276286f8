cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_short
timeout -s INT 150 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_short -- python3 $R/bench.py --steps 20 --warmup 5 --cpu-rows 0 --no-variants --spinup-ms 100 > $R/gpurun_out/prof_short.log 2>&1
tail -1 $R/gpurun_out/prof_short.log | cut -c1-200
python3 $R/tools/trace_timeline.py $R/gpurun_out/prof_short 8
