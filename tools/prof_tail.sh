cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_tail
timeout -s INT 150 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_tail -- python3 $R/tools/tail_probe.py > $R/gpurun_out/prof_tail.log 2>&1
grep -v Warn $R/gpurun_out/prof_tail.log | tail -9
python3 $R/tools/trace_timeline.py $R/gpurun_out/prof_tail 8
