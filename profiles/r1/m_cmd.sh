# rocprofv3 kernel stats of the widened paths (tools/time_widened.py) -> gpurun_out/prof_widened2/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_widened2
mkdir -p $O
timeout -s INT 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/tools/time_widened.py > $O.log 2>&1 < /dev/null
echo "exit $?"
f=$(find $O -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then head -16 "$f" | cut -c1-200; else echo "no stats file"; tail -5 $O.log; fi
