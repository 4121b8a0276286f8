# Round profile set: default bench line, rocprofv3 kernel stats of the same command, two-pass line for comparison
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
timeout -s INT 400 python3 $R/bench.py > $O/bench_default_n1.json 2> $O/bench_default_n1.err
tail -c 600 $O/bench_default_n1.json
timeout -s INT 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --cpu-rows 0 > $O/bench_under_rocprof.json 2> $O/rocprof.err
python3 $R/tools/trace_gaps.py $O/stats
timeout -s INT 200 python3 $R/bench.py --cpu-rows 0 --plan two-pass > $O/bench_two_pass_n1.json 2> /dev/null
cut -c1-200 $O/bench_two_pass_n1.json
