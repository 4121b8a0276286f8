# Round-2 profile set (run through gpurun on one MI355X; copies of the results live in profiles/r2/):
#   a  default bench line (with the exact / configs[3] / grids-only variants and the CPU baseline)
#   b  rocprofv3 --kernel-trace --stats of the same command
#   c  the other workloads as bench lines of their own: --exact, --magnetic, --no-hints, --shared-image, --plan two-pass, --upload
#   d  rocprofv3 stats of --exact and --magnetic
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2
mkdir -p $O
timeout -s INT 400 python3 $R/bench.py > $O/a_bench_default_n1.json 2> $O/a_bench_default_n1.err
python3 - <<PY
import json; d=json.load(open('$O/a_bench_default_n1.json')); print('default', round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['ms_per_launch']/3,4), round(d['roofline']['frac'],3))
PY
timeout -s INT 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b_stats -- python3 $R/bench.py --cpu-rows 0 --no-variants > $O/b_bench_under_rocprof.json 2> $O/b_rocprof.err
python3 $R/tools/trace_gaps.py $O/b_stats | tail -8
for v in exact magnetic no-hints shared-image upload; do
  timeout -s INT 300 python3 $R/bench.py --cpu-rows 0 --no-variants --$v > $O/c_bench_${v}_n1.json 2> $O/c_bench_${v}.err
  python3 - <<PY
import json; d=json.load(open('$O/c_bench_${v}_n1.json')); print('$v', round(d['value']), round(d['ms_per_step'],4), round(d['kernels']['k_georef_rows']['ms'],4), round(d['roofline']['frac'],3))
PY
done
timeout -s INT 300 python3 $R/bench.py --cpu-rows 0 --no-variants --plan two-pass > $O/c_bench_two-pass_n1.json 2> $O/c_bench_two-pass.err
for v in exact magnetic; do
  timeout -s INT 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/d_stats_$v -- python3 $R/bench.py --cpu-rows 0 --no-variants --$v > $O/d_bench_${v}_under_rocprof.json 2> $O/d_rocprof_$v.err
done
find $O -name "*kernel_stats.csv" | head
# p  frames with a pole in view (single-pass pole plan vs two-pass), g  one rank with a process group (RCCL gather inside the timed region)
timeout 300 python3 $R/tools/pole_probe.py 2>&1 | grep "frame" > $O/p_pole_frames.txt
AMT_BENCH_DEBUG=1 AMT_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29571 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout -s INT 300 python3 $R/bench.py --gpus 1 --cpu-rows 0 --no-variants > $O/g_bench_one_rank_rccl.json 2> $O/g_bench_one_rank_rccl.err
grep "timed region" $O/g_bench_one_rank_rccl.err
cat $O/p_pole_frames.txt
