# Round-4 profile set (ONE gpurun call on one MI355X; tools/collect_r4.py copies the results into profiles/r4/):
#   a   default bench line (variants, CPU baselines, parity)          a3  the driver's command (--steps 20 --warmup 5)
#   b   rocprofv3 --kernel-trace --stats of the default command: stats CSV + the kernel trace of the timed region
#   c   the other workloads as bench lines of their own (--magnetic, --magnetic --nine-arrays, --exact, --plan two-pass, --upload)
#   e   PMC passes of the fused kernel (one counter set per run): traffic, busy / wait cycles, dynamic instruction mix
#   g   BASELINE configs[4] rehearsed with ONE rank: a process group over RCCL, 32 frames, gather inside the timed region
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4/final
mkdir -p $O
timeout -s INT 500 python3 $R/bench.py > $O/a_bench_default_n1.json 2> $O/a_bench_default_n1.err
echo "a done"
timeout -s INT 300 python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/a3_bench_driver_command_steps20.json 2> /dev/null
timeout -s INT 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b_stats -- python3 $R/bench.py --cpu-rows 0 --no-variants > $O/b_bench_under_rocprof.json 2> $O/b_rocprof.err
echo "b done"
timeout -s INT 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b_stats_magnetic -- python3 $R/bench.py --cpu-rows 0 --no-variants --magnetic > $O/b_bench_magnetic_under_rocprof.json 2> $O/b_rocprof_magnetic.err
timeout -s INT 300 python3 $R/bench.py --cpu-rows 0 --no-variants --magnetic > $O/c_bench_magnetic_n1.json 2> /dev/null
timeout -s INT 300 python3 $R/bench.py --cpu-rows 0 --no-variants --magnetic --nine-arrays > $O/c_bench_magnetic_nine_arrays_n1.json 2> /dev/null
for v in exact upload; do
  timeout -s INT 300 python3 $R/bench.py --cpu-rows 0 --no-variants --$v > $O/c_bench_${v}_n1.json 2> /dev/null
done
timeout -s INT 300 python3 $R/bench.py --cpu-rows 0 --no-variants --plan two-pass > $O/c_bench_two-pass_n1.json 2> /dev/null
echo "c done"
AMT_BENCH_DEBUG=1 AMT_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29571 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout -s INT 300 python3 $R/bench.py --gpus 1 --steps 32 --warmup 5 --cpu-rows 0 --no-variants > $O/g_bench_configs4_one_rank_rccl_steps32.json 2> $O/g_bench_configs4_one_rank_rccl.err
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS" "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s INT 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/e_pmc/set$i -- python3 $R/bench.py --steps 12 --warmup 3 --spinup-ms 0 --cpu-rows 0 --no-variants > $O/e_pmc_set$i.log 2>&1 < /dev/null
  echo "pmc set $i exit $?"
done
python3 $R/profiles/summarize_pmc.py $O/e_pmc > $O/e_pmc_summary_per_launch.txt
tail -c 300 $O/a_bench_default_n1.json
