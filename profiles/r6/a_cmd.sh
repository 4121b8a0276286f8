# Round-6 profile set (gpurun calls on one MI355X each; tools/collect_r6.py copies the results into profiles/r6/ and refreshes
# profiles/traffic.json with the PMC traffic and the hash of the kernel sources it was measured on):
#   part 1   a   default bench line (variants incl. directions_in and upload, CPU baselines, parity)
#            a3  the driver's command (--steps 20 --warmup 5)
#            b   rocprofv3 --kernel-trace --stats of the default workload: stats CSV + the kernel trace of the timed region
#            e   PMC passes of the fused kernel (one counter set per run): traffic, busy / wait cycles, dynamic instruction mix
#   part 2   c   other workloads as bench lines of their own (--magnetic, --plan two-pass, --upload)
#            p   PMC traffic of the georef-only kernel + k_bin_frame (two-pass plan) and of the MLat/MLT variants
#            g   BASELINE configs[4] rehearsed with ONE rank over RCCL (32 frames, gather inside the timed region)
#            k   class route wall times
# usage (on the GPU box): bash tools/profile_r6.sh 1|2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6/final
mkdir -p $O
pmc() {   # pmc <out dir> <log prefix> <bench args...>: one rocprofv3 run per counter set
  local out=$1 log=$2; shift 2
  local i=0
  for set in "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout -s INT 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/set$i -- python3 $R/bench.py --steps 6 --warmup 2 --spinup-ms 0 --cpu-rows 0 --no-variants --batch 1 "$@" > ${log}_set$i.log 2>&1 < /dev/null
    echo "pmc $(basename $out) set $i exit $?"
  done
}
if [ "$1" = 1 ]; then
  timeout -s INT 500 python3 $R/bench.py > $O/a_bench_default_n1.json 2> $O/a_bench_default_n1.err
  echo "a done"
  timeout -s INT 300 python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/a3_bench_driver_command_steps20.json 2> /dev/null
  timeout -s INT 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b_stats -- python3 $R/bench.py --cpu-rows 0 --no-variants > $O/b_bench_under_rocprof.json 2> $O/b_rocprof.err
  echo "b done"
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS" "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout -s INT 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/e_pmc/set$i -- python3 $R/bench.py --steps 12 --warmup 3 --spinup-ms 0 --cpu-rows 0 --no-variants > $O/e_pmc_set$i.log 2>&1 < /dev/null
    echo "pmc set $i exit $?"
  done
  python3 $R/profiles/summarize_pmc.py $O/e_pmc > $O/e_pmc_summary_per_launch.txt
  tail -c 300 $O/a_bench_default_n1.json
else
  timeout -s INT 300 python3 $R/bench.py --cpu-rows 0 --no-variants --magnetic > $O/c_bench_magnetic_n1.json 2> /dev/null
  timeout -s INT 300 python3 $R/bench.py --cpu-rows 0 --no-variants --upload > $O/c_bench_upload_n1.json 2> /dev/null
  timeout -s INT 300 python3 $R/bench.py --cpu-rows 0 --no-variants --plan two-pass > $O/c_bench_two-pass_n1.json 2> /dev/null
  echo "c done"
  pmc $O/p_pmc_two $O/p_two --plan two-pass --streams 1
  pmc $O/p_pmc_magonly $O/p_magonly --magnetic
  pmc $O/p_pmc_nine $O/p_nine --magnetic --nine-arrays
  AMT_BENCH_DEBUG=1 AMT_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29571 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout -s INT 300 python3 $R/bench.py --gpus 1 --steps 32 --warmup 5 --cpu-rows 0 --no-variants > $O/g_bench_configs4_one_rank_rccl_steps32.json 2> $O/g_bench_configs4_one_rank_rccl.err
  echo "g done"
  timeout -s INT 200 python3 $R/tools/class_api_time.py > $O/k_class_api.txt 2>&1
  timeout -s INT 100 python3 $R/tools/class_host_steps.py >> $O/k_class_api.txt 2>&1
  timeout -s INT 400 python3 $R/tools/cubic_full_probe.py > $O/n_cubic_full_size.txt 2>&1
  tail -3 $O/k_class_api.txt
fi
