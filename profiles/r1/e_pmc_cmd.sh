cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_r1e/$tag -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-rows 0 --streams 1 > $R/gpurun_out/pmc_r1e_$tag.log 2>&1
  tail -1 $R/gpurun_out/pmc_r1e_$tag.log | cut -c1-150
done
