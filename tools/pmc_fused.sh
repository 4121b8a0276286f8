# PMC counters of the fused kernel (one counter set per pass; --pmc only with --kernel-trace)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${TAG:-pmc_fused}
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS" "SQ_LDS_BANK_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -s INT 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -- python3 $R/bench.py --steps 6 --warmup 3 --spinup-ms 0 --cpu-rows 0 --no-variants --plan ${PLAN:-fused} > ${OUT}_$tag.log 2>&1
done
python3 $R/profiles/summarize_pmc.py $OUT
