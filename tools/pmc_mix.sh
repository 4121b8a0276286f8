# Dynamic VALU instruction mix of the fused kernel (FP64 add / mul / fma / transcendental, integer, conversions)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${TAG:-pmc_mix}
for set in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64" "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -s INT 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-rows 0 --no-variants > ${OUT}_$tag.log 2>&1 < /dev/null
  echo "set $tag exit $?"
done
python3 $R/profiles/summarize_pmc.py $OUT < /dev/null | grep -A40 "k_georef_rows" | head -60
