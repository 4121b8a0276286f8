# PMC passes for the two-pass kernels and for the MLat/MLT fused variant (same counter sets as pmc_fused.sh, fewer)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for variant in "two:--plan two-pass --streams 1" "mag:--magnetic"; do
  tag=${variant%%:*}; args=${variant#*:}
  OUT=$R/gpurun_out/pmc_$tag
  for set in "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
    t=$(echo $set | tr ' ' '_' | cut -c1-40)
    timeout -s INT 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$t -- python3 $R/bench.py --steps 6 --warmup 2 --cpu-rows 0 --batch 1 $args > ${OUT}_$t.log 2>&1
  done
  echo "== $tag"; python3 $R/profiles/summarize_pmc.py $OUT
done
