# PMC passes (one counter set per run) for the MLat/MLT-only fused variant (bench.py --magnetic, one frame per launch) and,
# beside it, the nine-array variant (--nine-arrays): HBM traffic per frame and VALU work
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for variant in "magonly:--magnetic" "nine:--magnetic --nine-arrays"; do
  tag=${variant%%:*}; args=${variant#*:}
  OUT=$R/gpurun_out/r4/pmc_$tag
  mkdir -p $OUT
  for set in "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
    t=$(echo $set | tr ' ' '_' | cut -c1-40)
    timeout -s INT 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$t -- python3 $R/bench.py --steps 6 --warmup 2 --spinup-ms 0 --cpu-rows 0 --no-variants --batch 1 $args > ${OUT}_$t.log 2>&1 < /dev/null
    echo "pmc $tag $t exit $?"
  done
  echo "== $tag" >> $R/gpurun_out/r4/e_pmc_magnetic_variants.txt; python3 $R/profiles/summarize_pmc.py $OUT >> $R/gpurun_out/r4/e_pmc_magnetic_variants.txt
done
cat $R/gpurun_out/r4/e_pmc_magnetic_variants.txt
