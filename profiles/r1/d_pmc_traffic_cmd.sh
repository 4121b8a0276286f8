# HBM traffic counters, one counter set per pass (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_r1d/$set -- python3 $R/bench.py --steps 8 --warmup 2 --cpu-rows 0 > $R/gpurun_out/pmc_r1d_$set.log 2>&1
  tail -1 $R/gpurun_out/pmc_r1d_$set.log | cut -c1-120
done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r1e -- python3 $R/bench.py --steps 30 --warmup 5 --cpu-rows 0 > $R/gpurun_out/prof_r1e.log 2>&1
tail -1 $R/gpurun_out/prof_r1e.log | cut -c1-120
