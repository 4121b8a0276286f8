# Where do the row kernel's wave cycles go?  SQ / SQC counters of the fused kernel (tools/kernel_us.py) and of the same kernel
# without coordinate arrays ("grids only": the arithmetic, the image loads and the binning alone), one rocprofv3 --pmc pass per set.
# -> gpurun_out/r5/wc_summary.txt (profiles/r5/e_pmc_wave_cycles.txt)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5/wc
mkdir -p $O
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM" \
  "SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VMEM_WR" \
  "SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_STALL" \
  "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQC_DCACHE_BUSY_CYCLES" \
  "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_CMD_FIFO_FULL"; do
  i=$((i+1))
  for mode in fused grids; do
    arg=""; [ $mode = grids ] && arg="keep=0"
    timeout -s INT 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/$mode$i -- python3 $R/tools/kernel_us.py $arg > $O/$mode$i.log 2>&1 < /dev/null
    echo "pmc set $i $mode exit $?"
  done
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob('$O/**/*counter_collection.csv', recursive=True):
    mode = 'grids only (no coordinate arrays)' if '/grids' in path else 'fused (five coordinate arrays)'
    for r in csv.DictReader(open(path)):
        if 'k_georef_rows' not in r['Kernel_Name'] or int(r['Grid_Size']) < 2000000:
            continue                                   # three-frame launches only
        acc[mode][r['Counter_Name']].append(float(r['Counter_Value']))
with open('$O/../wc_summary.txt', 'w') as fp:
    for k in sorted(acc):
        fp.write('k_georef_rows, ' + k + ', three frames per launch\n')
        for c, v in sorted(acc[k].items()):
            fp.write('   %-36s mean per launch %.5g  (n=%d)\n' % (c, sum(v) / len(v), len(v)))
print(open('$O/../wc_summary.txt').read())
PY
