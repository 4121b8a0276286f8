# Write-path counters of the fused frame kernel (and, for comparison, of torch's fill kernel that bench.py runs for
# measured_fill_GBs): are the stores leaving L2 as full 64-byte requests, and how often is the fabric pushing back?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5/wp
mkdir -p $O
i=0
for set in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum GRBM_GUI_ACTIVE" "MemUnitStalled TA_BUSY_avr SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_WRITE_DRAM_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s INT 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/set$i -- python3 $R/bench.py --steps 12 --warmup 3 --spinup-ms 0 --cpu-rows 0 --no-variants > $O/set$i.log 2>&1 < /dev/null
  echo "pmc set $i exit $?"
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob('$O/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        name = r['Kernel_Name']
        key = 'k_georef_rows (fused, 3 frames per launch)' if 'k_georef_rows' in name else ('torch fill (5 x 96 MB per call of measured_fill_gbs)' if 'FillFunctor' in name else None)
        if key is None:
            continue
        if key.startswith('k_georef') and int(r['Grid_Size']) < 2000000:
            continue                                   # the one-frame launch at the start of a call
        acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
with open('$O/../wp_summary.txt', 'w') as fp:
    for k in acc:
        fp.write(k + '\n')
        for c, v in sorted(acc[k].items()):
            fp.write('   %-40s mean per launch %.5g  (n=%d)\n' % (c, sum(v) / len(v), len(v)))
print(open('$O/../wp_summary.txt').read())
PY
