# clock (GRBM_GUI_ACTIVE / duration) and wait counters of the fused kernel per store variant of tools/store_probe.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${TAG:-pmc_store}
for set in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -s INT 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -- python3 $R/tools/store_probe.py > ${OUT}_$tag.log 2>&1
done
python3 - <<PY
import csv, glob
from collections import defaultdict
kt = {}
for f in glob.glob('$OUT/*/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_georef_rows' in r['Kernel_Name']:
            kt[r['Dispatch_Id']] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
cc = defaultdict(dict)
for f in glob.glob('$OUT/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_georef_rows' in r['Kernel_Name']:
            cc[r['Dispatch_Id']][r['Counter_Name']] = float(r['Counter_Value'])
ids = sorted(cc, key=int)
# 23 dispatches per variant (3 warm-up + 20 timed), five variants
for v in range(5):
    sel = ids[v * 23 + 3:(v + 1) * 23]
    if not sel: continue
    n = len(sel)
    dur = sum(kt[i] for i in sel) / n
    m = {k: sum(cc[i][k] for i in sel) / n for k in cc[sel[0]]}
    print('variant %d  %.1f us  clock %.0f MHz  valu_busy %.2f  wave_cycles %.3g  wait_inst %.3g (%.0f%%)  active_any %.3g  vmem_cycles %.3g  insts_valu %.3g' % (
        v, dur / 1e3, m['GRBM_GUI_ACTIVE'] / 8 / (dur / 1e3), 4 * m['SQ_ACTIVE_INST_VALU'] / (m['GRBM_GUI_ACTIVE'] / 8 * 1024),
        m['SQ_WAVE_CYCLES'], m['SQ_WAIT_INST_ANY'], 100 * m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES'], m['SQ_ACTIVE_INST_ANY'],
        m['SQ_INST_CYCLES_VMEM'], m['SQ_INSTS_VALU']))
PY
