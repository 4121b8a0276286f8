# PMC counters of the row kernel on the three frames of tools/phase_probe.py (bench frame, sky, all Earth), for the shipped
# build and (if present) the timing-only skip build: where do the wave-cycles go?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3/pmc_phase
mkdir -p $O
for lib in shipped skip; do
  if [ $lib = skip ]; then export AMT_LIB_PATH=$R/build/libamt_skiptiming.so; [ -f $AMT_LIB_PATH ] || continue; fi
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout -s INT 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/${lib}_$i -- python3 $R/tools/phase_probe.py > $O/${lib}_$i.log 2>&1 < /dev/null
    echo "$lib set $i exit $?"
  done
done
python3 - <<PY
import csv, glob, collections
for lib in ('shipped', 'skip'):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    trace = {}
    for path in sorted(glob.glob('$O/%s_*/**/*kernel_trace.csv' % lib, recursive=True)):
        for r in csv.DictReader(open(path)):
            trace[(path.split('/')[-3] if False else path.rsplit('/', 2)[0], r['Dispatch_Id'])] = (r['Kernel_Name'], int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    for path in sorted(glob.glob('$O/%s_*/**/*counter_collection.csv' % lib, recursive=True)):
        rows = list(csv.DictReader(open(path)))
        # dispatches of k_georef_rows in order: phase_probe runs 18 georef-only + 18 fused per frame kind (3 warm-up + 15)
        seq = collections.OrderedDict()
        for r in rows:
            if 'k_georef_rows' not in r['Kernel_Name']:
                continue
            seq.setdefault(r['Dispatch_Id'], (r['Kernel_Name'], {}))[1][r['Counter_Name']] = float(r['Counter_Value'])
        ids = list(seq)
        for n, did in enumerate(ids):
            name, vals = seq[did]
            kind = ('bench', 'sky', 'earth')[min(2, n // 36)] + ('/fused' if '0, 2>' in name or '0, 1>' in name else '/georef')
            for c, v in vals.items():
                acc[kind][c].append(v)
    for kind in acc:
        print(lib, kind, '  '.join('%s=%.3g' % (c.replace('SQ_', ''), sum(v) / len(v)) for c, v in sorted(acc[kind].items())), 'n=%d' % len(next(iter(acc[kind].values()))))
PY
