# Kernel traces of the bench loop with the frame loop in the library (AMT_SEQ_NATIVE=1) and in Python (=0): the gaps between
# consecutive launches of the big kernel in the timed region
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3/gaps
mkdir -p $O
for cfg in "AMT_SEQ_NATIVE=1" "AMT_SEQ_NATIVE=0" "AMT_SEQ_NATIVE=1 AMT_NO_SKY_PATH=1" "AMT_SEQ_NATIVE=1 AMT_ITEM_ORDER=1" "AMT_SEQ_NATIVE=1 AMT_ITEM_ORDER=2"; do
  v=$(echo $cfg | tr " =" "__")
  env $cfg timeout -s INT 200 rocprofv3 --kernel-trace --output-format csv -d $O/native$v -- python3 $R/bench.py --cpu-rows 0 --no-variants --steps 96 --spinup-ms 100 > $O/native$v.json 2> $O/native$v.err
  python3 - <<PY
import csv, glob, json
rows = []
for r in csv.DictReader(open(glob.glob('$O/native$v/**/*kernel_trace.csv', recursive=True)[0])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
big = [r for r in rows if 'k_georef_rows' in r[2]]
last = big[-33:]          # 96 frames = 1 + 31 x 3 + 2
gaps = [(b[0] - a[1]) / 1e3 for a, b in zip(last, last[1:])]
dur = [(e - s) / 1e3 for s, e, _ in last]
print('$cfg  launches', len(last), 'kernel us/launch mean %.1f' % (sum(dur[1:-1]) / len(dur[1:-1])), 'gap us mean %.1f min %.1f max %.1f' % (sum(gaps) / len(gaps), min(gaps), max(gaps)))
print('   gaps', ' '.join('%.0f' % g for g in gaps))
# what runs inside the gaps
t0, t1 = last[0][0], last[-1][1]
other = {}
for s, e, n in rows:
    if t0 <= s <= t1 and 'k_georef_rows' not in n:
        k = n.split('(')[0][-40:]
        other.setdefault(k, [0, 0.0]); other[k][0] += 1; other[k][1] += (e - s) / 1e3
for k, (c, d) in sorted(other.items(), key=lambda kv: -kv[1][1]): print('   %-42s n %4d  total us %8.1f' % (k, c, d))
d = json.loads(open('$O/native$v.json').read().strip().splitlines()[-1]); print('   line:', round(d['value']), round(d['ms_per_step'], 4), round(d['kernels']['k_georef_rows']['ms'], 4))
PY
done
