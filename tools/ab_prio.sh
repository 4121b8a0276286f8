run() {
  timeout -s INT 120 python bench.py --steps ${STEPS:-80} --warmup 6 --cpu-rows 0 --plan ${PLAN:-fused} $EXTRA > /tmp/line.json 2> /tmp/err.txt || { echo "$1 FAILED"; tail -5 /tmp/err.txt; return; }
  tail -1 /tmp/line.json > /tmp/last.json
  python -c "import json; d=json.load(open('/tmp/last.json')); k=d['kernels']; print('$1', round(d['value']), round(d['ms_per_step'],4), round(k['k_georef_rows']['ms'],4))"
}
for rep in 1 2; do
EXTRA="--batch 2" run high-b2
AMT_AUX_PRIO=0 EXTRA="--batch 2" run normal-b2
AMT_AUX_PRIO=1 EXTRA="--batch 2" run low-b2
EXTRA="--batch 1" run high-b1
AMT_AUX_PRIO=0 EXTRA="--batch 1" run normal-b1
AMT_AUX_PRIO=1 EXTRA="--batch 1" run low-b1
done
