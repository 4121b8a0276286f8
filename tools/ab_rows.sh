run() {
  timeout -s INT 120 python bench.py --steps ${STEPS:-80} --warmup 6 --cpu-rows 0 --plan ${PLAN:-fused} > /tmp/line.json 2> /tmp/err.txt || { echo "$1 FAILED"; tail -5 /tmp/err.txt; return; }
  tail -1 /tmp/line.json > /tmp/last.json
  python -c "import json; d=json.load(open('/tmp/last.json')); k=d['kernels']; print('$1', round(d['value']), round(d['ms_per_step'],4), round(k['k_georef_rows']['ms'],4))"
}
for r in 16 8 12 20 24 32 16; do AMT_GEOREF_ROWS=$r run rows$r; done
