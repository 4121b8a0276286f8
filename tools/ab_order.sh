run() {
  timeout -s INT 120 python bench.py --steps ${STEPS:-80} --warmup 6 --cpu-rows 0 --plan $1 $EXTRA > /tmp/line.json 2> /tmp/err.txt || { echo "$1 FAILED"; tail -3 /tmp/err.txt; return; }
  tail -1 /tmp/line.json > /tmp/last.json
  python -c "import json; d=json.load(open('/tmp/last.json')); k=d['kernels']; print('$2', round(d['value']), round(d['ms_per_step'],4), round(k['k_georef_rows']['ms'],4))"
}
for rep in 1 2; do
AMT_CHUNK_ORDER=0 run fused order0
AMT_CHUNK_ORDER=1 run fused order1
done
AMT_CHUNK_ORDER=0 EXTRA="--streams 1" run two-pass plain-order0
AMT_CHUNK_ORDER=1 EXTRA="--streams 1" run two-pass plain-order1
