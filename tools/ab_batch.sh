run() {
  timeout -s INT 120 python bench.py --steps ${STEPS:-90} --warmup 6 --cpu-rows 0 --plan ${PLAN:-fused} $EXTRA > /tmp/line.json 2> /tmp/err.txt || { echo "$1 FAILED"; tail -5 /tmp/err.txt; return; }
  tail -1 /tmp/line.json > /tmp/last.json
  python -c "import json; d=json.load(open('/tmp/last.json')); k=d['kernels']; print('$1', round(d['value']), round(d['ms_per_step'],4), round(k['k_georef_rows']['ms'],4), d['config']['single_pass_frames'], d['config']['frames_without_prepass'])"
}
for rep in 1 2; do
EXTRA="--batch 1" run batch1
EXTRA="--batch 2" run batch2
EXTRA="--batch 3" run batch3
done
EXTRA="--batch 3 --magnetic" run batch3-mag
