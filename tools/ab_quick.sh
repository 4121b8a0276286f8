# quick A/B of both plans (value, ms/step, georef kernel ms, bin kernel ms)
run() {
  timeout -s INT 120 python bench.py --steps ${STEPS:-80} --warmup 6 --cpu-rows 0 --no-variants --plan $1 $EXTRA > /tmp/line.json 2> /tmp/err.txt || { echo "$1 FAILED"; tail -3 /tmp/err.txt; return; }
  tail -1 /tmp/line.json > /tmp/last.json
  python -c "import json; d=json.load(open('/tmp/last.json')); k=d['kernels']; print('$1', round(d['value']), round(d['ms_per_step'],4), round(k['k_georef_rows']['ms'],4), k['k_bin_frame']['ms'] if isinstance(k['k_bin_frame'], dict) else '-', d['config']['single_pass_frames'])"
}
for rep in 1 2; do run fused; run two-pass; done
EXTRA="--streams 1" run two-pass
