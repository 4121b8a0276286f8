# A/B: fused single-pass plan vs two-pass plan (each run bounded by `timeout`)
for plan in fused two-pass fused two-pass; do
  timeout -s INT 120 python -X faulthandler bench.py --steps ${STEPS:-60} --warmup 6 --cpu-rows 0 --plan $plan > /tmp/line.json 2> /tmp/err.txt || { echo "$plan FAILED rc=$?"; tail -25 /tmp/err.txt; continue; }
  tail -1 /tmp/line.json > /tmp/last.json
  python -c "import json; d=json.load(open('/tmp/last.json')); k=d['kernels']; print('$plan', round(d['value']), round(d['ms_per_step'],4), round(k['k_georef_rows']['ms'],4), k['k_bin_frame']['ms'] if isinstance(k['k_bin_frame'], dict) else '-', round(d['roofline']['frac'],3))"
done
