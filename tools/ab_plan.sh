# A/B: fused single-pass plan vs two-pass plan
for plan in fused two-pass fused two-pass; do
  python bench.py --steps 60 --warmup 6 --cpu-rows 0 --plan $plan 2>&1 | tail -1 > /tmp/line.json
  python -c "import json; d=json.load(open('/tmp/line.json')); k=d['kernels']; print('$plan', round(d['value']), round(d['ms_per_step'],4), round(k['k_georef_rows']['ms'],4), k['k_bin_frame']['ms'] if isinstance(k['k_bin_frame'], dict) else '-', round(d['roofline']['frac'],3))"
done
