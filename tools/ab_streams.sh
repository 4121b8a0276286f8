# A/B: one vs two HIP streams in bench.py
for s in 1 2 1 2; do
  python bench.py --steps 60 --warmup 6 --cpu-rows 0 --streams $s 2>&1 | tail -1 > /tmp/line.json
  python -c "import json; d=json.load(open('/tmp/line.json')); print('streams', $s, round(d['value']), round(d['ms_per_step'],4), round(d['kernels']['k_georef_rows']['ms'],4), round(d['kernels']['k_bin_frame']['ms'],4))"
done
