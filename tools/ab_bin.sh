for a in 0 1 2 3 4 8 15; do
  AMT_BIN_ABLATE=$a python bench.py --steps 40 --warmup 4 --cpu-rows 0 --streams 1 2>&1 | tail -1 > /tmp/line.json
  python -c "import json; d=json.load(open('/tmp/line.json')); print('ablate', $a, round(d['kernels']['k_bin_frame']['ms'],4))"
done
