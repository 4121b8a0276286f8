# A/B of bench.py flag sets on one box, interleaved twice.  usage: bash tools/ab_flags.sh "" "--no-hints" "--warmup 12" ...
for rep in 1 2; do
for v in "$@"; do
  timeout -s INT 200 python bench.py --cpu-rows 0 --no-variants $v > /tmp/line.json 2> /tmp/err.txt
  python -c "import json; d=json.load(open('/tmp/line.json')); print('[$v]', round(d['value']), round(d['ms_per_step'],4), round(d['kernels']['k_georef_rows']['ms'],4), d['config']['frames_without_prepass'], d['config']['single_pass_frames'])"
done
done
