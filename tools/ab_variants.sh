# A/B of compile-time / env variants of the fused kernel: prints value, ms/step, kernel ms
run() {
  timeout -s INT 120 python bench.py --steps ${STEPS:-80} --warmup 6 --cpu-rows 0 --plan ${PLAN:-fused} > /tmp/line.json 2> /tmp/err.txt || { echo "$1 FAILED"; tail -5 /tmp/err.txt; return; }
  tail -1 /tmp/line.json > /tmp/last.json
  python -c "import json; d=json.load(open('/tmp/last.json')); k=d['kernels']; print('$1', round(d['value']), round(d['ms_per_step'],4), round(k['k_georef_rows']['ms'],4), k['k_bin_frame']['ms'] if isinstance(k['k_bin_frame'], dict) else '-')"
}
for rep in 1 2 3; do
  run default
  AMT_LIB_PATH=$PWD/auromat_amd/lib/libauromat_hip_w1.so run w1
  PLAN=two-pass run two-pass
  PLAN=two-pass AMT_LIB_PATH=$PWD/auromat_amd/lib/libauromat_hip_w1.so run two-pass-w1
done
