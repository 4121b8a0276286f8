run() {
  timeout -s INT 120 python bench.py --steps ${STEPS:-60} --warmup 6 --cpu-rows 0 --plan ${PLAN:-fused} $EXTRA > /tmp/line.json 2> /tmp/err.txt || { echo "$1 FAILED"; return; }
  tail -1 /tmp/line.json > /tmp/last.json
  python -c "import json; d=json.load(open('/tmp/last.json')); k=d['kernels']; print('$1', round(d['value']), round(d['ms_per_step'],4), round(k['k_georef_rows']['ms'],4))"
}
for a in 0 25; do AMT_FUSED_ABLATE=$a run ablate$a; done
for a in 0 25; do AMT_LIB_PATH=$PWD/auromat_amd/lib/libauromat_hip_w1.so AMT_FUSED_ABLATE=$a run noimage-ablate$a; done
PLAN=two-pass EXTRA="--streams 1" run two-pass-1stream
