# A/B of library builds through bench.py, interleaved on one box: bash tools/ab_bench.sh "<bench flags>" name [name ...]
# ("default" = the in-tree library, other names = tools/build_variant.sh builds)
flags="$1"; shift
for rep in 1 2; do
for v in default "$@"; do
  if [ $v = default ]; then unset AMT_LIB_PATH; else export AMT_LIB_PATH=$PWD/auromat_amd/lib/libauromat_hip_$v.so; fi
  timeout -s INT 200 python bench.py --cpu-rows 0 --no-variants $flags 2>/dev/null | tail -1 > /tmp/last.json
  python -c "import json; d=json.load(open('/tmp/last.json')); print('$v', '[$flags]', round(d['value']), round(d['ms_per_step'],4), round(d['kernels']['k_georef_rows']['ms'],4))"
done
done
