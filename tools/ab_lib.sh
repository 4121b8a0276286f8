# A/B of library builds on one box, interleaved: kernel alone (tools/kernel_alone.py) and the pipelined bench
# usage: bash tools/ab_lib.sh base [other ...]   (names of tools/build_variant.sh builds; "default" = the in-tree lib)
libs="default $@"
for rep in 1 2; do
for v in $libs; do
  if [ $v = default ]; then unset AMT_LIB_PATH; else export AMT_LIB_PATH=$PWD/auromat_amd/lib/libauromat_hip_$v.so; fi
  a=$(timeout 100 python tools/kernel_alone.py 2>&1 | grep "fuse True" | awk '{print $6}')
  timeout -s INT 120 python bench.py --steps 80 --warmup 6 --cpu-rows 0 --no-variants > /tmp/line.json 2> /tmp/err.txt
  tail -1 /tmp/line.json > /tmp/last.json
  python -c "import json; d=json.load(open('/tmp/last.json')); print('$v', 'alone_ms', round(float('$a'),4), 'bench', round(d['value']), round(d['ms_per_step'],4), round(d['kernels']['k_georef_rows']['ms'],4))"
done
done
