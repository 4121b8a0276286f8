# A/B of library builds: georef-only kernel on the Earth frame and on a frame of sky (tools/sky_probe.py) + fused kernel alone
libs="default $@"
for rep in 1 2; do
for v in $libs; do
  if [ $v = default ]; then unset AMT_LIB_PATH; else export AMT_LIB_PATH=$PWD/auromat_amd/lib/libauromat_hip_$v.so; fi
  echo "$v $(timeout 100 python tools/sky_probe.py 2>&1 | grep frame | awk '{printf "%s=%s ", $1, $5}') fused=$(timeout 100 python tools/kernel_alone.py 2>&1 | grep 'fuse True' | awk '{print $6}')"
done
done
