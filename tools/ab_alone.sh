# kernel-alone A/B of library builds (tools/build_variant.sh names; "default" = in-tree): fused + georef-only kernel, 2 rounds
libs="default $@"
for rep in 1 2; do
for v in $libs; do
  if [ $v = default ]; then unset AMT_LIB_PATH; else export AMT_LIB_PATH=$PWD/auromat_amd/lib/libauromat_hip_$v.so; fi
  echo "$v $(timeout 100 python tools/kernel_alone.py 2>&1 | grep fuse | awk '{printf "%s=%s/%s ", $2, $6, $10}')"
done
done
