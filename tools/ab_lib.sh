# Same-box A/B of two builds of the library: tools/ab_lib.sh <name of the other build (libauromat_hip_<name>.so)> <out file> [reps]
# alternates the builds; per build: the fused bench sequence (kernel_us.py), its MLat/MLT form, and the georef-only / directions-in
# kernels alone (dirs_in_probe.py)
other=$1; out=$2; reps=${3:-2}
mkdir -p "$(dirname "$out")"
: > "$out"
for rep in $(seq 1 $reps); do
  for lib in "$other" current; do
    if [ "$lib" = current ]; then unset AMT_LIB_PATH; else export AMT_LIB_PATH=$PWD/auromat_amd/lib/libauromat_hip_$lib.so; fi
    echo "== rep $rep build $lib" >> "$out"
    python tools/kernel_us.py >> "$out" 2>&1 || exit 1
    python tools/kernel_us.py magnetic >> "$out" 2>&1 || exit 1
    python tools/dirs_in_probe.py >> "$out" 2>&1 || exit 1
  done
done
