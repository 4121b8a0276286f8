# Build the current tree into auromat_amd/lib/libauromat_hip_<name>.so (for A/B runs with AMT_LIB_PATH); extra hipcc flags follow
name=$1; shift
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared "$@" -Wl,-rpath,/opt/rocm/lib -o auromat_amd/lib/libauromat_hip_$name.so auromat_amd/csrc/*.hip
