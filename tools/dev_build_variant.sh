#!/bin/bash
# developer A/B builds: tools/dev_build_variant.sh <name> <extra hipcc flags...>
# builds build/dev/<name>/liboffthip.so with only the kernels of offt_reg_dev.hip (the 1024 f64 defaults plus
# whatever -DOFFT_DEV_* switches select), for OFFT_AMD_LIB=build/dev/<name>/liboffthip.so
set -e
name=$1; shift
d=build/dev/$name; mkdir -p $d
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iofft_amd/csrc -Iinclude -Ibuild -DOFFT_DEV_REGISTRY"
/opt/rocm/bin/hipcc $FLAGS "$@" -c offt_amd/csrc/offt_kernels.hip -o $d/k.o &
/opt/rocm/bin/hipcc $FLAGS "$@" -c offt_amd/csrc/offt_reg_dev.hip -o $d/r.o -Rpass-analysis=kernel-resource-usage 2> $d/res.txt
wait
g++ -shared -o $d/liboffthip.so $d/k.o $d/r.o build/offt_host.o -Wl,--allow-shlib-undefined -ldl -lm -lpthread
rm -f $d/k.o $d/r.o
python tools/kernel_resources.py $d/res.txt 10
