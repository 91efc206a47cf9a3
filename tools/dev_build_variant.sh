#!/bin/bash
# developer A/B builds: tools/dev_build_variant.sh <name> <extra hipcc flags...>
# builds build/dev/<name>/liboffthip.so with only the 1024 kernels (fast), for OFFT_AMD_LIB=...
set -e
name=$1; shift
d=build/dev/$name; mkdir -p $d
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Ioffamd -Ioffamd/csrc -Ioffamd -Iofft_amd/csrc -Iinclude -DOFFT_DEV_ONLY_1024 "$@" -c offt_amd/csrc/offt_kernels.hip -o $d/k.o -Rpass-analysis=kernel-resource-usage 2> $d/res.txt
g++ -shared -o $d/liboffthip.so $d/k.o build/offt_host.o -Wl,--allow-shlib-undefined -ldl -lm -lpthread
python tools/kernel_resources.py $d/res.txt 1024
