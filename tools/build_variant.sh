#!/bin/bash
# build_variant.sh <name> <kernel> "<extra hipcc flags>"  -> llzlab_amd/libllz_var_<name>.so
# One kernel file recompiled with extra -D flags, linked with the other objects of the regular build (for same-box A/B
# runs through LLZ_LIB; the variant libraries are git-ignored).
set -e
cd "$(dirname "$0")/../llzlab_amd/csrc"
name=$1; kern=$2; flags=$3
make -s
extra=""
[ "$kern" = fir_td ] && extra="-fno-slp-vectorize"
[ "$kern" = fft ] && extra="-mllvm -pragma-unroll-threshold=262144"      # as csrc/Makefile (HIPFLAGS_fft)
/opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -std=c++17 -Wno-unused-function $extra $flags -c kernels/$kern.hip -o build/var_$name.o
objs=$(ls build/host_*.o build/hip_*.o | grep -v "hip_$kern.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libllz_var_$name.so $objs build/var_$name.o -lm
echo built llzlab_amd/libllz_var_$name.so
