#!/bin/bash
# a variant of libudcore.so: one source recompiled with extra -D flags, linked with the objects of the default build
#   bash profiles/tools/mkvariant.sh <tag> <source.hip> [-DFLAG=..] ...   -> u-dales_amd/lib/libudcore_<tag>.so
set -e
TAG=$1; SRC=$2; shift 2
cd "$(dirname "$0")/../../u-dales_amd/csrc"
make -j8 > /dev/null
O=/tmp/variant_${TAG}_${SRC%.hip}.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -Wno-unused-variable "$@" -c $SRC -o $O
OBJS=$(ls udc_*.o | grep -v udc_comm_test.o | grep -v "^${SRC%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o ../lib/libudcore_$TAG.so $OBJS $O -L/opt/rocm/lib -lrocfft -lrccl -lpthread -Wl,-rpath,/opt/rocm/lib
echo built ../lib/libudcore_$TAG.so
