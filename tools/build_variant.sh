#!/bin/bash
# build_variant.sh NAME SRC.hip "FLAGS": liblhrs_NAME.so = the in-tree objects with SRC compiled under extra FLAGS (kernel experiments; select with LHRS_HIP_LIB)
set -e
cd "$(dirname "$0")/../lhrs_bot_amd/csrc"
NAME=$1; SRC=$2; FLAGS=$3
OBJ=${SRC%.hip}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $FLAGS -c $SRC -o var_${NAME}_${OBJ}.o
OBJS=$(ls *.o | grep -v '^var_' | grep -v "^${OBJ}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o liblhrs_${NAME}.so $OBJS var_${NAME}_${OBJ}.o
echo built liblhrs_${NAME}.so
