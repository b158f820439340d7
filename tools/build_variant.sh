#!/bin/bash
# tools/build_variant.sh <name> <source.hip> <-D...>: an experimental libintfft with ONE translation unit rebuilt with extra defines
# (build/variants/libintfft_<name>.so; run with INTFFT_LIB=<that path>).  Diagnostics only.
set -e
NAME=$1; SRC=$2; shift 2
mkdir -p build/variants
OBJ=build/variants/${NAME}_$(basename $SRC .hip).o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c intfftk_amd/csrc/$SRC -o $OBJ
OBJS=$(ls intfftk_amd/lib/*.o | grep -v "/$(basename $SRC .hip).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/libintfft_$NAME.so $OBJS $OBJ
echo build/variants/libintfft_$NAME.so
