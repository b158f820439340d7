#!/bin/bash
# tools/build_variant_multi.sh <name> "<-D flags>" <source.hip> [more sources]: an experimental libintfft with SEVERAL translation units rebuilt with
# extra defines (build/variants/libintfft_<name>.so; run with INTFFT_LIB=<that path>).  Diagnostics only.
set -e
NAME=$1; FLAGS=$2; shift 2
mkdir -p build/variants
OBJS=$(ls intfftk_amd/lib/*.o)
for SRC in "$@"; do
  OBJ=build/variants/${NAME}_$(basename $SRC .hip).o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $FLAGS -c intfftk_amd/csrc/$SRC -o $OBJ &
  OBJS=$(echo "$OBJS" | grep -v "/$(basename $SRC .hip).o")
  OBJS="$OBJS $OBJ"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/libintfft_$NAME.so $OBJS
echo build/variants/libintfft_$NAME.so
