#!/bin/bash
# tools/build_variant_multi.sh <name> "<-D flags>" <source.hip> [more sources]: an experimental libintfft with SEVERAL translation units rebuilt with
# extra defines (build/variants/libintfft_<name>.so; run with INTFFT_LIB=<that path>).  Diagnostics only.
set -e
NAME=$1; FLAGS=$2; shift 2
mkdir -p build/variants
declare -A REBUILT
NEW=()
for SRC in "$@"; do
  B=$(basename $SRC .hip)
  OBJ=build/variants/${NAME}_$B.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $FLAGS -c intfftk_amd/csrc/$SRC -o $OBJ &
  REBUILT[$B]=1
  NEW+=($OBJ)
done
wait
OBJS=()
for O in intfftk_amd/lib/*.o; do
  B=$(basename $O .o)
  [ -z "${REBUILT[$B]:-}" ] && OBJS+=($O)
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/libintfft_$NAME.so "${OBJS[@]}" "${NEW[@]}"
echo build/variants/libintfft_$NAME.so
