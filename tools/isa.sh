#!/bin/bash
# tools/isa.sh <source.hip> [extra -D flags]: gfx950 disassembly + resource usage of one translation unit -> /tmp/isa/<name>.s, /tmp/isa/<name>.res
set -e
SRC=$1; shift
N=$(basename $SRC .hip)
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-function "$@" --cuda-device-only -c intfftk_amd/csrc/$SRC -o /tmp/isa/$N.co -Rpass-analysis=kernel-resource-usage 2> /tmp/isa/$N.rpass || { cat /tmp/isa/$N.rpass | grep -v remark | head -30; exit 1; }
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=/tmp/isa/$N.co --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=/tmp/isa/$N.elf
/opt/rocm/lib/llvm/bin/llvm-objdump -d /tmp/isa/$N.elf --no-show-raw-insn > /tmp/isa/$N.s
grep -E "Function Name|VGPRs:|Spill|ScratchSize" /tmp/isa/$N.rpass | sed 's/.*remark: //; s/\[-Rpass.*//' | paste - - - - - | sed 's/ \+/ /g' > /tmp/isa/$N.res
echo /tmp/isa/$N.s
