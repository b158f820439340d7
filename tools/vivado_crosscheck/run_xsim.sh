#!/bin/bash
# run_xsim.sh -- one command for the owner of a Vivado install: simulate the reference RTL on the kit's stimulus and
# compare with the GPU engine's expected dumps.  Needs: Vivado (xvhdl / xelab / xsim on PATH; 2018.1 is what the
# reference project names, vivado/intfftk.xpr:2) and a checkout of hukenovs/intfftk.
#
#   INTFFTK_DIR=/path/to/intfftk  tools/vivado_crosscheck/run_xsim.sh
#
# UNTESTED in this repository's build image (no Vivado there); the steps are the standard xsim flow.
set -euo pipefail
KIT=$(cd "$(dirname "$0")" && pwd)
: "${INTFFTK_DIR:?set INTFFTK_DIR to a checkout of hukenovs/intfftk}"
WORK=${WORK:-$PWD/xsim_crosscheck}
mkdir -p "$WORK" && cd "$WORK"

# every synthesisable VHDL source of the reference (not its testbenches), then the kit's four testbenches (integer text I/O up to 32 bits, hex text I/O beyond)
find "$INTFFTK_DIR/src/vhdl" -name '*.vhd' ! -path '*/tb/*' | sort > sources.f
xvhdl -work work $(cat sources.f) "$KIT/tb_single_dump.vhd" "$KIT/tb_pair_dump.vhd" "$KIT/tb_single_hex.vhd" "$KIT/tb_pair_hex.vhd"

status=0
run_case() { # tb case mode nfft format rndmode stimulus data_width twdl_width xseries
    local tb=$1 case=$2 mode=$3 nfft=$4 fmt=$5 rnd=$6 stim=$7 dw=$8 tw=$9 xs=${10} infile outfile
    if [ "$tb" = tb_single_dump ]; then infile=IN_FILE; else infile=IN_FILE; fi
    outfile="$WORK/${case}_${mode}_rtl.dat"   # (hex cases: the same name, hex words inside -- compare.py reads the manifest)
    xelab -L unisim -L unimacro work.$tb -s snap_${case}_${mode} \
        -generic_top "NFFT=$nfft" -generic_top "FORMAT=$fmt" -generic_top "RNDMODE=$rnd" \
        -generic_top "DATA_WIDTH=$dw" -generic_top "TWDL_WIDTH=$tw" -generic_top "XSERIES=$xs" \
        -generic_top "$infile=$KIT/expected/$stim" -generic_top "OUT_FILE=$outfile"
    xsim snap_${case}_${mode} -runall
    python3 "$KIT/compare.py" "$case" "$mode" "$outfile" || status=1
}
python3 - "$KIT/expected/manifest.json" <<'PY' > cases.txt
import json, sys
for c in json.load(open(sys.argv[1]))["cases"]:
    print(c["tb"], c["case"], c["mode"], c["nfft"], c["format"], c["rndmode"], c["stimulus"], c.get("data_width", 16), c.get("twdl_width", 16),
          c.get("xser", "NEW"))
PY
while read -r tb case mode nfft fmt rnd stim dw tw xs; do run_case "$tb" "$case" "$mode" "$nfft" "$fmt" "$rnd" "$stim" "$dw" "$tw" "$xs"; done < cases.txt
if [ $status -eq 0 ]; then echo "ALL PASS: the GPU engine is bit-exact to this RTL simulation on every case"; else echo "SOME CASES FAILED"; fi
exit $status
