"""tools/pmc_table.py <dir with rocprofv3 --pmc outputs> [substring filter] -> per-kernel mean of every counter found (JSON on stdout).
Generic companion of tools/pmc_sets.sh (any program, any counter sets)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""


def short(name):
    return name.split("(")[0].replace("void ", "").replace("intfft::", "")


acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = short(r["Kernel_Name"])
        if flt in n:
            acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {n: {c: sum(v) / len(v) for c, v in d.items()} for n, d in acc.items()}
for n, d in out.items():
    d["_launches"] = max(len(v) for v in acc[n].values())
print(json.dumps(out, indent=1, sort_keys=True))
