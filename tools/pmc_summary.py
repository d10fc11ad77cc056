"""Sum rocprofv3 --pmc counter CSVs per kernel: python tools/pmc_summary.py <dir> [kernel-substring]
Prints {counter: total / n_dispatches} for the dispatches whose kernel name contains the substring."""
import csv
import glob
import json
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else "dd_"
    tot = defaultdict(float)
    disp = defaultdict(set)
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if pat not in row.get("Kernel_Name", ""):
                continue
            name = row["Counter_Name"]
            tot[name] += float(row["Counter_Value"])
            disp[name].add((f, row.get("Dispatch_Id")))
    print(json.dumps({k: {"per_dispatch": tot[k] / max(1, len(disp[k])), "dispatches": len(disp[k])} for k in sorted(tot)}, indent=1))


if __name__ == "__main__":
    main()
