"""Average rocprofv3 --pmc counter CSVs per dispatch of one kernel.
  python tools/pmc_summary.py <dir> <kernel-substring> [bench.json]
With bench.json (the un-profiled line of the same configuration) the output carries pairs_per_launch and the derived
figures bench.py and profiles/README quote: raw HBM bytes (FETCH_SIZE + WRITE_SIZE are in KB), the FETCH-doubled bound
MI355X_MICROARCH.md prescribes for wide coalesced reads, VALU instructions per pair and VALU busy (SQ_ACTIVE_INST_VALU and
SQ_WAVE_CYCLES are both in quad-cycles; busy = active / wave-cycles x resident waves per SIMD is left to the reader)."""
import csv
import glob
import json
import sys
from collections import defaultdict


def main():
    root, pat = sys.argv[1], sys.argv[2]
    tot, disp, names = defaultdict(float), defaultdict(set), set()
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if pat not in row.get("Kernel_Name", ""):
                continue
            names.add(row["Kernel_Name"].split("(")[0])
            tot[row["Counter_Name"]] += float(row["Counter_Value"])
            disp[row["Counter_Name"]].add((f, row.get("Dispatch_Id")))
    cfg = {k: tot[k] / max(1, len(disp[k])) for k in sorted(tot)}
    out = {"kernel": sorted(names), "note": "rocprofv3 --pmc passes, one counter set per run with --kernel-trace only; "
           "averages over the launches of each run"}
    if len(sys.argv) > 3:
        b = json.load(open(sys.argv[3]))
        pairs = b["config"]["pairs_per_gpu"]
        cfg["pairs_per_launch"] = pairs
        cfg["workload"] = b["config"]["workload"]
        if "FETCH_SIZE" in cfg and "WRITE_SIZE" in cfg:
            cfg["hbm_bytes_raw"] = (cfg["FETCH_SIZE"] + cfg["WRITE_SIZE"]) * 1024.0
            cfg["hbm_bytes_fetch_doubled"] = (2 * cfg["FETCH_SIZE"] + cfg["WRITE_SIZE"]) * 1024.0
        cfg["algorithmic_bytes"] = b["roofline"]["algorithmic_bytes_per_launch"]
        if "SQ_INSTS_VALU" in cfg:
            cfg["valu_instr_per_pair"] = cfg["SQ_INSTS_VALU"] / pairs
            cfg["salu_instr_per_pair"] = cfg.get("SQ_INSTS_SALU", 0) / pairs
            cfg["lds_instr_per_pair"] = cfg.get("SQ_INSTS_LDS", 0) / pairs
        if "SQ_ACTIVE_INST_VALU" in cfg and "SQ_WAVE_CYCLES" in cfg:
            cfg["valu_active_over_wave_cycles"] = cfg["SQ_ACTIVE_INST_VALU"] / cfg["SQ_WAVE_CYCLES"]
        if "SQ_LDS_BANK_CONFLICT" in cfg and "SQ_LDS_IDX_ACTIVE" in cfg:
            cfg["lds_bank_conflict_frac"] = cfg["SQ_LDS_BANK_CONFLICT"] / cfg["SQ_LDS_IDX_ACTIVE"]
        cfg["kernel_ms_unprofiled"] = b["roofline"]["kernel_ms"]
    out["config"] = cfg
    try:                                          # which sources the counters belong to (bench.py quotes them only for that kernel)
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from dindel_tgi_amd import capi
        out["source_id"] = capi.kernel_source_id(pat)
    except Exception as e:                        # noqa: BLE001
        out["source_id"] = "unknown: %r" % (e,)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
