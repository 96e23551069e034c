#!/usr/bin/env python3
"""Combine the two `rocprofv3 --pmc` passes over tools/pmc_cycle.py into one record: HBM bytes per launch of the smoother's
residual kernel and of the water-column kernel beside their algorithmic bytes.

  python tools/pmc_cycle_to_json.py <fetch counter_collection.csv> <write counter_collection.csv> <log of pmc_cycle.py> <out.json>

FETCH_SIZE is doubled (gfx950 tallies 128-byte requests at 64 B, MI355X_MICROARCH.md HBM section); the factor is checked in
the same run against scale_to_kernel, a plain stream of n doubles in and n out."""
import csv
import json
import sys


def collect(path, counter):
    per = {}
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] == counter:
            per.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
    return per


fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
info = json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])
scale = [k for k in fetch if k.startswith("scale_to_kernel")][0]
cal_f = sum(fetch[scale]) / len(fetch[scale])
rec = {"workload": "320x384x60 upwind3+isop (K33), fine level, one colour", "n": info["n"], "nnz": info["nnz"],
       "calibration": {"kernel": "scale_to_kernel", "FETCH_SIZE_KB": cal_f, "algorithmic_read_KB": info["n"] * 8 / 1024.0,
                       "fetch_factor": info["n"] * 8 / 1024.0 / cal_f}, "kernels": []}
for prefix, key, ms in (("void csr_spmv_pipe_kernel<1, float", "smoother_spmv_bytes", "smoother_ms"),
                        ("void colblock_apply_ldspack_kernel<2", "column_solve_bytes", "column_ms")):
    names = [k for k in fetch if k.startswith(prefix)]
    if not names:
        raise SystemExit(f"kernel {prefix} not found in {list(fetch)[:10]}")
    k = names[0]
    f_kb = sum(fetch[k]) / len(fetch[k])
    w_kb = sum(write[k]) / len(write[k])
    traffic = int((2.0 * f_kb + w_kb) * 1024.0)
    rec["kernels"].append({"kernel": k[:70], "launches": len(fetch[k]), "FETCH_SIZE_KB_avg": f_kb, "WRITE_SIZE_KB_avg": w_kb,
                           "traffic_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": info[key],
                           "traffic_over_algorithmic": traffic / info[key], "avg_launch_ms_under_the_profiler_serialised": info[ms]})
json.dump(rec, open(sys.argv[4], "w"), indent=1)
print(json.dumps(rec["kernels"]))
