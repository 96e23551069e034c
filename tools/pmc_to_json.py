#!/usr/bin/env python3
"""Combine the two `rocprofv3 --pmc` passes over tools/pmc_spmv.py (FETCH_SIZE, WRITE_SIZE: separate runs, as the TCC
counter budget demands) into the JSON record bench.py reads for `roofline.traffic`.

  python tools/pmc_to_json.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>

FETCH_SIZE is doubled (gfx950 tallies 128-byte requests at 64 B, MI355X_MICROARCH.md HBM section); the factor is
checked in the same run against scale_to_kernel, a plain 16 B/lane stream of n doubles in and n out."""
import csv
import json
import sys


def collect(path, counter):
    per = {}
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        per.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
    return per


fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
spmv = [k for k in fetch if k.startswith("void csr_spmv_pipe_kernel<0, double")]
scale = [k for k in fetch if k.startswith("scale_to_kernel")]
if not spmv or not scale:
    raise SystemExit(f"kernels not found: {list(fetch)[:8]}")
n, nnz = 4229814, 73405094
f_kb = sum(fetch[spmv[0]]) / len(fetch[spmv[0]])
w_kb = sum(write[spmv[0]]) / len(write[spmv[0]])
cal_f = sum(fetch[scale[0]]) / len(fetch[scale[0]])
cal_w = sum(write[scale[0]]) / len(write[scale[0]])
alg_read_kb = n * 8 / 1024.0
factor = alg_read_kb / cal_f
traffic = int((2.0 * f_kb + w_kb) * 1024.0)
alg = 12 * nnz + 4 * (n + 1) + 16 * n
rec = {"kernel": spmv[0][:60], "workload": "320x384x60 upwind3+isop (K33)", "n": n, "nnz": nnz,
       "FETCH_SIZE_KB_avg": f_kb, "WRITE_SIZE_KB_avg": w_kb, "launches": len(fetch[spmv[0]]),
       "calibration": {"kernel": "scale_to_kernel (16 B/lane stream of n doubles)", "FETCH_SIZE_KB": cal_f, "WRITE_SIZE_KB": cal_w,
                       "algorithmic_read_KB": alg_read_kb, "fetch_factor": factor},
       "fetch_correction": "x2 (gfx950 FETCH_SIZE tallies 128-B requests at 64 B, MI355X_MICROARCH.md HBM section; checked by the calibration kernel of the same run)",
       "traffic_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": traffic / alg}
json.dump(rec, open(sys.argv[3], "w"), indent=1)
print(json.dumps(rec))
