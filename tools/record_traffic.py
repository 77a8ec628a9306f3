"""Condense the two rocprofv3 --pmc passes of `bench.py` (FETCH_SIZE, WRITE_SIZE: they do not
fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots") into the record `bench.py` reports as
`roofline.traffic`, stamped with the hash of the kernel's source so that the number goes null
instead of stale when the kernel changes.

usage: python tools/record_traffic.py <fetch.csv> <write.csv> <size> [out.json]
       (the CSVs are outputs of tools/summarize_pmc.py)

FETCH_SIZE is in KB and under-reports on gfx950: x2 for 16-byte-per-lane streaming reads
(the guide), x1.605 for the 4-byte-per-lane row loads of the tile visit -- calibrated on a kernel
of the same access shape that reads a known 1.0737 GB; WRITE_SIZE (KB) is exact."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

FETCH_FACTOR = 1.605


def per_dispatch(path, counter):
    for row in csv.DictReader(open(path)):
        if bench.FILL_KERNEL in row["Kernel_Name"] and row["Counter_Name"] == counter:
            return float(row["Per_Dispatch"]) * 1024.0, int(row["Dispatches"])
    raise SystemExit(f"{path}: no {counter} row of {bench.FILL_KERNEL}")


def main():
    fetch_csv, write_csv, size = sys.argv[1], sys.argv[2], int(sys.argv[3])
    out = sys.argv[4] if len(sys.argv) > 4 else bench.TRAFFIC_RECORD
    fetch, n_f = per_dispatch(fetch_csv, "FETCH_SIZE")
    write, n_w = per_dispatch(write_csv, "WRITE_SIZE")
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"],
                          capture_output=True, text=True).stdout.strip() or None
    rec = {"kernel": bench.FILL_KERNEL, "size": size, "head": head,
           "kernel_hash": bench.kernel_source_hash(),
           "fetch_bytes_raw": fetch, "write_bytes": write, "fetch_factor": FETCH_FACTOR,
           "bytes_per_launch": fetch * FETCH_FACTOR + write,
           "dispatches": [n_f, n_w],
           "source": f"{os.path.relpath(fetch_csv, ROOT)}, {os.path.relpath(write_csv, ROOT)}: "
                     f"separate rocprofv3 --pmc passes of bench.py, bytes per launch, "
                     f"FETCH_SIZE x{FETCH_FACTOR} (gfx950 calibration, 4-byte-per-lane loads)"}
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
