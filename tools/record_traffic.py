"""Condense the two rocprofv3 --pmc passes of `bench.py` (FETCH_SIZE, WRITE_SIZE: they do not
fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots") into the record `bench.py` reports as
`roofline.traffic`, stamped with the hash of the kernel's source so that the number goes null
instead of stale when the kernel changes.

usage: python tools/record_traffic.py <fetch.csv> <write.csv> <size> [out.json]
       (the CSVs are outputs of tools/summarize_pmc.py)

FETCH_SIZE is in KB and reports exactly half of the bytes on gfx950 (the guide: 128-byte
requests tallied at 64 B).  Round 3 measured it for both load shapes of this library with a
kernel that reads a known 1.0737 GB once (tools/micro/fetch_calib.hip,
profiles/r03_fetch_calibration.csv): 524 310 KB for one dword per lane and 256-byte row pieces
-- the tile visit's shape -- and 524 298 KB for 16-byte vectors: x2.00 for both.  (Rounds 1-2
used x1.605, from a window-load benchmark whose overlapping windows were partly served by L2.)
WRITE_SIZE (KB) is exact."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

FETCH_FACTOR = 2.0


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
                     f"FETCH_SIZE x{FETCH_FACTOR} (gfx950: tools/micro/fetch_calib.hip)"}
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
