#!/usr/bin/env python3
"""Build profiles/traffic.json (HBM-side bytes per launch of the dominant kernel) from two rocprofv3
PMC passes of the bench command:

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d F -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d W -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline
    python profiles/make_traffic.py F/*/*counter_collection.csv W/*/*counter_collection.csv k_blur_solve 1920x1080 32 [tag]

profiles/traffic.json holds one entry per workload ("1920x1080_b32", "3840x2160_b32", "256x256_b256", ...).

The file records the signature of the device sources it was measured on (bench.kernel_signature()); bench.py
reports `roofline.traffic` only while that signature matches the sources it runs.

Units and gfx950 correction (MI355X_MICROARCH.md, HBM section, checked with profiles/tools/calib_fetch.hip
on this pool: 1 GiB read at 4/8/16 B per lane reports FETCH_SIZE = 524,300; 256 MiB written reports
WRITE_SIZE = 262,100): both counters are in KiB, FETCH_SIZE reports exactly half of a coalesced read
stream, WRITE_SIZE is exact.  bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024, averaged over every launch
of the kernel (all levels, fused and unfused variants), like roofline.achieved."""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def per_launch(path, kernel, counter):
    tot, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
            tot += float(r["Counter_Value"])
            n += 1
    return tot, n


def main():
    fcsv, wcsv, kernel, workload, batch = sys.argv[1:6]
    tag = sys.argv[6] if len(sys.argv) > 6 else "?"
    import bench
    f, nf = per_launch(fcsv, kernel, "FETCH_SIZE")
    w, nw = per_launch(wcsv, kernel, "WRITE_SIZE")
    assert nf == nw and nf > 0, (nf, nw)
    entry = {"workload": workload, "batch": int(batch), "kernel": kernel, "launches": nf,
             "fetch_size_kib_per_launch_raw": f / nf, "write_size_kib_per_launch": w / nw,
             "hbm_bytes_per_launch": (2.0 * f / nf + w / nw) * 1024.0,
             "kernel_signature": bench.kernel_signature(), "fuse_first": bench.FUSE_FIRST, "captured": tag}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "traffic.json")
    doc = {"workloads": {}}
    if os.path.exists(path):
        try:
            old = json.load(open(path))
            if "workloads" in old:
                doc = old
        except ValueError:
            pass
    doc["correction"] = "FETCH_SIZE x2 on gfx950 (calibrated, profiles/tools/calib_fetch.hip), WRITE_SIZE exact, KiB units"
    doc["workloads"][f"{workload}_b{int(batch)}"] = entry      # one entry per workload: bench.load_traffic() looks it up
    json.dump(doc, open(path, "w"), indent=1)
    print(json.dumps(entry))


if __name__ == "__main__":
    main()
