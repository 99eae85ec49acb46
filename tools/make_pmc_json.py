#!/usr/bin/env python3
"""profiles/rN/pmc_traffic.json from the raw per-kernel counter sums tools/pmc_traffic.sh leaves in pmc_traffic_raw.json.
Units and corrections as MI355X_MICROARCH.md §HBM prescribes: counter values are KiB; on gfx950 FETCH_SIZE reports half
of the bytes of wide coalesced reads, so it is doubled; WRITE_SIZE is taken as read: exact for 16-B/lane stores and, by
tools/pmc_write_cal.sh (profiles/r2/write_size_cal.txt), also for 4-B/lane row stores and the GEMM epilogue's two-row store shape."""
import collections
import json
import sys

raw = json.load(open(sys.argv[1]))
rnd = sys.argv[2] if len(sys.argv) > 2 else "r2"
fam = collections.defaultdict(lambda: {"launches": 0, "fetch": 0.0, "write": 0.0})
per = {}
for tag, key in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    for name, v in raw[tag].items():
        n = name.replace("void ", "")
        f = n.split("<")[0].replace("dsg::", "")
        scale = 2048.0 if key == "fetch" else 1024.0
        per.setdefault(n, {"launches": v["launches"]})[key + "_bytes_per_launch"] = v["sum"] * scale / v["launches"]
        fam[f][key] += v["sum"] * scale
        if key == "fetch":
            fam[f]["launches"] += v["launches"]
out = {"source": f"tools/pmc_traffic.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `bench.py --no-graph --num-steps 3 "
                 f"--warmup 0 --no-cpu-baseline`, MI355X, round {rnd[1:]}",
       "units": "bytes; counter values are KiB; FETCH_SIZE doubled (gfx950 reports 1/2 of 16-B/lane coalesced reads, MI355X_MICROARCH.md §HBM); "
                "WRITE_SIZE as read (calibrated 1.0000 for 4-B/lane and 16-B/lane stores: profiles/r2/write_size_cal.txt)",
       "kernels": {}, "instantiations": {}}
for f, v in sorted(fam.items(), key=lambda kv: -(kv[1]["fetch"] + kv[1]["write"])):
    if v["launches"] == 0:
        continue
    out["kernels"][f] = {"launches": v["launches"], "fetch_bytes_per_launch": v["fetch"] / v["launches"],
                         "write_bytes_per_launch": v["write"] / v["launches"],
                         "hbm_bytes_per_launch": (v["fetch"] + v["write"]) / v["launches"]}
for n, v in per.items():
    if "fetch_bytes_per_launch" in v and "write_bytes_per_launch" in v:
        v["hbm_bytes_per_launch"] = v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]
        out["instantiations"][n] = v
json.dump(out, sys.stdout, indent=1)
