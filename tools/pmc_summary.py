#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; own runs).

    python tools/pmc_summary.py gpurun_out/pmc1 profiles/r03_pmc_traffic <signature>

`signature` (bench.py: "<workload>/b<batch>/r<rois>/<variant>") keys the table inside the JSON ({"signatures": {sig: {"kernels":
...}}}: several configurations in one file); bench.py only quotes a traffic number whose signature equals the configuration
it is running.  The .txt next to it gets one section per signature.

Units / corrections per /opt/skills/guides/MI355X_MICROARCH.md §HBM: both counters are in KiB;
on gfx950 FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced reads, so the read side is
doubled before it is compared with byte counts; WRITE_SIZE is used as reported (it matched the
known output size of the RoiPool kernel exactly: 58 800 KiB = R*49*512*8 B).
"""
import json
import sqlite3
import sys


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, count(*), avg(value), min(value), max(value) from counters_collection "
                     "where counter_name=? group by kernel_name", (counter,)).fetchall()
    return {r[0]: dict(calls=r[1], avg=r[2], min=r[3], max=r[4]) for r in rows}


def main(src, dst):
    f = per_kernel(f"{src}/fetch/r_results.db", "FETCH_SIZE")
    w = per_kernel(f"{src}/write/r_results.db", "WRITE_SIZE")
    out = {}
    lines = ["%-60s %6s %14s %14s %16s" % ("kernel", "calls", "read_MB(x2)", "write_MB", "hbm_MB/launch")]
    for k in sorted(set(f) | set(w), key=lambda k: -(2 * f.get(k, {}).get("avg", 0) + w.get(k, {}).get("avg", 0))):
        rd = 2.0 * f.get(k, {}).get("avg", 0.0) * 1024 / 1e6
        wr = w.get(k, {}).get("avg", 0.0) * 1024 / 1e6
        out[k.split("(")[0]] = dict(read_MB_corrected=round(rd, 3), write_MB=round(wr, 3), hbm_bytes_per_launch=int((rd + wr) * 1e6),
                                    fetch_kib_raw=f.get(k, {}), write_kib_raw=w.get(k, {}))
        lines.append("%-60s %6d %14.3f %14.3f %16.3f" % (k[:60], f.get(k, w.get(k))["calls"], rd, wr, rd + wr))
    sig = sys.argv[3] if len(sys.argv) > 3 else ""
    import os
    table = json.load(open(dst + ".json")) if os.path.exists(dst + ".json") else {"signatures": {}}
    table["signatures"][sig] = {"kernels": out}
    open(dst + ".json", "w").write(json.dumps(table, indent=1))
    open(dst + ".txt", "a").write("# signature %s\n# rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate runs), bench.py --launch eager --streams 1 "
                                  "--steps 2 --batches-per-step 64 (tools/gpu_pmc.sh)\n"
                                  "# FETCH_SIZE doubled (gfx950 wide-read correction, MI355X_MICROARCH.md); KiB -> MB\n" % sig + "\n".join(lines) + "\n\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
