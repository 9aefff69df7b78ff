#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE calibration factors per access shape: counter bytes vs the known bytes profiles/calib_shapes.hip prints.
usage: python profiles/calib_shapes.py <dir with fetch/ write/ (rocprofv3 csv) and known.txt> <out.json>"""
import collections
import csv
import glob
import json
import re
import sys


def counters(d, name):
    rows = list(csv.DictReader(open(glob.glob(d + "/*/*_counter_collection.csv")[0])))
    val = collections.defaultdict(float); ns = {}
    for r in rows:
        if r["Counter_Name"] != name:
            continue
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        val[k] += float(r["Counter_Value"]) * 1024.0            # KiB
        ns[k] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return val, ns


def main(src, dst):
    known = {}
    for line in open(src + "/known.txt"):
        m = re.match(r"known (\S+) fetch (\d+) write (\d+)", line)
        if m:
            known[m.group(1)] = (float(m.group(2)), float(m.group(3)))
    fv, fns = counters(src + "/fetch", "FETCH_SIZE")
    wv, wns = counters(src + "/write", "WRITE_SIZE")
    out = {}
    for k, (kf, kw) in known.items():
        e = {"known_fetch_bytes": kf, "known_write_bytes": kw, "FETCH_SIZE_bytes": fv.get(k), "WRITE_SIZE_bytes": wv.get(k),
             "launch_ms": fns.get(k, 0) / 1e6}
        if kf and fv.get(k):
            e["fetch_factor_known_over_counter"] = kf / fv[k]
        if kw and wv.get(k):
            e["write_factor_known_over_counter"] = kw / wv[k]
        if fns.get(k):
            e["known_TBps"] = (kf + kw) / fns[k] / 1e3
        out[k] = e
    json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
    for k, e in out.items():
        print("%-14s fetch x%s write x%s  %.2f ms  %.2f TB/s of known bytes" % (k, ("%.3f" % e["fetch_factor_known_over_counter"]) if "fetch_factor_known_over_counter" in e else "-",
              ("%.3f" % e["write_factor_known_over_counter"]) if "write_factor_known_over_counter" in e else "-", e["launch_ms"], e.get("known_TBps", 0.0)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
