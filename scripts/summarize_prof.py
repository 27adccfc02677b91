#!/usr/bin/env python
"""Summarise rocprofv3 output directories (csv or rocpd sqlite) into the small text / json files kept under profiles/.

    python scripts/summarize_prof.py trace <dir>            per-kernel calls / total ms / mean us / min / max / vgpr / lds
    python scripts/summarize_prof.py pmc <dir> [<dir> ...]  per-kernel mean of every counter found (one --pmc pass per dir)
    python scripts/summarize_prof.py pmc-json <out.json> <dir> [<dir> ...]   the same as json {kernel: {counter: mean, "n": dispatches}}
"""
import csv
import glob
import json
import os
import sqlite3
import sys
from collections import defaultdict


def _csvs(d, suffix):
    return sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True))


def trace(d):
    rows = defaultdict(list)
    meta = {}
    for f in _csvs(d, "kernel_trace.csv"):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                name = r.get("Kernel_Name", "?")
                rows[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
                meta[name] = (r.get("VGPR_Count", r.get("Arch_VGPR_Count", "")), r.get("Accum_VGPR_Count", ""), r.get("LDS_Block_Size", ""))
    if not rows:
        for db in sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True)):
            c = sqlite3.connect(db)
            try:
                for r in c.execute("select name, duration/1e3, vgpr_count, lds_size from kernels"):
                    rows[str(r[0])].append(float(r[1])); meta[str(r[0])] = (r[2], "", r[3])
            except sqlite3.Error as e:   # schema differs between ROCm versions
                print("# %s: %s" % (db, e))
    out = ["== rocprofv3 --kernel-trace: per-kernel calls, total ms, mean us, min us, max us; VGPR, AGPR, LDS =="]
    for name, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        m = meta.get(name, ("", "", ""))
        out.append("%-120s calls=%7d total_ms=%10.3f mean_us=%10.2f min_us=%9.2f max_us=%10.2f vgpr=%s agpr=%s lds=%s" %
                   (name[:120], len(v), sum(v) / 1e3, sum(v) / len(v), min(v), max(v), m[0], m[1], m[2]))
    return "\n".join(out)


def pmc(dirs):
    acc = defaultdict(lambda: defaultdict(list))
    for d in dirs:
        for f in _csvs(d, "counter_collection.csv"):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


KERNEL_SOURCES = ("vecchia_kernels.hip", "dev_common.h", "hist_kernels.hip", "dense_kernels.hip")


def kernel_source_hashes(root=None):
    """sha256 of the kernel sources whose counters the PMC json holds (relative to gpboost_amd/csrc)."""
    import hashlib
    root = root or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpboost_amd", "csrc")
    out = {}
    for f in KERNEL_SOURCES:
        try:
            with open(os.path.join(root, f), "rb") as fh:
                out[f] = hashlib.sha256(fh.read()).hexdigest()
        except OSError:
            out[f] = None
    return out


def main():
    if len(sys.argv) < 3:
        raise SystemExit(__doc__)
    if sys.argv[1] == "trace":
        print(trace(sys.argv[2]))
    elif sys.argv[1] == "pmc":
        acc = pmc(sys.argv[2:])
        for k in sorted(acc):
            for c in sorted(acc[k]):
                v = acc[k][c]
                print("%-100s %-28s mean=%.6g n=%d" % (k[:100], c, sum(v) / len(v), len(v)))
    elif sys.argv[1] == "pmc-json":
        acc = pmc(sys.argv[3:])
        out = {k: dict({c: sum(v) / len(v) for c, v in cs.items()}, n=max(len(v) for v in cs.values())) for k, cs in acc.items()}
        # which kernel sources the counters belong to: bench.py reports `traffic` only while these hashes match the tree it runs in
        out["_meta"] = {"kernel_sources_sha256": kernel_source_hashes()}
        with open(sys.argv[2], "w") as fh:
            json.dump(out, fh, indent=1, sort_keys=True)
        print("wrote", sys.argv[2], len(out), "kernels")


if __name__ == "__main__":
    main()
