"""Summarises the rocprofv3 --pmc passes of tools/profile_r4.sh / profile_r3.sh (fp32 headline configuration and bf16 10x128): per kernel the
average counter values per dispatch -> pmc_summary.json, with a `kernels` list in the form bench.py's pmc_lookup reads
(HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KB: FETCH is counted in 64-byte... units that the guide says to double
on gfx950)."""
import collections
import csv
import glob
import json
import os
import sys

out_dir = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(list)))
for f in sorted(glob.glob(os.path.join(out_dir, "*_pass*_counters.csv"))):
    cfg = os.path.basename(f).split("_pass")[0]
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name") or r.get("Kernel Name")
        if not name or not any(k in name for k in ("k_tower", "k_tree", "k_heads")):
            continue
        acc[cfg][name.split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
summ = {}
for cfg, ks in acc.items():
    summ[cfg] = {}
    for k, d in ks.items():
        # the first launches of a run see cold caches and the game's first moves: average the second half of the dispatches
        summ[cfg][k] = {c: sum(v[len(v) // 2:]) / max(len(v[len(v) // 2:]), 1) for c, v in d.items()}
        summ[cfg][k]["_dispatches"] = max(len(v) for v in d.values())
kernels = []
for cfg, ks in summ.items():
    for k, d in ks.items():
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            short = k.replace(", false", "").replace(" ", "")
            if short.startswith("k_tower16<") or short.startswith("k_tower16b<"):      # names as az_net_last_kernel prints them
                parts = short[short.index("<") + 1:-1].split(",")
                short = "%s<%s,%s,NT=%s>" % (short[:short.index("<")], parts[0], parts[1], parts[2] if len(parts) > 2 else "11")
            kernels.append({"match": short.split("<")[0] if short.startswith("k_tree") else short, "config": cfg, "FETCH_SIZE_KB": d["FETCH_SIZE"],
                            "WRITE_SIZE_KB": d["WRITE_SIZE"], "units_per_launch": 4096,
                            "source": "tools/profile_r4.sh (r3: profile_r3.sh): separate rocprofv3 --pmc passes, 4096 slots / leaves per launch"})
summ["kernels"] = kernels
json.dump(summ, open(os.path.join(out_dir, "pmc_summary.json"), "w"), indent=1)
for cfg, ks in summ.items():
    if cfg == "kernels":
        continue
    for k, d in ks.items():
        print(cfg, k[:60], {c: round(v, 1) for c, v in d.items()})
