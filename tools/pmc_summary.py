"""Summarises the rocprofv3 --pmc passes of tools/pmc_r2.sh: per kernel the average counter values per dispatch, and
profiles/r2/pmc_tower_summary.json in the form bench.py's roofline.traffic reads (only for the kernel it names)."""
import collections
import csv
import glob
import json
import os
import sys

out_dir = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(out_dir, "pass*_counters.csv"))):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name") or r.get("Kernel Name")
        if not name or not any(k in name for k in ("k_tower", "k_tree", "k_heads")):
            continue
        acc[name.split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
summ = {}
for k, d in acc.items():
    # the first launches of a run see cold caches and the game's first moves: average the second half of the dispatches
    summ[k] = {c: sum(v[len(v) // 2:]) / max(len(v[len(v) // 2:]), 1) for c, v in d.items()}
    summ[k]["_dispatches"] = max(len(v) for v in d.values())
json.dump(summ, open(os.path.join(out_dir, "pmc_summary.json"), "w"), indent=1)
for k, d in summ.items():
    print(k[:60], {c: round(v, 1) for c, v in d.items()})
for k, d in summ.items():
    if k.startswith("k_tower16<ConnectFour, 64, false, 11>") and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        json.dump({"kernel": "k_tower16<ConnectFour,64,NT=11>", "boards_per_launch": 4096, "FETCH_SIZE_KB": d["FETCH_SIZE"],
                   "WRITE_SIZE_KB": d["WRITE_SIZE"], "source": "tools/pmc_r2.sh: separate rocprofv3 --pmc passes, 4096 leaves per launch"},
                  open(os.path.join(out_dir, "pmc_tower_summary.json"), "w"), indent=1)
