#!/usr/bin/env python
"""The LAST n dispatches of a kernel in a rocprofv3 --kernel-trace CSV: their average duration.

`rocprofv3 --stats` averages a kernel over the whole process; bench.py's warm-up (a long phase's steady state: ~14 000 waves whose
launches shrink from 4096 boards to the steady ~1800 as the games spread out and the evaluation cache fills) is in that average, the
timed region is its last `--steps x slot groups` launches.  With `--headline-only` nothing is launched after the timed region, so the
window below IS the timed region and its mean is comparable with the line's roofline.avg_launch_ms (HIP events of the same run).

    tools/trace_window.py <kernel_trace.csv> <kernel name prefix> <n> [rows_out.csv]  ->  one JSON line
"""
import csv
import json
import sys


def main():
    path, prefix, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
    rows_out = sys.argv[4] if len(sys.argv) > 4 else None
    with open(path, newline="") as f:
        rd = csv.reader(f)
        head = next(rd)
        col = {h.strip().lower(): i for i, h in enumerate(head)}
        kn = next(i for h, i in col.items() if h == "kernel_name")
        st = next(i for h, i in col.items() if h == "start_timestamp")
        en = next(i for h, i in col.items() if h == "end_timestamp")
        sel, total = [], 0
        for r in rd:
            name = r[kn].replace("void ", "")
            if name.startswith(prefix):
                total += 1
                sel.append((int(r[st]), int(r[en]), r))
    sel.sort(key=lambda x: x[0])
    win = sel[-n:]
    d = [e - s for s, e, _ in win]
    out = {"trace": path.split("/")[-1], "kernel_prefix": prefix, "dispatches_in_process": total, "window": "last %d dispatches" % len(win),
           "avg_us": sum(d) / len(d) / 1e3, "min_us": min(d) / 1e3, "max_us": max(d) / 1e3,
           "window_span_ms": (win[-1][1] - win[0][0]) / 1e6, "avg_us_whole_process": sum(e - s for s, e, _ in sel) / len(sel) / 1e3}
    print(json.dumps(out))
    if rows_out:
        with open(rows_out, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(head)
            for _, _, r in win:
                w.writerow(r)


if __name__ == "__main__":
    main()
