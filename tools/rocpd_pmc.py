#!/usr/bin/env python3
"""Per-kernel averages of the PMC counters in a rocprofv3 rocpd database.
usage: rocpd_pmc.py results.db [out.md]"""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*$", "", name).replace("void ", "").replace("lg::", "")
    return name[:70]


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select kernel_name, counter_name, avg(value), count(*), avg(duration) from counters_collection group by kernel_name, counter_name").fetchall()
    tab = defaultdict(dict)
    dur = {}
    for k, c, v, n, d in rows:
        tab[k][c] = v
        dur[k] = (n, d)
    counters = sorted({c for k in tab for c in tab[k]})
    lines = ["| kernel | n | avg us | " + " | ".join(counters) + " |", "|---|---|---|" + "---|" * len(counters)]
    for k in sorted(tab, key=lambda k: -dur[k][0] * (dur[k][1] or 0)):
        lines.append(f"| `{short(k)}` | {dur[k][0]} | {(dur[k][1] or 0) / 1e3:.1f} | " + " | ".join(f"{tab[k].get(c, 0):.4g}" for c in counters) + " |")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
