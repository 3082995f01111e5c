#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average duration.
usage: rocpd_stats.py results.db [out.md]"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "").replace("lg::", "").replace("(anonymous namespace)::", "")
    return name[:110]


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(lds_size), "
                      "max(grid_x*grid_y*grid_z), max(workgroup_x) from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % | vgpr | lds B | grid threads | wg |", "|---|---|---|---|---|---|---|---|---|---|---|"]
    for n, c, tot, avg, mn, mx, vg, lds, grid, wg in rows:
        lines.append(f"| `{short(n)}` | {c} | {tot / 1e6:.3f} | {avg / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | {100 * tot / total:.1f} | {vg} | {lds} | {grid} | {wg} |")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
