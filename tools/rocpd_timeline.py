"""Kernel timeline of ONE steady-state step from a rocprofv3 rocpd database: every dispatch between two consecutive launches of an
anchor kernel (default: the stem conv, first kernel of the encoder graph) with its start offset, duration and the idle gap in front.
Usage: python tools/rocpd_timeline.py DB [anchor substring] [which occurrence, default: the last but one]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\((?:[^()]|\([^()]*\))*\)( \[clone.*\])?$", "", name)[:70]


def main():
    db = sqlite3.connect(sys.argv[1])
    anchor = sys.argv[2] if len(sys.argv) > 2 else "conv3x3_c4_o64"
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    nm = "name" if "name" in cols else "kernel_name"
    rows = cur.execute("select %s, start, end from kernels order by start" % nm).fetchall()
    idx = [i for i, r in enumerate(rows) if anchor in r[0]]
    k = int(sys.argv[3]) if len(sys.argv) > 3 else len(idx) - 2
    a, b = idx[k], idx[k + 1]
    t0, prev = rows[a][1], rows[a][1]
    busy = 0
    print("%9s %9s %8s  %s" % ("start_us", "dur_us", "gap_us", "kernel"))
    for n, s, e in rows[a:b]:
        print("%9.1f %9.1f %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, short(n)))
        prev = e
        busy += e - s
    print("step %.1f us, kernels busy %.1f us, idle %.1f us, %d dispatches" % ((rows[b][1] - t0) / 1e3, busy / 1e3,
                                                                          (rows[b][1] - t0 - busy) / 1e3, b - a))


if __name__ == "__main__":
    main()
