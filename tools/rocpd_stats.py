"""Per-kernel statistics (calls, total / average / min / max duration, % of kernel time) from a rocprofv3 rocpd
sqlite database (rocprofv3 --kernel-trace ... -d DIR -o NAME  ->  DIR/NAME_results.db), as CSV on stdout.
Usage: python tools/rocpd_stats.py DB [skip_first_N_dispatches]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\((?:[^()]|\([^()]*\))*\)( \[clone.*\])?$", "", name)


def main():
    db = sqlite3.connect(sys.argv[1])
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    nm = "name" if "name" in cols else "kernel_name"
    rows = cur.execute("select %s, start, end from kernels order by start" % nm).fetchall()[skip:]
    agg = {}
    for n, s, e in rows:
        a = agg.setdefault(short(n), [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    print('"Name","Calls","TotalDurationNs","AverageNs","MinNs","MaxNs","Percentage"')
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('"%s",%d,%d,%.1f,%d,%d,%.2f' % (n, a[0], a[1], a[1] / a[0], a[2], a[3], 100.0 * a[1] / tot))


if __name__ == "__main__":
    main()
