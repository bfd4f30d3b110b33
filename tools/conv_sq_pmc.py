"""Per-launch SQ counters of the nine frozen-encoder convs from rocprofv3 PMC databases made over tools/conv_pmc.py:
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d DIR -o m -- python3 tools/conv_pmc.py
    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d DIR -o l -- python3 tools/conv_pmc.py
Usage: python tools/conv_sq_pmc.py DB [DB ...]  -> a table of the last repetition's launches (one row per launch, every counter found)."""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"\(.*$", "", re.sub(r"^void ", "", n))


def main():
    table, order = {}, []
    counters = []
    for db in sys.argv[1:]:
        cur = sqlite3.connect(db).cursor()
        rs = cur.execute("select kernel_name, counter_name, value, start, end from counters_collection order by start").fetchall()
        names = sorted(set(r[1] for r in rs))
        counters += [n for n in names if n not in counters]
        per = {}
        for n in names:
            sel = [r for r in rs if r[1] == n and ("h2p_kernel" in r[0] or "h2_kernel" in r[0])]
            per[n] = sel[-9:]  # the last repetition
        for n, sel in per.items():
            for i, r in enumerate(sel):
                key = (i, short(r[0]))
                if key not in table:
                    table[key] = {}
                    order.append(key)
                table[key][n] = r[2]
                table[key]["us"] = (r[4] - r[3]) / 1e3
    print("| launch | kernel | us (under the counters) | " + " | ".join(counters) + " |")
    print("|---|---|---|" + "---|" * len(counters))
    for key in sorted(set(order)):
        row = table[key]
        print("| %d | %s | %.1f | " % (key[0], key[1], row.get("us", 0.0)) + " | ".join("%.4g" % row.get(c, float("nan")) for c in counters) + " |")


if __name__ == "__main__":
    main()
