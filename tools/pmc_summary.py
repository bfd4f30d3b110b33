"""HBM-side traffic per launch of the f16x2 conv kernels from the two rocprofv3 PMC databases made with tools/conv_pmc.py
(FETCH_SIZE and WRITE_SIZE in separate passes, as MI355X_MICROARCH.md prescribes). FETCH_SIZE is in KB and counts a wide
coalesced streaming read at half its bytes on gfx950 -> doubled; the streaming calibration kernel in the same run
(fl_affine_act, 134 MB read + 134 MB written) shows the correction. Writes profiles/<TAG>_conv_h2_pmc.{md,json}.
Usage: python tools/pmc_summary.py FETCH_DB WRITE_DB [TAG, default r03]"""
import json
import os
import re
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rows(db, counter):
    cur = sqlite3.connect(db).cursor()
    return cur.execute("select kernel_name, grid_size, value, start from counters_collection where counter_name=? order by start",
                       (counter,)).fetchall()


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"\(.*$", "", re.sub(r"^void ", "", n))


def last_rep(rs):
    """the launches of the last repetition: everything after the last calibration kernel, plus that kernel"""
    idx = [i for i, r in enumerate(rs) if "affine_act" in r[0]]
    return rs[idx[-1]:]


def main():
    fdb, wdb = sys.argv[1], sys.argv[2]
    tag = sys.argv[3] if len(sys.argv) > 3 else "r03"
    f, w = last_rep(rows(fdb, "FETCH_SIZE")), last_rep(rows(wdb, "WRITE_SIZE"))
    names = ["calibration: affine_act 256x32x32x128"] + ["down1 4x4s2", "res1.conv1 3x3", "res1.conv2 3x3", "down2 4x4s2", "res2.conv1 3x3",
                                                        "res2.conv2 3x3", "down3 4x4s2", "res3.conv1 3x3", "res3.conv2 3x3"]
    conv = lambda n: "affine_act" in n or "h2_kernel" in n or "h2p_kernel" in n
    fk = [r for r in f if conv(r[0])]
    wk = [r for r in w if conv(r[0])]
    assert len(fk) == len(wk) == 10, (len(fk), len(wk))
    B = 256
    alg = [2 * 256 * 32 * 32 * 128 * 4.0]
    for (C, H) in ((128, 32), (256, 16), (512, 8)):
        alg.append(4.0 * B * ((2 * H) ** 2 * (C // 2) + H * H * C) + 4.0 * C * 16 * (C // 2))
        alg += [4.0 * B * (2 * H * H * C) + 4.0 * C * 9 * C] * 2
    lines = ["# HBM-side traffic of the f16x2 conv kernels (%s)" % tag, "",
             "rocprofv3 `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, `--kernel-trace` only) over `tools/conv_pmc.py`: one launch of",
             "each frozen-encoder conv at B = 256 exactly as the step launches it, third repetition. FETCH_SIZE (KB) is doubled (gfx950 counts a",
             "128-B request of a wide coalesced read as 64 B, MI355X_MICROARCH.md section HBM); the calibration row - a streaming kernel that reads",
             "134.2 MB and writes 134.2 MB - shows the corrected figures landing on the known byte counts.", "",
             "| launch | kernel | FETCH x2 (MB) | WRITE (MB) | traffic (MB) | algorithmic (MB) | traffic / algorithmic |", "|---|---|---|---|---|---|---|"]
    out = {"h2_3x3": [], "h2_4x4": []}
    for n, fr, wr, a in zip(names, fk, wk, alg):
        fb, wb = 2.0 * fr[2] * 1024.0, wr[2] * 1024.0
        tot = fb + wb
        lines.append("| %s | %s | %.1f | %.1f | %.1f | %.1f | %.2f |" % (n, short(fr[0])[:34], fb / 1e6, wb / 1e6, tot / 1e6, a / 1e6, tot / a))
        if "3x3" in n:
            out["h2_3x3"].append(tot)
        elif "4x4" in n:
            out["h2_4x4"].append(tot)
    res = {k: sum(v) / len(v) for k, v in out.items()}
    lines += ["", "Average per launch: the 3x3 convs %.1f MB (6 launches), the 4x4 / stride-2 convs %.1f MB (3 launches) - `roofline.traffic` in the"
              % (res["h2_3x3"] / 1e6, res["h2_4x4"] / 1e6), "bench line (read from profiles/%s_conv_h2_pmc.json)." % tag]
    open(os.path.join(ROOT, "profiles", "%s_conv_h2_pmc.md" % tag), "w").write("\n".join(lines) + "\n")
    json.dump(res, open(os.path.join(ROOT, "profiles", "%s_conv_h2_pmc.json" % tag), "w"))
    print("\n".join(lines))


if __name__ == "__main__":
    main()
