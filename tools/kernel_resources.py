"""hipcc -Rpass-analysis=kernel-resource-usage of one csrc/*.hip file as a table (VGPRs, spills, scratch).
Usage: python tools/kernel_resources.py csrc/conv_h2.hip [name filter]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-value", "-I",
                        os.path.join(ROOT, "include"), "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"],
                       capture_output=True, text=True)
    cur = None
    rows = {}
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            cur = re.sub(r"\(anonymous namespace\)::", "", cur)
            cur = re.sub(r"^void ", "", cur)
            cur = re.sub(r"\(.*$", "", cur)
            rows[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur:
            rows[cur][m.group(1).strip()] = int(m.group(2))
    if r.returncode:
        print(r.stderr[-3000:])
    for k, v in rows.items():
        if flt in k:
            print("%-48s VGPR %3d  AGPR %3d  SGPR %3d  spill v/s %d/%d  scratch %d  occ %d" % (
                k, v.get("VGPRs", -1), v.get("AGPRs", -1), v.get("SGPRs", -1), v.get("VGPRs Spill", -1), v.get("SGPRs Spill", -1),
                v.get("ScratchSize", -1), v.get("Occupancy", -1)))


if __name__ == "__main__":
    main()
