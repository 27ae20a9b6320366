"""Print registers / LDS / occupancy per kernel of one HIP source as hipcc reports them (-Rpass-analysis=kernel-resource-usage).

    python tools/kernel_resources.py mv3d_tf_amd/csrc/roi_pool.hip [name filter] [-- extra hipcc flags]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        k = args.index("--")
        args, extra = args[:k], args[k + 1:]
    src = args[0]
    flt = args[1] if len(args) > 1 else ""
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
           "-I" + os.path.join(ROOT, "include"), "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + extra
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = None
    rows = {}
    for line in err.splitlines():
        m = re.search(r"remark: .*?(Function Name|Name): (\S+)", line)
        if m:
            cur = subprocess.run(["c++filt", m.group(2)], capture_output=True, text=True).stdout.strip()
            cur = re.sub(r"\(.*", "", cur)
            rows[cur] = {}
            continue
        m = re.search(r"remark: .*?\s+(\w[\w \[\]/]*): (\d+)", line)
        if m and cur:
            rows[cur][m.group(1).strip()] = int(m.group(2))
    if not rows:
        print(err[-3000:])
    for name, r in rows.items():
        if flt in name:
            print("%-70s VGPR %3d AGPR %3d SGPR %3d LDS %6d scratch %4d occ %d" % (name[:70], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("TotalSGPRs", -1),
                  r.get("LDS Size [bytes/block]", -1), r.get("ScratchSize [bytes/lane]", -1), r.get("Occupancy [waves/SIMD]", -1)))


if __name__ == "__main__":
    main()
