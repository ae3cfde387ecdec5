"""Per-step view of a rocprofv3 kernel-stats CSV:  python scripts/kstats.py gpurun_out/TAG/r_kernel_stats.csv [steps=28] [top=45]"""
import csv
import re
import sys

path = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 28
top = int(sys.argv[3]) if len(sys.argv) > 3 else 45
rows = list(csv.DictReader(open(path)))
print("total kernel ms/step", round(sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e6, 2))
for r in rows[:top]:
    n = r["Name"]
    m = re.search(r"rowstream_kernelINS_\d+(\w+?)I", n)
    short = m.group(1) + " " + n[n.find("Li8"):][:24] if m else n.replace("(anonymous namespace)::", "")[:64]
    print(f"{short:66s} {int(r['Calls']) / steps:6.1f}/step {float(r['AverageNs']) / 1e3:8.1f} us  {float(r['TotalDurationNs']) / steps / 1e6:6.3f} ms/step")
