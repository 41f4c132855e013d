"""Per-launch durations of the LAST forward in a rocprofv3 --kernel-trace CSV (one line per launch, launch order):
python tools/micro/trace_forward.py <dir with *_kernel_trace.csv> [first kernel of a forward = segments_gather_kernel]"""
import csv
import glob
import re
import sys

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
first = sys.argv[2] if len(sys.argv) > 2 else "segments_gather_kernel"
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
lo = starts[-1]
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:]:
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void mi::", "").replace("mi::", "")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:9.1f} us  grid {int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])):7d}  {name[:90]}")
