#!/bin/bash
# rocprofv3 kernel stats of the hdemucs_mmi 3-minute-track pass: bash tools/micro/profile_hdemucs.sh <tag> [dtype] [max_batch]
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/hprof -o p --output-format csv -- python3 $R/tools/micro/hdemucs_profile.py ${2:-f16} ${3:-5} > $O/hprof.log 2>&1 || { tail -5 $O/hprof.log; exit 1; }
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$O/hprof/p_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel ms per pass", round(tot / 2e6, 2))
for r in rows[:24]:
    print(r["Name"][:100], r["Calls"], round(float(r["TotalDurationNs"]) / 2e6, 2), "ms/pass", round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
