#!/bin/bash
# bash tools/micro/lstm_bench.sh <tag>: kernel durations of the persistent LSTM kernel (rocprofv3) + its polling share
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
MI_LSTM_DEBUG=1 python3 $R/tools/micro/lstm_bench.py 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/lprof -o p --output-format csv -- python3 $R/tools/micro/lstm_bench.py > $O/lprof.log 2>&1 || { tail -5 $O/lprof.log; exit 1; }
python3 - <<PY
import csv
for r in csv.DictReader(open("$O/lprof/p_kernel_stats.csv")):
    if "lstm" in r["Name"]: print(r["Name"][:60], r["Calls"], "avg us", round(float(r["AverageNs"]) / 1e3, 1), "min", round(float(r["MinNs"]) / 1e3, 1), "-> per step", round(float(r["AverageNs"]) / 200e3, 2), "us")
PY
