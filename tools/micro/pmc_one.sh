# usage: pmc_one.sh <tag> <bench_kernels filter> <counters...>   (run on the GPU box through gpurun)
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT
tag=$1; filt=$2; shift 2
BENCH_B=31 timeout -k 10 150 rocprofv3 --pmc "$@" --kernel-trace -d $R/gpurun_out/pmcx_$tag -o p --output-format csv -- python3 $R/tools/bench_kernels.py $filt > $R/gpurun_out/pmcx_$tag.log 2>&1
python3 - <<PY
import csv,collections
rows=[r for r in csv.DictReader(open('$R/gpurun_out/pmcx_$tag/p_counter_collection.csv')) if 'conv_gemm' in r['Kernel_Name']]
agg=collections.defaultdict(list)
for r in rows: agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items(): print('$tag',k, sum(v)/len(v), len(v))
PY
