# usage (on the GPU box through gpurun): bash tools/micro/pmc_transpose.sh <tag> <counter>
# one rocprofv3 --pmc pass over tools/micro/transpose_bench.py and the per-launch averages of the four transpose kernels (strip / tile)
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT
tag=$1; shift 1
timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace -d $R/gpurun_out/pmct_$tag -o p --output-format csv -- python3 $R/tools/micro/transpose_bench.py > $R/gpurun_out/pmct_$tag.log 2>&1
python3 - <<PY
import csv,collections,glob
f=glob.glob('$R/gpurun_out/pmct_$tag/**/p_counter_collection.csv', recursive=True)[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name'].split('(')[0]
    if 'transpose' in n: agg[(n, r['Counter_Name'])].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()): print('$tag', k[0], k[1], sum(v)/len(v), len(v))
PY
