# usage (on the GPU box through gpurun): bash tools/micro/pmc_istft.sh <tag> <counters...>
# one rocprofv3 --pmc pass over tools/micro/istft_bench.py and the per-launch averages of the fused inverse-STFT kernel
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT
tag=$1; shift 1
timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace -d $R/gpurun_out/pmci_$tag -o p --output-format csv -- python3 $R/tools/micro/istft_bench.py > $R/gpurun_out/pmci_$tag.log 2>&1
python3 - <<PY
import csv,collections,glob
f=glob.glob('$R/gpurun_out/pmci_$tag/**/p_counter_collection.csv', recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'istft_fused_kernel' in r['Kernel_Name']]
agg=collections.defaultdict(list)
for r in rows: agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()): print('$tag',k, sum(v)/len(v), len(v))
PY
