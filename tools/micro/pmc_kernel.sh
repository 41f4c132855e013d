# usage (on the GPU box through gpurun): bash tools/micro/pmc_kernel.sh <tag> <kernel-name substring> <counters...>
# one rocprofv3 --pmc pass over a short float32 bench.py run (one stream) and the per-launch averages of the named kernel
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT
tag=$1; name=$2; shift 2
export MI_ONE_STREAM=1
timeout -k 10 250 rocprofv3 --pmc "$@" --kernel-trace -d $R/gpurun_out/pmck_$tag -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-iso-pass --no-cpu-baseline --no-host-leg --no-fixed-leg --no-modes-leg ${BENCH_ARGS} > $R/gpurun_out/pmck_$tag.log 2>&1
python3 - "$name" <<PY
import csv,collections,glob,sys
f=glob.glob('$R/gpurun_out/pmck_$tag/**/p_counter_collection.csv', recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if sys.argv[1] in r['Kernel_Name']]
agg=collections.defaultdict(list)
for r in rows: agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()): print('$tag',k, sum(v)/len(v), len(v))
PY
rm -f $R/gpurun_out/pmck_$tag/*.db $R/gpurun_out/pmck_$tag/*/*.db 2>/dev/null
