# usage (on the GPU box through gpurun): BENCH_ARGS="--dtype bf16" bash tools/micro/pmc_all.sh <tag> <counters...>
# one rocprofv3 --pmc pass over a short bench.py run (one stream); per kernel: launches and the per-launch average of every counter
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT
tag=$1; shift 1
export MI_ONE_STREAM=1
timeout -k 10 250 rocprofv3 --pmc "$@" --kernel-trace -d $R/gpurun_out/pmcall_$tag -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-iso-pass --no-cpu-baseline --no-host-leg --no-fixed-leg --no-modes-leg ${BENCH_ARGS} > $R/gpurun_out/pmcall_$tag.log 2>&1
python3 - <<PY
import csv,collections,glob,re
f=glob.glob('$R/gpurun_out/pmcall_$tag/**/p_counter_collection.csv', recursive=True)[0]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n=re.sub(r"\(.*","",r['Kernel_Name'])[:70]
    agg[n][r['Counter_Name']].append(float(r['Counter_Value']))
names=sorted({c for v in agg.values() for c in v})
print("kernel;launches;"+";".join(names))
for n,v in sorted(agg.items(), key=lambda kv:-sum(kv[1].get(names[0],[0]))):
    print(n+";"+str(len(next(iter(v.values()))))+";"+";".join(f"{sum(v[c])/max(1,len(v[c])):.4g}" for c in names))
PY
rm -f $R/gpurun_out/pmcall_$tag/*.db $R/gpurun_out/pmcall_$tag/*/*.db $R/gpurun_out/pmcall_$tag/*trace.csv 2>/dev/null
