# usage (on the GPU box through gpurun): bash tools/micro/pmc_launches.sh <tag> <kernel-name substring> <counter>
# one rocprofv3 --pmc pass over a short float32 bench.py run (one stream): the counter of EVERY launch of the named kernel with its grid
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT
tag=$1; name=$2; shift 2
export MI_ONE_STREAM=1
timeout -k 10 250 rocprofv3 --pmc "$@" --kernel-trace -d $R/gpurun_out/pmcl_$tag -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-iso-pass --no-cpu-baseline --no-host-leg --no-fixed-leg --no-modes-leg ${BENCH_ARGS} > $R/gpurun_out/pmcl_$tag.log 2>&1
python3 - "$name" <<PY
import csv,glob,sys
f=glob.glob('$R/gpurun_out/pmcl_$tag/**/p_counter_collection.csv', recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if sys.argv[1] in r['Kernel_Name']]
for r in rows[-16:]: print('$tag', r['Counter_Name'], r['Grid_Size'], r['Counter_Value'])
PY
rm -rf $R/gpurun_out/pmcl_$tag
