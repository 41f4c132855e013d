# Round-end GPU evidence (run through gpurun from the repo root):  bash tools/run_round_profile.sh <tag>
# full GPU suite, bench line, rocprofv3 kernel stats of the same bench command, and the two PMC passes for HBM traffic.
set -o pipefail
R=$GRAFT_REPO_ROOT; tag=${1:-v5}; O=$R/gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/tests_$tag.log 2>&1; tail -3 $O/tests_$tag.log
timeout -k 10 300 python bench.py --steps 10 --warmup 2 > $O/bench_$tag.log 2>&1 && tail -1 $O/bench_$tag.log > $O/bench_$tag.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof_$tag -o p --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/prof_$tag.log 2>&1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch_$tag -o p --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_fetch_$tag.log 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write_$tag -o p --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_write_$tag.log 2>&1
cd $R && python3 tools/pmc_traffic.py $O/pmc_fetch_$tag $O/pmc_write_$tag $O/traffic_$tag.json | head -8
ls $O/prof_$tag | head
