R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; export MI_X6=1 MI_X6_MODE=1
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_x6plain.log 2>&1 && tail -1 $O/bench_x6plain.log > $O/bench_x6plain.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof_x6plain -o p --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/prof_x6plain.log 2>&1
ls $O/prof_x6plain
