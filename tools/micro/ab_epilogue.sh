set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4i; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log; tail -3 $O/tests.log
grep -q "tests rc 0" $O/tests.log || exit 1
BARE="--no-cpu-baseline --no-host-leg --no-fixed-leg --no-modes-leg"
for dt in f32 bf16 f32 bf16; do
  timeout -k 10 120 python bench.py --dtype $dt --steps 10 --warmup 2 $BARE 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$dt', d['ms_per_step'])" || exit 1
done | tee $O/ab.txt
timeout -k 10 200 python tools/micro/hdemucs_time.py 2>&1 | tail -3 | tee $O/hdemucs.txt
cd /tmp && export TMPDIR=/tmp
export MI_ONE_STREAM=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_f32_one -o p --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-iso-pass $BARE > $O/prof_f32_one.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_bf16_one -o p --output-format csv -- python3 $R/bench.py --dtype bf16 --steps 5 --warmup 1 --no-iso-pass $BARE > $O/prof_bf16_one.log 2>&1 || exit 1
cp $(find $O/prof_f32_one -name 'p_kernel_stats.csv' | head -1) $O/f32_one_stats.csv
cp $(find $O/prof_bf16_one -name 'p_kernel_stats.csv' | head -1) $O/bf16_one_stats.csv
rm -rf $O/prof_f32_one $O/prof_bf16_one
