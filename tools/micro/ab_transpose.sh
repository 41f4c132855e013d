# usage (through gpurun from the repo root): bash tools/micro/ab_transpose.sh
# parity of the kernels around the transforms, then their durations: tools/micro/transpose_bench.py (strip / walking kernels against
# the round-3 ones) under rocprofv3 --kernel-trace --stats
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4h; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "stft or transposes" > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log; tail -3 $O/tests.log
grep -q "tests rc 0" $O/tests.log || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/tb -o p --output-format csv -- python3 $R/tools/micro/transpose_bench.py > $O/tb.log 2>&1 || exit 1
f=$(find $O/tb -name 'p_kernel_stats.csv' | head -1); cp $f $O/transpose_bench_kernel_stats.csv
rm -rf $O/tb
