# usage (through gpurun from the repo root): bash tools/micro/ab_transpose.sh
# parity of the kernels around the transforms and of the scheduler's overlap-add, then their durations: tools/micro/transpose_bench.py
# (strip / walking kernels against the round-3 ones) and one bare fp32 bench run under rocprofv3 --kernel-trace --stats
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4g; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_apply.py -m gpu -x -q -k "stft or transposes or apply" > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log; tail -3 $O/tests.log
grep -q "tests rc 0" $O/tests.log || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/tb -o p --output-format csv -- python3 $R/tools/micro/transpose_bench.py > $O/tb.log 2>&1 || exit 1
f=$(find $O/tb -name 'p_kernel_stats.csv' | head -1); grep -E "transpose|stft" $f | cut -c1-60,120-260 | tee $O/tb_stats.txt
cp $f $O/transpose_bench_kernel_stats.csv
BARE="--no-cpu-baseline --no-host-leg --no-fixed-leg --no-modes-leg"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/pb -o p --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-iso-pass $BARE > $O/pb.log 2>&1 || exit 1
f=$(find $O/pb -name 'p_kernel_stats.csv' | head -1); grep -E "transpose|stft|ola_|segments_gather" $f | cut -c1-60,120-260 | tee $O/pb_stats.txt
rm -rf $O/tb $O/pb
