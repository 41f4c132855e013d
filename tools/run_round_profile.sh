# Round-end GPU evidence (run through gpurun from the repo root):  bash tools/run_round_profile.sh <tag> [tests|profiles|all]
# tests: full GPU suite + the default bench line; profiles: rocprofv3 kernel stats of the bench command (fp32, bf16), of the
# hdemucs_mmi fp16 pass, and the two PMC passes for HBM traffic.
set -o pipefail
R=$GRAFT_REPO_ROOT; tag=${1:-round4}; stage=${2:-all}; O=$R/gpurun_out/$tag; mkdir -p $O
if [ $stage != profiles ]; then
timeout -k 10 1000 python -m pytest tests -m gpu -q -s > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log; tail -3 $O/tests.log
timeout -k 10 400 python bench.py --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
fi
[ $stage = tests ] && exit 0
cd /tmp && export TMPDIR=/tmp
BARE="--no-cpu-baseline --no-host-leg --no-fixed-leg --no-modes-leg"
# default schedule (waveform branch on a side stream): the timed region alone
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_f32 -o p --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-iso-pass $BARE > $O/prof_f32.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_bf16 -o p --output-format csv -- python3 $R/bench.py --dtype bf16 --steps 5 --warmup 1 --no-iso-pass $BARE > $O/prof_bf16.log 2>&1
# the same commands with every launch on one stream (one kernel on the GPU at a time): the per-kernel durations bench.py's roofline quotes
export MI_ONE_STREAM=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_f32_one -o p --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-iso-pass $BARE > $O/prof_f32_one.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_bf16_one -o p --output-format csv -- python3 $R/bench.py --dtype bf16 --steps 5 --warmup 1 --no-iso-pass $BARE > $O/prof_bf16_one.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o p --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-iso-pass $BARE > $O/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -o p --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-iso-pass $BARE > $O/pmc_write.log 2>&1
unset MI_ONE_STREAM
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_hdemucs_f16 -o p --output-format csv -- python3 $R/tools/micro/hdemucs_profile.py f16 5 > $O/prof_hdemucs.log 2>&1
cd $R && python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/traffic.json | head -30
rm -rf $O/pmc_fetch/*/*.db $O/pmc_write/*/*.db 2>/dev/null; du -sh $O
