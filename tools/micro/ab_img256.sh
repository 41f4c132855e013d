# alternating bf16 bench runs: default tiles vs MI_IMG256=1 (256 x 256 tile for the residual-epilogue linears); through gpurun
B="python bench.py --dtype bf16 --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg --no-fixed-leg --no-modes-leg --no-iso-pass"
for i in 1 2 3; do
  $B 2>/dev/null | python -c "import sys,json; print('default   ', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])" &&
  MI_IMG256=1 $B 2>/dev/null | python -c "import sys,json; print('MI_IMG256=1', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])" || exit 1
done
