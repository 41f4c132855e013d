# same-box A/B of two builds of the library: demucs_amd/libdemucs_amd.so.base (DEMUCS_AMD_LIB) against the default one; through gpurun
J='import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])["ms_per_step"])'
for dt in bf16 f32; do
  B="python bench.py --dtype $dt --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg --no-fixed-leg --no-modes-leg --no-iso-pass"
  for i in 1 2 3; do
    echo -n "$dt base "; DEMUCS_AMD_LIB=$PWD/demucs_amd/libdemucs_amd.so.base $B 2>/dev/null | python -c "$J" || exit 1
    echo -n "$dt new  "; $B 2>/dev/null | python -c "$J" || exit 1
  done
done
