# alternating bench runs: side stream at normal / least / greatest priority (MI_SIDE_PRIO); through gpurun
J='import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])["ms_per_step"])'
for dt in f32 bf16; do
  B="python bench.py --dtype $dt --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg --no-fixed-leg --no-modes-leg --no-iso-pass"
  for i in 1 2 3; do
    echo -n "$dt normal   "; $B 2>/dev/null | python -c "$J" || exit 1
    echo -n "$dt low      "; MI_SIDE_PRIO=low $B 2>/dev/null | python -c "$J" || exit 1
    echo -n "$dt high     "; MI_SIDE_PRIO=high $B 2>/dev/null | python -c "$J" || exit 1
  done
done
for i in 1 2; do
  echo -n "hdemucs normal "; python tools/micro/hdemucs_time.py 2>/dev/null | tail -1 || exit 1
  echo -n "hdemucs low    "; MI_SIDE_PRIO=low python tools/micro/hdemucs_time.py 2>/dev/null | tail -1 || exit 1
  echo -n "hdemucs high   "; MI_SIDE_PRIO=high python tools/micro/hdemucs_time.py 2>/dev/null | tail -1 || exit 1
done
