# usage (build container, no GPU needed): bash tools/micro/spill_audit.sh [file.hip ...]
# Compiles the kernel files with -Rpass-analysis=kernel-resource-usage (the Makefile's flags: AGPR accumulators for gemm_half / gemm_tap /
# gemm_x6, VGPR form elsewhere) and lists every kernel whose frame has scratch bytes: registers, scratch bytes per lane, waves per SIMD.
# A spill inside an epilogue is HBM traffic: 176 bytes per lane = 45 KB per 256-thread workgroup, written and read back.
cd "$(dirname "$0")/../../demucs_amd/csrc"
files=${@:-$(ls *.hip)}
for f in $files; do
  form=1; case $f in gemm_half.hip|gemm_tap.hip|gemm_x6.hip|attention_half.hip) form=0;; esac
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=$form \
      -Rpass-analysis=kernel-resource-usage -c $f -o /dev/null 2>&1 |
    grep -E "Function Name|VGPRs:|AGPRs:|ScratchSize|Occupancy" | sed 's/.*remark: //; s/ \[-Rpass.*//' | paste - - - - - |
    awk -v f=$f '{ if ($NF+0 >= 0 && $0 ~ /ScratchSize \[bytes\/lane\]: [1-9]/) print f, $0 }' |
    sed 's/Function Name: //' | while read -r line; do
      name=$(echo "$line" | awk '{print $2}' | c++filt | cut -c1-90)
      echo "$line" | awk -v n="$name" '{for (i = 1; i <= NF; i++) { if ($i == "VGPRs:") v = $(i+1); if ($i == "AGPRs:") a = $(i+1); if ($i == "[bytes/lane]:") s = $(i+1); if ($i == "[waves/SIMD]:") o = $(i+1) }
        printf "%-16s %-92s VGPR %s AGPR %s scratch %s B/lane, %s waves/SIMD\n", $1, n, v, a, s, o}'
    done
done
