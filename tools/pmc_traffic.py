"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; one counter per pass as the TCC slots
require) of `python3 bench.py --steps K --warmup 1 --no-cpu-baseline` into per-launch HBM traffic of
the engine's kernel classes, written to profiles/<name>.json (bench.py copies the matching entry
into roofline.traffic).

    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/round1_traffic.json

gfx950 corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are in KiB;
FETCH_SIZE reports exactly half of the bytes of wide (16 B/lane) coalesced reads -- the LDS-DMA
GEMMs, the attention kernels, the float4 / float2 streaming kernels -- so their reads are doubled;
WRITE_SIZE is taken as is.  The guide states the factor for 16 B/lane streaming reads only: for the
table-driven gather loops (dword loads) it is uncalibrated, so those classes carry `read_access:
"dword gather"` and both bounds (`read_bytes_x1`, the raw counter, and the doubled figure).
"""
import collections
import csv
import glob
import json
import re
import sys

EPI = {0: "linear", 1: "glu", 2: "bias_stats", 3: "stats_only", 4: "gn_glu", 5: "convtr"}
TILE = {(2, 2, 2, 2): 128, (1, 4, 3, 1): 96, (1, 4, 2, 1): 64, (1, 4, 1, 1): 32}


def klass(name):
    m = re.search(r"conv_gemm_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (true|false)>", name)
    if m:
        wm, wn, tm, tn, epi, _, plain = m.groups()
        return f"conv_gemm<{EPI[int(epi)]},tile{TILE[(int(wm), int(wn), int(tm), int(tn))]}{',1x1' if plain == 'true' else ''}>"
    m = re.search(r"conv_gemm_dma_kernel<(\d+), (\d+)>", name)      # LDS-DMA main loop of the plain 128-row LINEAR tile
    if m:
        return f"conv_gemm<{EPI[int(m.group(1))]},tile128,1x1>"
    m = re.search(r"conv_gemm_dmatap_kernel<(\d+), (\d+), (\d+)(?:, \d+)?>", name)      # round 4: LDS-DMA main loop of the float32 k x k GLU convs
    if m:
        return f"conv_gemm<{EPI[int(m.group(2))]},tile{m.group(1)}>"
    m = re.search(r"conv_gemm_dmarow_kernel<(\d+), (\d+), (\d+)>", name)      # round 4: LDS-DMA main loop of the float32 row-tap layers
    if m:
        epi = int(m.group(2))
        return f"conv_gemm<{EPI[epi]},tile{m.group(1)}{',1x1' if epi in (1, 4) else ''}>"      # GLU / GroupNorm-GLU run it as plain 1x1 layers only
    m = re.search(r"conv_gemm_x6_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (true|false)>", name)
    if m:
        wm, wn, tm, tn, epi, _, plain = m.groups()
        return f"conv_gemm_x6<{EPI[int(epi)]},tile{TILE[(int(wm), int(wn), int(tm), int(tn))]}{',1x1' if plain == 'true' else ''}>"
    m = re.search(r"conv_gemm_half_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (true|false)>", name)
    if m:
        ht, wm, wn, tm, tn, epi, _, plain = m.groups()
        tile = 128 if (wm, wn, tm, tn) == ("2", "2", "4", "2") else TILE[(int(wm), int(wn), int(tm), int(tn))]     # 256-row tile: bench class tile128
        return f"conv_gemm_{'bf16' if ht == '1' else 'f16'}<{EPI[int(epi)]},tile{tile}{',1x1' if plain == 'true' else ''}>"
    m = re.search(r"conv_gemm_half_img(?:256)?_kernel<(\d+)", name)      # operand-image inputs (LDS-DMA): the transformer's linear layers
    if m:
        return f"conv_gemm_{'bf16' if m.group(1) == '1' else 'f16'}<linear,tile128,1x1>"
    # kernels named in north_star's HBM-bound list: STFT (K1), iSTFT (K15), the fused DConv kernels, the scheduler's OLA
    for k in ("attention_heads_kernel", "attention_half_kernel", "attention_kernel", "dconv_rowlds_kernel", "dconv_row_kernel", "dconv_t_conv3_kernel", "dconv_t_gram_kernel", "dconv_t_out_kernel",
              "istft_fused_kernel", "istft_frames_kernel", "istft_ola_kernel", "lstm_persist_kernel", "stft_frames_kernel", "stft_walk_kernel", "cac_transpose_kernel", "spec_transpose_kernel", "cac_transpose_strip_kernel", "spec_transpose_strip_kernel",
              "ola_accumulate_kernel", "ola_finish_kernel", "segments_gather_kernel", "token_tile_kernel", "row_stats_kernel", "gn_gelu_kernel"):
        if k in name:
            return k
    return None


DMA_TAP = set()        # classes seen running conv_gemm_dmatap_kernel / conv_gemm_dmarow_kernel (16-byte DMA runs)


def collect(d, counter):
    out = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = klass(r["Kernel_Name"])
            if k and ("conv_gemm_dmatap_kernel" in r["Kernel_Name"] or "conv_gemm_dmarow_kernel" in r["Kernel_Name"]):
                DMA_TAP.add(k)
            if k:
                out[k][0] += 1
                out[k][1] += float(r["Counter_Value"])
    return out


def main():
    fetch, write, dst = sys.argv[1:4]
    f, w = collect(fetch, "FETCH_SIZE"), collect(write, "WRITE_SIZE")
    res = {}
    for k in sorted(f):
        n, fs = f[k]
        wn, ws = w.get(k, [0, 0.0])
        rd = 2.0 * fs * 1024 / n
        wr = ws * 1024 / max(1, wn)
        # table-driven dword gathers: the x2 factor is uncalibrated (the float32 k x k GLU convs left that class in round 4: 16-byte DMA runs)
        gather = k.startswith("conv_gemm") and not k.endswith(",1x1>") and k not in DMA_TAP
        res[k] = {"launches_sampled": n, "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr),
                  "traffic_bytes_per_launch": round(rd + wr), "read_access": "dword gather" if gather else "wide (>= 8 B per lane)",
                  "read_bytes_x1": round(rd / 2),
                  "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, KiB units, FETCH_SIZE x2 (gfx950 wide-read correction"
                            + ("; uncalibrated for this class's dword gathers: the true read figure lies between read_bytes_x1 and read_bytes_per_launch)" if gather else ")")}
    json.dump(res, open(dst, "w"), indent=1)
    for k, v in res.items():
        print(f"{k:40s} {v['traffic_bytes_per_launch'] / 1e6:9.1f} MB/launch  (read {v['read_bytes_per_launch'] / 1e6:.1f}, write {v['write_bytes_per_launch'] / 1e6:.1f})")


if __name__ == "__main__":
    main()
