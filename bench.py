"""Benchmark of the segmented htdemucs inference path on MI355X.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

A "step" = one `apply_model(shifts=0, split=True, overlap=0.25)` pass over a synthetic 44.1 kHz stereo
track that is already resident in HBM; the separated stems stay in HBM.  Per GPU the workload is
BASELINE.json configs[1]: htdemucs 4-stem, fp32, a 3-minute track (31 segments of 7.8 s).  For N > 1
the track is N x 3 minutes (weak scaling): segments are sharded over the ranks and the slabs are
exchanged with one RCCL all-gather (demucs_amd/distributed.py).  value = audio-seconds / wall-seconds
(whole job).  Rank 0 prints ONE JSON line with `roofline` (dominant kernel, HIP-event timed on the
launch stream inside the timed region) and, at N = 1, `cpu_baseline` (the CPU oracle, i.e. a port of
the reference `-d cpu` path, on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: dense f32-in MFMA peak
BF16_MFMA_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA peak (the 2:1-sparsity figure is not used)
X6_PRODUCTS = 6                   # gemm_x6.hip: bf16 MFMA products issued per fp32 multiply-accumulate
HBM_PEAK_GBPS = 8000.0
SR = 44100
TRACK_SECONDS_PER_GPU = 180


def cpu_baseline(sd, sources, seconds=24):
    """CPU oracle (port of the reference CPU path) timed on the host cores, float32, all threads."""
    from demucs_amd.synth import synth_mix
    from oracle import apply_oracle as A
    from oracle import htdemucs_oracle as O
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))      # the GPU box gives one GPU a 16-core share
    model = O.OracleModel(sd, sources)
    length = seconds * SR
    mix = torch.from_numpy(synth_mix(1, length, "noise"))[None]
    A.apply_model(model, mix[..., :SR], shifts=0, split=True, overlap=0.25)      # warm-up forward
    t0 = time.perf_counter()
    A.apply_model(model, mix, shifts=0, split=True, overlap=0.25)
    dt = time.perf_counter() - t0
    n_seg = len(range(0, length, int(0.75 * 343980)))
    return dict(value=round(seconds / dt, 3), unit="audio-sec/wall-sec", cores=torch.get_num_threads(), kind="port",
                sample=f"{seconds} s of the same synthetic noise track ({n_seg} segment forwards), float32, "
                       f"oracle.apply_oracle.apply_model, {dt:.1f} s wall")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=32, help="segments per batched forward (workspace ~0.57 GB each)")
    ap.add_argument("--seconds", type=int, default=TRACK_SECONDS_PER_GPU, help="track seconds per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pcie", action="store_true", help="also time host->device->host apply_model (reported separately)")
    args = ap.parse_args()

    import torch.distributed as dist
    from demucs_amd import apply as P
    from demucs_amd.distributed import apply_model_sharded
    from demucs_amd.htdemucs import HTDemucs
    from demucs_amd.weights import HTDemucsConfig, synthetic_state_dict

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs `python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py ...`")
        args.gpus = world
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    cfg = HTDemucsConfig()
    sd = synthetic_state_dict(cfg, 0)
    model = HTDemucs(cfg.sources, max_batch=args.batch)
    model.load_state_dict(sd)
    model.to(dev).eval()

    length = args.seconds * SR * world
    gen = torch.Generator(device=dev).manual_seed(1)
    mix = (torch.randn(1, 2, length, device=dev, generator=gen) * 0.1).contiguous()     # synthetic, resident in HBM
    n_segments = len(range(0, length, int(0.75 * cfg.segment_length)))

    def step():
        if world > 1:
            return apply_model_sharded(model, mix, overlap=0.25)
        return P.apply_model(model, mix, shifts=0, split=True, overlap=0.25)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        out = step()
    fence()
    model.profile_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    elapsed = time.perf_counter() - t0
    rows = model.profile_end()
    assert out.shape == (1, 4, 2, length) and bool(torch.isfinite(out[0, 0, 0, ::997]).all())
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        sec_per_step = elapsed / args.steps
        dom = max(rows, key=lambda r: r["ms"])
        dom_ms = dom["ms"] / dom["launches"]
        algo_tflops = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        # conv_gemm_x6 classes run fp32 operands as 3 exact bf16 terms x 6 bf16 MFMA products: the matrix pipe is priced
        # by the bf16 flops it actually issues (6 x algorithmic) against the dense bf16 peak
        x6 = dom["name"].startswith("conv_gemm_x6")
        achieved = algo_tflops * (X6_PRODUCTS if x6 else 1)
        peak = BF16_MFMA_PEAK_TFLOPS if x6 else FP32_MFMA_PEAK_TFLOPS
        total_ms = sum(r["ms"] for r in rows)
        traffic, traffic_src = None, None        # HBM bytes per launch from the committed PMC passes of this same command
        tpath = os.path.join(ROOT, "profiles", "round1_traffic.json")
        if os.path.exists(tpath) and world == 1 and args.batch == 32 and args.seconds == TRACK_SECONDS_PER_GPU:
            entry = json.load(open(tpath)).get(dom["name"])
            if entry:
                traffic, traffic_src = entry["traffic_bytes_per_launch"], "profiles/round1_traffic.json: " + entry["method"]
        result = {
            "metric": "real-time factor (audio-sec/wall-sec) htdemucs 4-stem 44.1kHz stereo",
            "value": round(length / SR / sec_per_step, 2), "unit": "audio-sec/wall-sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(sec_per_step * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"htdemucs 4-stem fp32, {args.seconds * world // 60}-min synthetic 44.1 kHz stereo track "
                                   f"resident in HBM ({args.seconds} s per GPU), segment=7.8 s overlap=0.25 shifts=0, "
                                   f"{n_segments} segments, {args.batch} segments per batched forward, random-init weights "
                                   "(synthetic_state_dict seed 0), stems left in HBM",
                       "parallelism": f"segments sharded over {world} GPU(s)" + (", one RCCL all-gather of slabs" if world > 1 else "")},
            "roofline": {"bound": "mfma", "kernel": dom["name"], "achieved": round(achieved, 2), "peak": peak,
                         "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "pipe": ("bf16 MFMA, 6 products per fp32 MAC (exact 3-term operand split, fp32 accumulate)" if x6
                                  else "fp32 MFMA"),
                         "fp32_equivalent_tflops": round(algo_tflops, 2),
                         "algorithmic_bytes_per_launch": round(dom["bytes"] / dom["launches"]),
                         "algorithmic_flops_per_launch": round(dom["flops"] / dom["launches"]),
                         "launches": dom["launches"], "avg_launch_ms": round(dom_ms, 4),
                         "share_of_instrumented_time": round(dom["ms"] / total_ms, 3)},
            "kernels": [{"name": r["name"], "launches": r["launches"], "ms": round(r["ms"], 3),
                         "tflops": round(r["flops"] / (r["ms"] * 1e-3) / 1e12, 2),
                         "gbps": round(r["bytes"] / (r["ms"] * 1e-3) / 1e9, 1)} for r in sorted(rows, key=lambda r: -r["ms"])],
            "whole_path_tflops": round(334.9e9 * n_segments / world / sec_per_step / 1e12 * world, 2),
        }
        if args.pcie and world == 1:
            host_mix = mix.cpu().pin_memory()
            P.apply_model(model, host_mix, shifts=0, device=dev)
            t1 = time.perf_counter()
            P.apply_model(model, host_mix, shifts=0, device=dev)
            result["pcie_inclusive_value"] = round(length / SR / (time.perf_counter() - t1), 2)
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(sd, cfg.sources)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
