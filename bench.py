"""Benchmark of the segmented htdemucs inference path on MI355X.

    python bench.py [--gpus N --steps K --warmup W] [--dtype f32|bf16|f16]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

A "step" = one `apply_model(shifts=0, split=True, overlap=0.25)` pass over a synthetic 44.1 kHz stereo track.

N = 1   BASELINE.json configs[1]: htdemucs 4-stem, a 3-minute track (31 segments of 7.8 s).  `value` is measured with the
        mix already resident in HBM and the stems left in HBM (the bench contract).  The same run also reports
          * `host_to_host` -- SURVEY.md 8(d)'s span: pinned host mix in, stems materialised on the host (H2D + D2H inside
            the clock), median of the timed steps;
          * `fixed_track`  -- the 60-minute track of configs[3] on this one GPU: the N = 1 point of the fixed-length curve;
          * `cpu_baseline` -- the CPU oracle (a port of the reference `-d cpu` path on the same library primitives) on a
            bounded sample of the same workload, on the host cores.
N > 1   BASELINE.json configs[3]: ONE fixed 60-minute track (616 segments) at every N ("scaling": "strong"): every pass's
        segments are sharded over the ranks, each rank keeps its window of the mix and its slab of the stems, ONE RCCL
        all-gather of the slabs over xGMI per step (demucs_amd/distributed.py); every rank ends with the full result in
        HBM.  value = 3600 audio-seconds / max-over-ranks wall-seconds.
Rank 0 prints ONE JSON line with `roofline` for the kernel class with the largest summed duration (HIP-event timed on the
launch stream inside the timed region).
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

# RCCL and device-tensor sharing between processes need the dmabuf IPC path on this pool's host driver (the legacy path fails
# with `hipIpcGetMemHandle: invalid argument`); the image exports it already -- keep it for every child this file starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402  (importing torch does not touch the GPU)

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# MI355X_MICROARCH.md: dense MFMA peaks per input type (the 2:1-sparsity figures are never used), HBM3E
MFMA_PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0, "f16": 2500.0}
X6_PRODUCTS = 6                   # gemm_x6.hip: bf16 MFMA products issued per fp32 multiply-accumulate
HBM_PEAK_GBPS = 8000.0
SR = 44100
TRACK_SECONDS = 180               # configs[1]
FIXED_SECONDS = 3600              # configs[3]
FLOP_PER_SEGMENT = 334.9e9


def cpu_baseline(sd, sources, seconds=24, seconds_one_thread=7):
    """CPU oracle (port of the reference CPU path, same library primitives) timed on the host cores, float32: on all the
    cores this process may use (`value`, `cores`) and on ONE thread (`one_thread`, the convention of the reference's
    tools/bench.py:74) -- BASELINE.md section 3."""
    from demucs_amd.synth import synth_mix
    from oracle import apply_oracle as A
    from oracle import htdemucs_oracle as O
    avail = usable_cores()
    old_threads = torch.get_num_threads()
    O.FAST_PRIMITIVES = True      # th.stft / th.istft / fused attention, as the reference calls them
    stride = int(0.75 * 343980)

    def timed(n_threads, secs):
        torch.set_num_threads(n_threads)
        length = secs * SR
        mix = torch.from_numpy(synth_mix(1, length, "noise"))[None]
        if n_threads > 1:
            A.apply_model(model, mix[..., :SR], shifts=0, split=True, overlap=0.25)      # warm-up forward
        t0 = time.perf_counter()
        A.apply_model(model, mix, shifts=0, split=True, overlap=0.25)
        return time.perf_counter() - t0, len(range(0, length, stride))
    try:
        model = O.OracleModel(sd, sources)
        dt, n_seg = timed(max(1, avail), seconds)
        dt1, n_seg1 = timed(1, seconds_one_thread)
    finally:
        O.FAST_PRIMITIVES = False
        torch.set_num_threads(old_threads)
    return dict(value=round(seconds / dt, 3), unit="audio-sec/wall-sec", cores=max(1, avail), kind="port",
                cpu_model=cpu_model_name(), os_cpu_count=os.cpu_count(),
                one_thread={"value": round(seconds_one_thread / dt1, 3), "unit": "audio-sec/wall-sec", "cores": 1,
                            "sample": f"{seconds_one_thread} s of the same track ({n_seg1} segment forwards), {dt1:.1f} s wall"},
                vs_reference=">= the reference: in the build container (8 vCPU) this port ran 0.98-1.44x as fast as the imported "
                             "reference's own apply_model -d cpu on the same input (rounds 2-3: builder 0.98-1.28x, judge 1.17-1.44x; "
                             "outputs equal to 1.5e-6), so the GPU / CPU ratio quoted from it errs against the GPU",
                sample=f"{seconds} s of the same synthetic noise track ({n_seg} segment forwards), float32, "
                       f"oracle.apply_oracle.apply_model on th.stft / th.istft / fused attention, {dt:.1f} s wall on {max(1, avail)} threads")


_T0 = time.perf_counter()


def log(msg):
    """Progress to stderr (stdout carries only the JSON line): a leg that takes long is visible, and a silent run is not
    mistaken for a hung one."""
    print(f"[bench {time.perf_counter() - _T0:7.1f} s] {msg}", file=sys.stderr, flush=True)


def usable_cores():
    """Threads the CPU leg may use: the affinity mask, capped by the cgroup CPU quota (a GPU box gives one GPU a share of its
    host, 16 cores on this pool: more threads than the quota only spin against it) and by 16."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def self_launch(n):
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks as a CHILD `torch.distributed.run` (this
    process has made no GPU call yet and makes none), relay the child's output -- rank 0's JSON line is the last line on
    stdout -- and exit with its code.  Never an exec: a process replaced after GPU initialisation takes the box down."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    argv = [a for a in sys.argv[1:] if a != "--spawn"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    p = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ), stdout=subprocess.PIPE, text=True)
    sys.stdout.write(p.stdout)
    sys.stdout.flush()
    raise SystemExit(p.returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=32, help="segments per batched forward (workspace ~0.57 GB each)")
    ap.add_argument("--seconds", type=int, default=None, help="track seconds (default: 180 at N = 1, 3600 at N > 1)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16", "f16"], help="compute mode of the engine")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-leg", action="store_true")
    ap.add_argument("--no-fixed-leg", action="store_true")
    ap.add_argument("--no-modes-leg", action="store_true")
    ap.add_argument("--no-iso-pass", action="store_true", help="skip the one-kernel-at-a-time passes behind the timed region (profiling "
                    "the default two-stream schedule alone); the roofline then quotes the timed region's own, overlapping, durations")
    ap.add_argument("--force-dist", action="store_true", help="run the N > 1 branch (RCCL process group, sharded apply_model with its "
                    "collectives, max-over-ranks all_reduce) whatever WORLD_SIZE is: rehearses the multi-GPU code on one GPU")
    ap.add_argument("--spawn", action="store_true", help="start the ranks as a child torch.distributed.run even for --gpus 1 (what "
                    "--gpus N > 1 does by itself when no torchrun environment is present)")
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.spawn):
        self_launch(args.gpus)

    import torch.distributed as dist
    from demucs_amd import apply as P
    from demucs_amd.htdemucs import HTDemucs
    from demucs_amd.weights import HTDemucsConfig, synthetic_state_dict

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world                 # under torchrun the launcher's world size is the truth
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    multi = world > 1 or args.force_dist
    saved_stdout = None
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # RCCL prints its version banner on STDOUT when a communicator is created: until the warm-up steps are over (every
        # communicator exists by then) file descriptor 1 points at stderr, so that stdout carries nothing but the JSON line
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
    if args.force_dist:
        from demucs_amd import distributed as _dd
        _dd._forced = True

    cfg = HTDemucsConfig()
    sd = synthetic_state_dict(cfg, 0)
    model = HTDemucs(cfg.sources, max_batch=args.batch, compute_dtype=args.dtype)
    model.load_state_dict(sd)
    model.to(dev).eval()

    seconds = args.seconds or (FIXED_SECONDS if multi else TRACK_SECONDS)
    stride = int(0.75 * cfg.segment_length)

    def make_mix(secs):
        gen = torch.Generator(device=dev).manual_seed(1)                # same stream on every rank
        return (torch.randn(1, 2, secs * SR, device=dev, generator=gen) * 0.1).contiguous()

    def step(m):
        # with a process group of more than one rank apply_model shards the segments by itself
        return P.apply_model(model, m, shifts=0, split=True, overlap=0.25, device=dev)

    def fence():
        if multi:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # engine handle, weight images and workspace are built by the first forward: with --warmup 0 do that on a 10-second
    # clip, outside the timed region (single-process, no collective)
    if args.warmup == 0:
        from demucs_amd.distributed import no_sharding
        with no_sharding():
            P.apply_model(model, make_mix(10), shifts=0, split=True, overlap=0.25, device=dev)
    log(f"engine ready; {seconds} s track, warm-up {args.warmup}, timed steps {args.steps}")
    mix = make_mix(seconds)                                             # synthetic, resident in HBM
    length = mix.shape[-1]
    n_segments = len(range(0, length, stride))
    out = None
    for _ in range(args.warmup):
        out = step(mix)
    if multi and args.warmup == 0:
        dist.barrier()                       # creates the communicator (and prints its banner) before stdout is restored
    fence()
    if saved_stdout is not None:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    model.profile_begin()
    import contextlib
    from demucs_amd import distributed as _dd
    with (_dd.collect_timing() if multi else contextlib.nullcontext([])) as dist_events:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step(mix)
        fence()
        elapsed = time.perf_counter() - t0
    dist_phases = _dd.timing_summary(dist_events) if multi else None
    rows_timed = model.profile_end()
    log(f"timed region done: {elapsed / args.steps * 1e3:.2f} ms per step")
    assert out.shape == (1, 4, 2, length) and out.device == dev and bool(torch.isfinite(out[0, 0, 0, ::997]).all())
    del out
    # Per-kernel roofline pass.  In the timed region the waveform branch runs on a side stream beside the spectral branch, so a
    # kernel's event-bracketed duration there includes the time it shares the GPU with the other branch's kernels (those
    # figures are kept as roofline.in_timed_region).  What the kernel itself reaches is timed right after, on the same inputs,
    # with one kernel on the GPU at a time (mi_set_two_streams(0)): same HIP events on the launch stream.
    from demucs_amd import _lib as _L
    iso_steps = 0 if args.no_iso_pass else max(1, min(args.steps, 3))
    rows = rows_timed
    if iso_steps:
        old_two = _L.load().mi_set_two_streams(0)
        model.profile_begin()
        for _ in range(iso_steps):
            step(mix)
        torch.cuda.synchronize(dev)
        rows = model.profile_end()
        _L.load().mi_set_two_streams(old_two)
    if multi:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        sec_per_step = elapsed / args.steps
        dom = max(rows, key=lambda r: r["ms"])
        dom_ms = dom["ms"] / dom["launches"]
        algo_tflops = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        x6 = dom["name"].startswith("conv_gemm_x6")
        pipe = "bf16" if x6 else dom.get("pipe", args.dtype if args.dtype != "f32" else "f32")
        if dom["name"].startswith("dconv") or dom["name"].startswith("stft") or dom["name"].startswith("istft"):
            bound, achieved, peak, unit = "hbm", dom["bytes"] / (dom["ms"] * 1e-3) / 1e9, HBM_PEAK_GBPS, "GB/s"
        else:
            # conv_gemm_x6 classes run fp32 operands as 3 exact bf16 terms x 6 bf16 MFMA products: the matrix pipe is
            # priced by the bf16 flops it actually issues (6 x algorithmic) against the dense bf16 peak
            bound, achieved, peak, unit = "mfma", algo_tflops * (X6_PRODUCTS if x6 else 1), MFMA_PEAK_TFLOPS[pipe], "TFLOP/s"
        total_ms = sum(r["ms"] for r in rows)
        # the same class inside the timed region (two streams: its launches share the GPU with the other branch's kernels)
        tdom = [r for r in rows_timed if r["name"] == dom["name"]]
        timed_roof = None
        if tdom:
            t_ach = (tdom[0]["bytes"] if bound == "hbm" else tdom[0]["flops"] * (X6_PRODUCTS if x6 else 1)) / (tdom[0]["ms"] * 1e-3) / (1e9 if bound == "hbm" else 1e12)
            timed_roof = {"avg_launch_ms": round(tdom[0]["ms"] / tdom[0]["launches"], 4), "launches": tdom[0]["launches"],
                          "achieved": round(t_ach, 2), "frac": round(t_ach / peak, 4),
                          "sum_of_class_ms_per_step": round(sum(r["ms"] for r in rows_timed) / args.steps, 2),
                          "note": "event-bracketed durations overlap across the two streams (their per-step sum exceeds ms_per_step): "
                                  "agrees with profiles/round4_f32_kernel_stats.csv (the default command)"}
        traffic, traffic_src = None, None        # HBM bytes per launch from the committed PMC passes of this same command
        for tname in ("round4_traffic.json", "round3_traffic.json", "round2_traffic.json", "round1_traffic.json"):
            tpath = os.path.join(ROOT, "profiles", tname)
            if traffic is None and os.path.exists(tpath) and world == 1 and args.batch == 32 and seconds == TRACK_SECONDS:
                entry = json.load(open(tpath)).get(dom["name"] + ("" if args.dtype == "f32" else "@" + args.dtype))
                if entry:
                    traffic = entry["traffic_bytes_per_launch"]
                    traffic_src = f"profiles/{tname} (separate rocprofv3 --pmc passes of this command, replayed here): " + entry["method"]
        dtype_words = {"f32": "fp32 (fp32 MFMA, exact)", "bf16": "bf16 MFMA operands, fp32 accumulate / statistics / softmax / iSTFT",
                       "f16": "fp16 MFMA operands, fp32 accumulate / statistics / softmax / iSTFT"}[args.dtype]
        result = {
            "metric": "real-time factor (audio-sec/wall-sec) htdemucs 4-stem 44.1kHz stereo, 1/8 GPU",
            "value": round(length / SR / sec_per_step, 2), "unit": "audio-sec/wall-sec",
            "value_span": "mix resident in HBM when the clock starts, stems left in HBM (the bench contract: a PCIe-inclusive rate is never "
                          "`value`); SURVEY.md 8(d)'s span -- host mix in, host stems out, PCIe inclusive -- is `value_host_to_host` "
                          "(details: `host_to_host`), and the user-facing Separator entry `separator_host_to_host`",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(sec_per_step * 1e3, 3),
            "higher_is_better": True, "scaling": "strong" if multi else "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": (f"BASELINE configs[{1 if seconds == TRACK_SECONDS else 3}]: htdemucs 4-stem, {dtype_words}, "
                                    f"{seconds // 60}-min synthetic 44.1 kHz stereo track resident in HBM when the clock starts, stems "
                                    f"left in HBM, segment=7.8 s overlap=0.25 shifts=0, {n_segments} segments, {args.batch} segments "
                                    "per batched forward, random-init weights (synthetic_state_dict seed 0)"
                                    + ("; the SAME fixed track at every N > 1 (strong scaling); N = 1 of that curve is "
                                       "`fixed_track` in the N = 1 line" if multi else "")),
                       "parallelism": (f"segments sharded over {world} GPUs by track interval, RCCL all-gather of the stem slabs over xGMI, one "
                                       "asynchronous piece per batched forward under the next forward"
                                       if multi else "one GPU")},
            "roofline": {"bound": bound, "kernel": dom["name"], "achieved": round(achieved, 2), "peak": peak,
                         "unit": unit, "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "measured": (f"HIP events on the launch stream over {iso_steps} passes of the same workload run right after the timed "
                                      "region with ONE kernel on the GPU at a time (mi_set_two_streams(0)); agrees with "
                                      "profiles/round4_f32_one_stream_kernel_stats.csv") if iso_steps else
                                     "HIP events on the launch streams over the timed region itself (--no-iso-pass: durations of the two streams overlap)",
                         "in_timed_region": timed_roof,
                         "pipe": ("bf16 MFMA, 6 products per fp32 MAC (exact 3-term operand split, fp32 accumulate)" if x6
                                  else f"{pipe} MFMA" if bound == "mfma" else "HBM"),
                         "algorithmic_bytes_per_launch": round(dom["bytes"] / dom["launches"]),
                         "algorithmic_flops_per_launch": round(dom["flops"] / dom["launches"]),
                         "launches": dom["launches"], "avg_launch_ms": round(dom_ms, 4),
                         "share_of_instrumented_time": round(dom["ms"] / total_ms, 3)},
            "kernels_note": f"per-class totals of the {iso_steps or args.steps} passes behind roofline.measured",
            "kernels_steps": iso_steps or args.steps,
            "kernels": [{"name": r["name"], "launches": r["launches"], "ms": round(r["ms"], 3),
                         "tflops": round(r["flops"] / (r["ms"] * 1e-3) / 1e12, 2),
                         "gbps": round(r["bytes"] / (r["ms"] * 1e-3) / 1e9, 1)} for r in sorted(rows, key=lambda r: -r["ms"])],
            "whole_path_tflops": round(FLOP_PER_SEGMENT * n_segments / sec_per_step / 1e12, 2),
        }
        if multi:
            # rank 0's phases of a sharded step (device events): its own segment forwards, the RCCL all-gather of the stem
            # slabs, the stitch + normalisation every rank repeats -- the last two are the serial tail of the design
            result["sharded_step_phases_rank0"] = dist_phases
        if not multi and not args.no_host_leg:
            log("host -> host legs")
            # apply_model entry with the mix on the host -> stems materialised on the host
            host_mix = torch.empty(mix.shape, dtype=torch.float32, pin_memory=True)
            host_mix.copy_(mix)
            torch.cuda.synchronize(dev)
            step(host_mix)
            times = []
            for _ in range(max(3, min(args.steps, 7))):
                t1 = time.perf_counter()
                host_out = step(host_mix)
                times.append(time.perf_counter() - t1)
            assert host_out.device.type == "cpu" and host_out.shape == (1, 4, 2, length)
            med = statistics.median(times)
            # SURVEY.md 8(d) defines the metric on THIS span; the bench contract keeps `value` HBM-resident (a PCIe-inclusive rate is
            # never `value`), so the span's figure travels beside it under its own top-level key
            result["value_host_to_host"] = round(length / SR / med, 2)
            result["host_to_host"] = {"value": round(length / SR / med, 2), "unit": "audio-sec/wall-sec", "ms_per_step": round(med * 1e3, 3),
                                      "runs": len(times), "span": "apply_model entry with a pinned host mix (63.5 MB H2D) -> the 254 MB of "
                                      "stems in a pinned host tensor (D2H), SURVEY.md 8(d); median"}
            # the user-facing entry (demucs/api.py:241-291): Separator.separate_tensor with the raw host wav -- H2D, mono mean / std
            # reduction, normalise, apply_model, de-normalise, D2H, all on the device between the two copies
            from demucs_amd.api import Separator
            sep = Separator(model, device=dev, shifts=0, overlap=0.25, split=True)
            wav = host_mix[0]
            sep.separate_tensor(wav)
            times = []
            for _ in range(max(3, min(args.steps, 7))):
                t1 = time.perf_counter()
                _, stems = sep.separate_tensor(wav)
                times.append(time.perf_counter() - t1)
            assert all(v.device.type == "cpu" and v.shape == (2, length) for v in stems.values())
            med_s = statistics.median(times)
            result["separator_host_to_host"] = {"value": round(length / SR / med_s, 2), "unit": "audio-sec/wall-sec",
                                                "ms_per_step": round(med_s * 1e3, 3), "runs": len(times),
                                                "vs_host_to_host": round(med / med_s, 4),
                                                "span": "demucs_amd.api.Separator.separate_tensor(host wav) -> dict of host stems: the "
                                                        "host_to_host span plus the mono mean / std reduction and the two affine passes, "
                                                        "all device kernels (mi_mono_stats, mi_track_affine); median"}
            del host_out, host_mix, stems, wav
            log("single-segment leg")
            # BASELINE configs[0]'s workload on the GPU: ONE 7.8 s segment (2 forwards: apply_model schedules offsets 0 and 0.75 SL)
            # through the user-facing entry, host wav -> host stems: the small-batch latency of the path
            m1 = HTDemucs(cfg.sources, max_batch=2, compute_dtype=args.dtype)
            m1.load_state_dict(sd)
            m1.to(dev).eval()
            sep1 = Separator(m1, device=dev, shifts=0, overlap=0.25, split=True)
            seg_wav = torch.empty(2, cfg.segment_length, dtype=torch.float32, pin_memory=True)
            seg_wav.copy_(mix[0, :, :cfg.segment_length])
            torch.cuda.synchronize(dev)
            for _ in range(3):
                sep1.separate_tensor(seg_wav)
            times = []
            for _ in range(15):
                t1 = time.perf_counter()
                _, stems1 = sep1.separate_tensor(seg_wav)
                times.append(time.perf_counter() - t1)
            med1 = statistics.median(times)
            fwd = []
            seg_dev = mix[:1, :, :cfg.segment_length].contiguous()
            for _ in range(15):
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                m1.forward_segments(seg_dev)
                torch.cuda.synchronize(dev)
                fwd.append(time.perf_counter() - t1)
            result["single_segment"] = {"value": round(cfg.segment_length / SR / med1, 2), "unit": "audio-sec/wall-sec",
                                        "ms_per_call": round(med1 * 1e3, 3), "forward_b1_ms": round(statistics.median(fwd) * 1e3, 3), "runs": 15,
                                        "span": "configs[0]'s workload (one 343 980-sample segment, 2 forwards in one batch of 2) through "
                                                "Separator.separate_tensor, pinned host wav -> host stems; forward_b1_ms = one B = 1 "
                                                "mi_model_forward, device-resident, synchronised; medians"}
            del stems1, seg_wav, seg_dev
            m1.release()
        if not multi and not args.no_fixed_leg and seconds != FIXED_SECONDS:
            log("fixed 60-minute track leg")
            del mix
            torch.cuda.empty_cache()
            long_mix = make_mix(FIXED_SECONDS)
            step(long_mix)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            o = step(long_mix)
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t1
            assert bool(torch.isfinite(o[0, 0, 0, ::99991]).all())
            result["fixed_track"] = {"seconds": FIXED_SECONDS, "segments": len(range(0, FIXED_SECONDS * SR, stride)),
                                     "value": round(FIXED_SECONDS / dt, 2), "unit": "audio-sec/wall-sec", "ms_per_step": round(dt * 1e3, 2),
                                     "note": "configs[3]'s 60-minute track on ONE GPU, HBM-resident like `value`: the N = 1 point of the "
                                             "fixed-length (strong-scaling) curve that `--gpus N > 1` continues"}
            del o, long_mix
        if not multi and not args.no_modes_leg and args.dtype == "f32":
            # the reduced-precision compute modes on the same 3-minute track (BASELINE configs[2]'s bf16, configs[4]'s fp16 +
            # 6 sources): HBM-resident like `value`, own roofline against the dense bf16 / fp16 MFMA peak
            result["modes"] = {}
            mix3 = make_mix(TRACK_SECONDS)
            cfg6 = HTDemucsConfig(sources=["drums", "bass", "other", "vocals", "guitar", "piano"])
            for tag, mcfg, dt in (("htdemucs bf16", cfg, "bf16"), ("htdemucs_6s fp16", cfg6, "f16")):
                log(f"mode {tag}")
                m2 = HTDemucs(mcfg.sources, max_batch=args.batch, compute_dtype=dt)
                m2.load_state_dict(sd if mcfg is cfg else synthetic_state_dict(mcfg, 0))
                m2.to(dev).eval()
                P.apply_model(m2, mix3, shifts=0, split=True, overlap=0.25, device=dev)
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                for _ in range(3):
                    o = P.apply_model(m2, mix3, shifts=0, split=True, overlap=0.25, device=dev)
                torch.cuda.synchronize(dev)
                dt_s = (time.perf_counter() - t1) / 3
                old_two = _L.load().mi_set_two_streams(0)         # per-kernel figures: one kernel on the GPU at a time
                m2.profile_begin()
                for _ in range(2):
                    P.apply_model(m2, mix3, shifts=0, split=True, overlap=0.25, device=dev)
                torch.cuda.synchronize(dev)
                rows2 = m2.profile_end()
                _L.load().mi_set_two_streams(old_two)
                assert bool(torch.isfinite(o[0, :, 0, ::997]).all())
                d2 = max(rows2, key=lambda r: r["ms"])
                tf = d2["flops"] / (d2["ms"] * 1e-3) / 1e12
                result["modes"][tag] = {
                    "dtype": dt, "sources": len(mcfg.sources), "value": round(TRACK_SECONDS / dt_s, 2), "unit": "audio-sec/wall-sec",
                    "ms_per_step": round(dt_s * 1e3, 3), "steps": 3,
                    "roofline": {"bound": "mfma", "kernel": d2["name"], "achieved": round(tf, 2), "peak": MFMA_PEAK_TFLOPS[dt], "unit": "TFLOP/s",
                                 "frac": round(tf / MFMA_PEAK_TFLOPS[dt], 4), "avg_launch_ms": round(d2["ms"] / d2["launches"], 4),
                                 "algorithmic_gbps": round(d2["bytes"] / (d2["ms"] * 1e-3) / 1e9, 1),
                                 "note": "one-kernel-at-a-time passes after the timed ones; the residual stream stays float32 in HBM in these "
                                         "modes (340 MB of residual read + output write per out_proj / lin2 launch)"},
                    "kernels": [{"name": r["name"], "ms": round(r["ms"], 3), "tflops": round(r["flops"] / (r["ms"] * 1e-3) / 1e12, 1)}
                                for r in sorted(rows2, key=lambda r: -r["ms"])[:6]]}
                del o
                m2.release()
            # BASELINE configs[2] as written: htdemucs_ft = bag of 4 models with one-hot per-source weights, shifts = 2 (random.seed(0)),
            # overlap 0.25, bf16, the same 3-minute track: 4 x 2 passes of ~31 segments, averages accumulated in HBM
            import random as _random
            log("mode htdemucs_ft bag of 4, shifts 2, bf16")
            bag_models = []
            for ws in (10, 11, 12, 13):
                bm = HTDemucs(cfg.sources, max_batch=args.batch, compute_dtype="bf16")
                bm.load_state_dict(synthetic_state_dict(cfg, ws))
                bag_models.append(bm.to(dev).eval())
            ft = P.BagOfModels(bag_models, [[1.0 if k == j else 0.0 for k in range(4)] for j in range(4)])
            _random.seed(0)
            P.apply_model(ft, mix3, shifts=2, split=True, overlap=0.25, device=dev)
            torch.cuda.synchronize(dev)
            times = []
            for _ in range(3):
                _random.seed(0)
                t1 = time.perf_counter()
                o = P.apply_model(ft, mix3, shifts=2, split=True, overlap=0.25, device=dev)
                torch.cuda.synchronize(dev)
                times.append(time.perf_counter() - t1)
            dt_s = sorted(times)[1]
            assert o.shape == (1, 4, 2, TRACK_SECONDS * SR) and bool(torch.isfinite(o[0, :, 0, ::997]).all())
            result["modes"]["htdemucs_ft bag4 shifts2 bf16"] = {
                "dtype": "bf16", "sources": 4, "models": 4, "shifts": 2, "value": round(TRACK_SECONDS / dt_s, 2), "unit": "audio-sec/wall-sec",
                "ms_per_step": round(dt_s * 1e3, 2), "steps": 3, "segment_forwards": 8 * 31,
                "note": "BASELINE configs[2]: BagOfModels of 4 htdemucs members (weight seeds 10-13, one-hot weights as "
                        "remote/htdemucs_ft.yaml), shifts=2, overlap 0.25, HBM-resident mix and result; median of 3"}
            del o, ft
            for bm in bag_models:
                bm.release()
            del mix3
            # BASELINE configs[4], second model: the hdemucs_mmi architecture (Hybrid Demucs v3: BLSTM + LocalState, 44 s
            # segments as remote/hdemucs_mmi.yaml sets) in the fp16 mode on a 3-minute track: the five full chunks in one batched
            # forward, the 15 s tail chunk on the side engine under it
            from demucs_amd.hdemucs import HDemucs
            from demucs_amd.hdemucs_weights import HDemucsConfig, synthetic_hdemucs_state_dict
            log("mode hdemucs_mmi fp16")
            hcfg = HDemucsConfig()
            hm = HDemucs(hcfg.sources, max_batch=5, compute_dtype="f16")
            hm.load_state_dict(synthetic_hdemucs_state_dict(hcfg, 0))
            hm.to(dev).eval()
            hbag = P.BagOfModels([hm], segment=44)
            hmix = make_mix(TRACK_SECONDS)
            P.apply_model(hbag, hmix, shifts=0, split=True, overlap=0.25, device=dev)
            torch.cuda.synchronize(dev)
            times = []
            for _ in range(3):
                t1 = time.perf_counter()
                o = P.apply_model(hbag, hmix, shifts=0, split=True, overlap=0.25, device=dev)
                torch.cuda.synchronize(dev)
                times.append(time.perf_counter() - t1)
            dt_s = sorted(times)[1]                       # median of three passes
            # what bounds this mode: the BLSTM recurrence of encoder layers 4 / 5 -- 1 600 DEPENDENT launches per forward, so a
            # latency chain, not a bandwidth or matrix-pipe figure: timed (HIP events on the launch stream) in one more pass
            hm.profile_begin()
            P.apply_model(hbag, hmix, shifts=0, split=True, overlap=0.25, device=dev)
            hrows = hm.profile_end()
            lstm = [r for r in hrows if r["name"] in ("lstm_persist_kernel", "lstm_step_kernel")]
            persist = bool(lstm) and lstm[0]["name"] == "lstm_persist_kernel"
            # floor of one time step: the persistent kernel (lstm.hip) pays one cross-CU hand-off of the hidden state per step
            # (MI355X_MICROARCH.md price list, row "handoff-1to1": 0.8-1.0 us on an idle chip); the per-step launch chain of
            # MI_LSTM_STEPS=1 one dependent kernel boundary (row "boundary": 1.45 us)
            BOUNDARY_US = 1.0 if persist else 1.45
            h_roof = None
            if lstm:
                us = lstm[0]["ms"] * 1e3 / lstm[0]["launches"]
                h_roof = {"bound": "latency", "kernel": lstm[0]["name"], "time_steps": lstm[0]["launches"], "avg_step_us": round(us, 3),
                          "floor_us": BOUNDARY_US, "frac": round(BOUNDARY_US / us, 4), "chain_ms": round(lstm[0]["ms"], 3),
                          "share_of_step": round(lstm[0]["ms"] * 1e-3 / dt_s, 3),
                          "note": "the BLSTM recurrence of encoder layers 4 / 5: 1 600 DEPENDENT time steps per forward (8 sequences of 200), "
                                  "main engine's batched forward only (the tail chunk's own chain runs under it on the side stream); "
                                  + ("ONE persistent launch per sequence, hidden state exchanged between workgroups as tagged granules: floor = "
                                     "one cross-CU hand-off per step" if persist else "one launch per step: floor = one dependent kernel boundary per step")
                                  + "; peak / achieved in microseconds per time step",
                          "achieved": round(us, 3), "peak": BOUNDARY_US, "unit": "us/step"}
            assert o.shape == (1, 4, 2, TRACK_SECONDS * SR) and bool(torch.isfinite(o[0, :, 0, ::997]).all())
            result["modes"]["hdemucs_mmi fp16"] = {"dtype": "f16", "sources": 4, "value": round(TRACK_SECONDS / dt_s, 2), "unit": "audio-sec/wall-sec",
                                                   "ms_per_step": round(dt_s * 1e3, 2), "steps": 3,
                                                   "device_bytes": hm.device_bytes(), "roofline": h_roof,
                                                   "note": "6 chunks (overlap 0.25): five of 44 s in one batched forward, the 15 s tail on the side engine and a side stream under it"}
            del o, hmix
            hm.release()
        if not multi and not args.no_cpu_baseline:
            log(f"cpu baseline on {usable_cores()} threads, then 1")
            result["cpu_baseline"] = cpu_baseline(sd, cfg.sources)
        log("done")
        print(json.dumps(result), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
