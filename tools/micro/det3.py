"""3 processes share the GPU; each repeats the same forward and reports which taps differ from its first run."""
import sys, torch, torch.multiprocessing as mp
sys.path.insert(0, ".")
TAPS = ["x0", "xt0", "enc0", "enc1", "enc2", "enc3", "tenc0", "tenc1", "tenc2", "tenc3", "tr_f", "tr_t", "yspec", "ytime"]

def worker(rank):
    import os, torch
    only = os.environ.get("X6_RANKS")                 # e.g. "0": only these ranks run the split-bf16 path, the others fp32
    if only is not None:
        if str(rank) in only.split(","):
            os.environ["MI_X6"] = "1"
        else:
            os.environ.pop("MI_X6", None)
        print(f"rank {rank}: MI_X6={os.environ.get('MI_X6')}", flush=True)
    if str(rank) in os.environ.get("BF16_RANKS", "").split(","):
        # neighbour load that never touches this package: dense bf16 GEMMs from the vendor BLAS (same bf16 MFMA instructions)
        import time
        a = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16); b = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
        t0 = time.time()
        while time.time() - t0 < float(os.environ.get("BF16_SECONDS", "12")):
            for _ in range(20):
                c = a @ b
            torch.cuda.synchronize()
        print(f"rank {rank}: bf16 matmul load done", flush=True)
        return
    if rank == 0 and os.environ.get("CULPRIT"):
        import subprocess                              # an external program as the neighbour, e.g. "tools/micro/mfma_neighbour 1 12"
        print(subprocess.run(os.environ["CULPRIT"].split(), capture_output=True, text=True).stdout.strip(), flush=True)
        return
    from demucs_amd.htdemucs import HTDemucs
    from demucs_amd.weights import HTDemucsConfig, synthetic_state_dict
    from demucs_amd.synth import synth_mix
    cfg = HTDemucsConfig()
    m = HTDemucs(cfg.sources, max_batch=2)
    m.load_state_dict(synthetic_state_dict(cfg, 4)); m.to("cuda")
    x = torch.stack([torch.from_numpy(synth_mix(50 + i, 343980, "tones")) for i in range(2)]).cuda()
    ref, reft = None, None
    for it in range(int(os.environ.get("DET_ITERS", "8"))):
        y = m.forward_segments(x).clone()
        taps = {t: m.tap(t, 2).clone() for t in TAPS}
        torch.cuda.synchronize()
        if ref is None:
            ref, reft = y, taps
            continue
        if not torch.equal(y, ref):
            bad = [(t, float((taps[t] - reft[t]).abs().max())) for t in TAPS if not torch.equal(taps[t], reft[t])]
            print(f"rank {rank} iter {it}: out diff {float((y - ref).abs().max()):.3e}; taps: {bad}", flush=True)
    print(f"rank {rank} done", flush=True)

if __name__ == "__main__":
    mp.spawn(worker, nprocs=int(sys.argv[1]) if len(sys.argv) > 1 else 3, join=True)
