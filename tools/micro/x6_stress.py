"""Kernel-level reproducer: P processes share the GPU, each repeats ONE split-bf16 GEMM launch and compares every
result bitwise with its first.   python tools/micro/x6_stress.py <procs> <plain 0|1> [iters]"""
import ctypes as C, os, sys, torch, torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

def worker(rank, plain, iters, x6):
    mix = os.environ.get("MIX")                      # e.g. "11,10,01": per-rank (plain, x6) -> different kernels share the GPU
    if mix:
        plain, x6 = int(mix.split(",")[rank][0]), int(mix.split(",")[rank][1])
    from demucs_amd import _lib
    from gpu_helpers import EPI_LINEAR, ktab, pack_w
    torch.manual_seed(rank)
    B, M, K, T = int(os.environ.get("SB", 8)), int(os.environ.get("SM", 2048)), int(os.environ.get("SK", 512)), int(os.environ.get("ST", 2688))
    W = torch.randn(M, K) * 0.05
    wt, bias, M_, Mpad, K_, Kpad, tile = pack_w(W, torch.randn(M))
    kt = ktab(K, 1, 1, 1, 1, 0, 0, T, T, Kpad)
    x = torch.randn(B, K, T, device="cuda")
    y = torch.empty(B, M, T, device="cuda")
    lib = _lib.load(); st = C.c_void_p(_lib.current_stream_ptr())
    wx = torch.empty(6 * Kpad * Mpad, dtype=torch.uint8, device="cuda")
    _lib.check(lib.mi_conv_pack_split(wt.data_ptr(), Kpad, Mpad, tile, wx.data_ptr(), st), "pack")
    d = _lib.MiConvDesc()
    kw = dict(wt=wt, M=M, Mpad=Mpad, K=K, Kpad=Kpad, ktab=kt, x=x, x_bstride=K * T, B=B, D1=1, D2=T, O1=1, O2=T, S1=1, S2=1,
              epi=EPI_LINEAR, bias=bias, y=y, y_bstride=M * T, y_cstride=T, tile_m=tile, plain=plain, wx=wx if x6 else 0)
    for f, _ in _lib.MiConvDesc._fields_:
        v = kw.get(f, 0)
        setattr(d, f, v.data_ptr() if isinstance(v, torch.Tensor) else v)
    ref, bad, worst = None, 0, 0.0
    chain = int(os.environ.get("CHAIN", "0"))        # 1: producer kernel -> conv -> consumer kernel back to back, no sync
    xs = [x.clone(), torch.randn_like(x)]
    refs = [None, None]
    for it in range(iters):
        if chain:
            x.copy_(xs[it & 1])                      # producer of the conv input, same stream, no sync
            _lib.check(lib.mi_conv_forward(C.byref(d), st), "conv")
            z = y * 1.0                               # consumer
            y.zero_()                                 # and something that overwrites the output right after
            torch.cuda.synchronize()
            y.copy_(z)
            ref = refs[it & 1]
            if ref is None:
                refs[it & 1] = y.clone()
                continue
        else:
            y.zero_()
            for _ in range(int(os.environ.get("BURST", "1"))):      # back-to-back launches without a host sync
                _lib.check(lib.mi_conv_forward(C.byref(d), st), "conv")
            torch.cuda.synchronize()
            if ref is None:
                ref = y.clone()
                continue
        if not torch.equal(y, ref):
            bad += 1
            diff = (y - ref).abs()
            worst = max(worst, float(diff.max()))
            if bad <= 3:
                idx = diff.flatten().nonzero().flatten()
                b, m, t = idx // (M * T), (idx // T) % M, idx % T
                print(f"rank {rank} it {it}: {idx.numel()} elements differ; b {b.min().item()}-{b.max().item()} m {m.min().item()}-{m.max().item()} "
                      f"t {t.min().item()}-{t.max().item()} (t tile {t.min().item() // 128}..{t.max().item() // 128})", flush=True)
    print(f"rank {rank}: {bad}/{iters - 1} runs differ, worst {worst:.3e}", flush=True)

if __name__ == "__main__":
    procs, plain = int(sys.argv[1]), int(sys.argv[2])
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    x6 = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    mp.spawn(worker, args=(plain, iters, x6), nprocs=procs, join=True)
