import sys, time, torch
sys.path.insert(0, "/root/repo")
from demucs_amd.htdemucs import HTDemucs
from demucs_amd.weights import HTDemucsConfig, synthetic_state_dict
cfg = HTDemucsConfig()
for B in (1, 2, 4):
    m = HTDemucs(cfg.sources, max_batch=B); m.load_state_dict(synthetic_state_dict(cfg, 0)); m.to("cuda")
    x = torch.randn(B, 2, 343980, device="cuda") * 0.1
    for _ in range(3): y = m(x)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20): y = m(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 20
    # CPU-side enqueue time only
    t = time.perf_counter()
    for _ in range(20): y = m(x)
    enq = (time.perf_counter() - t) / 20
    torch.cuda.synchronize()
    print(f"B={B}: {dt*1e3:.2f} ms per forward ({dt*1e3/B:.2f} per segment), CPU enqueue {enq*1e3:.2f} ms")
    m.release()
