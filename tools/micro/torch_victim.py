"""Are the vendor's own kernels victims too?  Rank 0 runs tools/micro/mfma_neighbour <variant>; ranks 1.. repeat a chain of
torch ops (fp32 matmul, conv2d, fft, softmax) and compare every result bitwise with their first."""
import os, subprocess, sys, torch, torch.multiprocessing as mp

def worker(rank, variant):
    if rank == 0:
        print(subprocess.run(["tools/micro/mfma_neighbour", str(variant), "10"], capture_output=True, text=True).stdout.strip(), flush=True)
        return
    torch.manual_seed(rank)
    a = torch.randn(4096, 4096, device="cuda"); b = torch.randn(4096, 4096, device="cuda")
    x = torch.randn(8, 64, 256, 256, device="cuda"); w = torch.randn(64, 64, 3, 3, device="cuda") * 0.05
    s = torch.randn(64, 1 << 16, device="cuda")
    def step():
        return [(a @ b).clone(), torch.nn.functional.conv2d(x, w, padding=1), torch.fft.rfft(s).abs(), torch.softmax(a, -1)]
    ref = step(); torch.cuda.synchronize()
    bad = [0, 0, 0, 0]
    for it in range(60):
        out = step(); torch.cuda.synchronize()
        for i, (o, r) in enumerate(zip(out, ref)):
            if not torch.equal(o, r):
                bad[i] += 1
    print(f"rank {rank}: mismatching runs of 60 -> matmul {bad[0]}, conv2d {bad[1]}, rfft {bad[2]}, softmax {bad[3]}", flush=True)

if __name__ == "__main__":
    mp.spawn(worker, args=(int(sys.argv[1]) if len(sys.argv) > 1 else 1,), nprocs=3, join=True)
