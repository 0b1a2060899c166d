"""Latency distribution of small RCCL collectives with one rank (tools only)."""
import os, sys, time
for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29534"), ("RANK", "0"), ("WORLD_SIZE", "1")):
    os.environ.setdefault(k, v)
import torch, torch.distributed as dist
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
x = torch.randn(256 * 1024, device=dev)
big = torch.randn(64, 1024, 1024, device=dev)
def run(name, fn, n=300, busy=False):
    ts = []
    for i in range(n):
        if busy:
            y = big * 1.0001  # some queued GPU work before the collective
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    ts = sorted(ts[10:])
    print(f"{name:34s} median {1e6 * ts[len(ts) // 2]:8.1f} us  p99 {1e6 * ts[int(len(ts) * .99)]:9.1f}  max {1e6 * ts[-1]:9.1f}")
run("all_reduce 1MB", lambda: dist.all_reduce(x))
run("all_reduce 1MB (after GPU work)", lambda: dist.all_reduce(x), busy=True)
out = torch.empty(1, x.numel(), device=dev)
run("all_gather_into_tensor", lambda: dist.all_gather_into_tensor(out, x))
e = torch.empty(0, 256, device=dev); s = torch.empty(0, 256, device=dev)
run("all_to_all_single (empty)", lambda: dist.all_to_all_single(e, s, [0], [0]))
run("elementwise only", lambda: x.mul_(1.0))
parts = [torch.randn(n, device=dev) for n in (32768, 256, 65536, 256, 65536, 256, 10240, 40, 65536, 256, 192000, 256)]
def fresh_allreduce():
    flat = torch.cat(parts); dist.all_reduce(flat)
    for q in parts: q.copy_(flat[:q.numel()])
run("cat + all_reduce + copies (grads)", fresh_allreduce)
a = torch.randn(169343, 256, device=dev); h = torch.empty(0, 256, device=dev)
run("cat [N,256] with empty halo", lambda: torch.cat([a, h], 0), busy=True)
def gather16():
    f = torch.randn(16384, 256, device=dev); o = torch.empty(16384, 256, device=dev); dist.all_gather_into_tensor(o, f)
run("all_gather fresh 16 MB", gather16, busy=True)
import gc
print("gc counts", gc.get_count(), "thresholds", gc.get_threshold(), "tracked objects", len(gc.get_objects()))
t = time.perf_counter(); gc.collect(); print(f"full gc.collect(): {1e3 * (time.perf_counter() - t):.1f} ms")
dist.destroy_process_group()
