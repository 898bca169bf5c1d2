"""Prints what the host gives us (cores, cgroup quota) and how torch CPU matmul scales with threads -- used to pick the
thread count of the CPU baseline / oracle on the GPU box."""
import os
import time

import torch

print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads())
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    if os.path.exists(p):
        print(p, open(p).read().strip())
os.system("lscpu | grep -E 'Model name|^CPU\\(s\\)|Thread|Socket' ; free -g | head -2")
a = torch.randn(2048, 2048)
for n in (1, 4, 8, 16, 32, 64, 128):
    if n > (os.cpu_count() or 1):
        break
    torch.set_num_threads(n)
    a @ a
    t = time.time()
    for _ in range(5):
        a @ a
    dt = (time.time() - t) / 5
    print(f"threads {n:4d}: {2 * 2048 ** 3 / dt / 1e9:8.1f} GFLOP/s")
