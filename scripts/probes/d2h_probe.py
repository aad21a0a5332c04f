"""Development aid: what the copy of a finished frame to pinned host memory costs on this box (the last stage of every
step of bench.py; the only device crossing the reference has is the opposite one, glTexImage2D of the frame,
gpu_and_windowing.c:371-376).  One copy, two concurrent copies of the halves on two streams into two pinned buffers,
for a strip, a 1080p frame and a 4K frame.  Run a second time with HSA_ENABLE_SDMA=0 in the environment to see the
blit-kernel path instead of the SDMA engines.  usage: d2h_probe.py"""
import os, time, torch
dev = torch.device("cuda", 0)
print("HSA_ENABLE_SDMA =", os.environ.get("HSA_ENABLE_SDMA", "(unset: SDMA engines)"))
for name, nbytes in (("1080p strip of 1/8", 1920 * 136 * 12), ("1080p frame", 1920 * 1080 * 12), ("4K frame", 3840 * 2160 * 12)):
    n = nbytes // 4
    src = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    dst = [torch.empty(n, dtype=torch.float32, pin_memory=True) for _ in range(2)]
    pageable = torch.empty(n, dtype=torch.float32)
    s = [torch.cuda.Stream(dev, priority=-1), torch.cuda.Stream(dev, priority=-1)]
    def timed(fn, reps=20):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps
    def one():
        with torch.cuda.stream(s[0]): dst[0].copy_(src, non_blocking=True)
    def two_halves():
        h = n // 2
        with torch.cuda.stream(s[0]): dst[0][:h].copy_(src[:h], non_blocking=True)
        with torch.cuda.stream(s[1]): dst[1][h:].copy_(src[h:], non_blocking=True)
    def two_whole():      # two frames at once (double-buffered destinations): the aggregate rate
        with torch.cuda.stream(s[0]): dst[0].copy_(src, non_blocking=True)
        with torch.cuda.stream(s[1]): dst[1].copy_(src, non_blocking=True)
    def to_pageable():
        pageable.copy_(src)
    t1, t2, t3, t4 = timed(one), timed(two_halves), timed(two_whole), timed(to_pageable, 5)
    print(f"{name:20s} {nbytes / 1e6:7.1f} MB: one copy {t1 * 1e3:7.3f} ms = {nbytes / t1 / 1e9:5.1f} GB/s | halves on two streams {t2 * 1e3:7.3f} ms = {nbytes / t2 / 1e9:5.1f} GB/s | "
          f"two whole copies at once {t3 * 1e3:7.3f} ms = {2 * nbytes / t3 / 1e9:5.1f} GB/s aggregate | pageable destination {t4 * 1e3:7.3f} ms = {nbytes / t4 / 1e9:5.1f} GB/s", flush=True)
