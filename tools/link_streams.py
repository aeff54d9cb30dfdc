#!/usr/bin/env python3
"""How much the host link moves with k streams per direction (pinned memory, torch copies): the experiment behind ipx_link_probe's
choice of streams.  Prints GB/s up, down and summed for k = 1, 2, 3, 4 and for chunk sizes of 8 / 32 / 128 MiB."""
import sys
import time

import torch


def run(k, chunk_mib, total_mib=1024, both=True, up=True, down=True):
    n = total_mib // chunk_mib
    hu = [torch.empty(chunk_mib << 20, dtype=torch.uint8).pin_memory() for _ in range(min(n, 8))]
    hd = [torch.empty(chunk_mib << 20, dtype=torch.uint8).pin_memory() for _ in range(min(n, 8))]
    du = [torch.empty(chunk_mib << 20, dtype=torch.uint8, device="cuda") for _ in range(min(n, 8))]
    dd = [torch.zeros(chunk_mib << 20, dtype=torch.uint8, device="cuda") for _ in range(min(n, 8))]
    su = [torch.cuda.Stream() for _ in range(k)]
    sd = [torch.cuda.Stream() for _ in range(k)]
    best = 1e9
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            if up:
                with torch.cuda.stream(su[i % k]):
                    du[i % len(du)].copy_(hu[i % len(hu)], non_blocking=True)
            if down:
                with torch.cuda.stream(sd[i % k] if both else su[i % k]):
                    hd[i % len(hd)].copy_(dd[i % len(dd)], non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if rep:
            best = min(best, dt)
    gb = total_mib * 1.048576e-3
    return (gb / best if up else 0), (gb / best if down else 0)


def main():
    for chunk in (8, 32, 128):
        for k in (1, 2, 3, 4):
            u, _ = run(k, chunk, down=False)
            _, d = run(k, chunk, up=False)
            bu, bd = run(k, chunk)
            su, sd_ = run(k, chunk, both=False)      # up and down of a chunk on the SAME stream, k streams (a feeder's pattern)
            print("chunk %4d MiB  k=%d  up %.1f  down %.1f  both %.1f + %.1f = %.1f  same-stream %.1f" % (chunk, k, u, d, bu, bd, bu + bd, su + sd_), flush=True)


if __name__ == "__main__":
    sys.exit(main())
