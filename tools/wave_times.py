#!/usr/bin/env python3
"""Summarise a VMR_DEBUG_TIMES file (-DSL_DEBUG build): per launch, how long the waves spend in the prologue, the step loop and the
epilogue, and how evenly they finish (development aid)."""
import sys
import numpy as np

def main(path):
    launches, cur, head = [], [], None
    for line in open(path):
        if line.startswith("launch"):
            if head is not None:
                launches.append((head, np.array(cur, dtype=np.float64)))
            head, cur = line.strip(), []
        else:
            cur.append([int(v) for v in line.split()])
    if head is not None:
        launches.append((head, np.array(cur, dtype=np.float64)))
    for head, t in launches[-6:]:
        t0 = t[:, 0].min()
        us = (t - t0) / 100.0   # 100 MHz -> microseconds
        q = lambda v: " ".join(f"{np.percentile(v, p):8.1f}" for p in (0, 10, 50, 90, 100))
        print(head)
        print("  start       (min p10 p50 p90 max) us:", q(us[:, 0]))
        print("  prologue    ", q(us[:, 1] - us[:, 0]))
        print("  step loop   ", q(us[:, 2] - us[:, 1]))
        print("  epilogue    ", q(us[:, 3] - us[:, 2]))
        print("  end         ", q(us[:, 3]))
        if t.shape[1] >= 8:
            print("  prologue: first loads issued", q(us[:, 4] - us[:, 0]), "| tables' barrier", q(us[:, 5] - us[:, 0]))
            print("  epilogue: nu share done     ", q(us[:, 6] - us[:, 2]), "| flush done", q(us[:, 7] - us[:, 2]))
        tail = us[:, 3] - us[:, 7]
        print("  after the flush (ticket, sums, finalize tail of the layer-last workgroups): p50 %.1f  p99 %.1f  max %.1f; 8 longest:" % (
            np.median(tail), np.percentile(tail, 99), tail.max()), np.round(np.sort(tail)[-8:], 1), "kernel end", round(us[:, 3].max(), 1))
        nw = int(head.split("tpb")[1]) // 64
        wg_end = us[:, 2].reshape(-1, nw)
        print("  loop end per workgroup: first/last wave spread (p50, max) us: %.1f %.1f" % (np.median(wg_end.max(1) - wg_end.min(1)), (wg_end.max(1) - wg_end.min(1)).max()))

if __name__ == "__main__":
    main(sys.argv[1])
