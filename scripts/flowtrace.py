"""Summarise a NG_SGS_TRACE dump (development aid): what the chunks of one dataflow SGS sweep did.

NG_SGS_TRACE=<path> makes every sweep write <path>.fwd / <path>.bwd: per chunk (dispatch order) the
100 MHz clock when its consumer started, after its first round of steps, at half-way and at the end, and
lane 0's counters: ticks in recovery waits, number of them, ticks waiting for the loaders and for the
publisher.  (The dump synchronises the stream: timings of the solve itself are not meaningful then.)"""
import sys

import numpy as np

t = np.loadtxt(sys.argv[1], dtype=np.int64)
t0 = t[:, 1].min()
us = lambda a: (a - t0) / 100.0
start, first, half, end = us(t[:, 1]), us(t[:, 2]), us(t[:, 3]), us(t[:, 4])
print(f"chunks {len(t)}  sweep {end.max():.1f} us")
print(f"a chunk from its first steps to its end: median {np.median(end - first):.1f}  min {np.min(end - first):.1f}  "
      f"max {np.max(end - first):.1f} us; its second half: median {np.median(end - half):.1f} us")
print("start of the consumer (dispatch): first chunks", start[:4].round(1), "last", start[-4:].round(1))
print(f"lane 0 of a chunk: recovery waits median {np.median(t[:, 6]):.0f}, time in them median {np.median(t[:, 5]) / 100:.1f} us; "
      f"waiting for the loaders median {np.median(t[:, 7]) / 100:.1f} us (max {t[:, 7].max() / 100:.1f}), "
      f"for the publisher median {np.median(t[:, 8]) / 100:.1f} us")
k = max(len(t) // 8, 1)
print("end of every", k, "th chunk:", end[::k].round(0))
