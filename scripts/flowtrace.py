"""Summarise a NG_SGS_TRACE dump (development aid): per-chunk start lag and duration of a dataflow sweep."""
import sys
import numpy as np
t = np.loadtxt(sys.argv[1], dtype=np.int64)
nyb = int(sys.argv[2]) if len(sys.argv) > 2 else 32
t0 = t[:, 1].min()
us = lambda a: (a - t0) / 100.0
start, first, half, end = us(t[:, 1]), us(t[:, 2]), us(t[:, 3]), us(t[:, 4])
print(f"chunks {len(t)}  sweep {end.max():.1f} us")
print(f"duration of a chunk (first step done -> end): median {np.median(end - first):.1f}  min {np.min(end - first):.1f}  max {np.max(end - first):.1f} us")
print(f"second half of a chunk: median {np.median(end - half):.1f} us")
print("dispatch (start) time: first", start[:4].round(1), "last", start[-4:].round(1))
nz = len(t) // nyb
F = first.reshape(nz, nyb)
E = end.reshape(nz, nyb)
print("first-step-done time along yb at Z'=0:", F[0, :8].round(1))
print("first-step-done time along Z' at yb=0:", F[:8, 0].round(1))
print("end time along the diagonal:", np.array([E[i, i] for i in range(min(nz, nyb))]).round(0)[::4])
if t.shape[1] >= 9:
    # per-lane counters of lane 0 only: a lower bound of the wave's waits
    print(f"lane 0 of a chunk: polls median {np.median(t[:, 6]):.0f} (of {270} steps), time in polls median {np.median(t[:, 5]) / 100:.1f} us, "
          f"waiting for loaders median {np.median(t[:, 7]) / 100:.1f} us max {t[:, 7].max() / 100:.1f}, for the publisher median {np.median(t[:, 8]) / 100:.1f} us")
print("per-hop lag in y (median of first-step-done differences):", np.median(np.diff(F, axis=1)).round(2), " in z:", np.median(np.diff(F, axis=0)).round(2))
