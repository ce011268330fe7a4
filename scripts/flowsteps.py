"""Summarise a NG_SGS_TRACE_STEPS dump (development aid): the step timelines of the chunks of one dataflow sweep.
usage: flowsteps.py <dump> nYb nZb [lag]   (tiles per direction; lag = By - 1: the producer's step a consumer's step waits for)"""
import sys

import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.int64)
nc, ms = int(raw[0]), int(raw[1])
t = raw[2:].reshape(nc, ms, 3)
nY, nZ = int(sys.argv[2]), int(sys.argv[3])
lag = int(sys.argv[4]) if len(sys.argv) > 4 else 5
idx, nxt = {}, 0
for d in range(nY + nZ - 1):
    for zb in range(max(0, d - (nY - 1)), min(d, nZ - 1) + 1):
        idx[(d - zb, zb)] = nxt
        nxt += 1
ts = t[:, :, 0].astype(float) / 100.0  # us
ts[t[:, :, 0] == 0] = np.nan
t0 = np.nanmin(ts)
ts -= t0
rec, prv = t[:, :, 1] / 100.0, t[:, :, 2] / 100.0
nst = np.sum(~np.isnan(ts), axis=1)
print(f"chunks {nc}, steps {ms}, sweep {np.nanmax(ts):.1f} us")
c0 = idx[(0, 0)]
d0 = np.diff(ts[c0, : nst[c0]])
print(f"first chunk: step time median {np.median(d0):.3f} us, mean {d0.mean():.3f}")
gaps = []
for (y, z), c in idx.items():
    ps = [idx[p] for p in ((y - 1, z), (y, z - 1)) if p in idx]
    if not ps:
        continue
    n = min([nst[c]] + [nst[p] for p in ps])
    s = np.arange(0, n)
    sp = np.minimum(s + lag, n - 1)
    need = np.max([ts[p, sp] for p in ps], axis=0)
    gaps.append(ts[c, :n] - need)
gaps = np.array([g[:ms - 8] for g in gaps if len(g) >= ms - 8])
print("consumer step s done - slower producer's step s + lag done (us), over chunks with two producers:")
for s in (0, 1, 2, 3, 4, 8, 16, 32, 64, 128, 200, ms - 9):
    if s < gaps.shape[1]:
        g = gaps[:, s]
        print(f"  step {s:3d}: median {np.median(g):6.2f}  p10 {np.percentile(g, 10):6.2f}  p90 {np.percentile(g, 90):6.2f}")
allg = gaps[:, 8:]
print(f"all steps >= 8: median {np.median(allg):.2f}, p10 {np.percentile(allg, 10):.2f}, p90 {np.percentile(allg, 90):.2f}")
nrec = np.sum(rec > 0, axis=1)
print(f"recover calls per chunk: median {np.median(nrec):.0f}; duration of one: median {np.median(rec[rec > 0]):.2f} us, "
      f"p90 {np.percentile(rec[rec > 0], 90):.2f}; per chunk total median {np.median(rec.sum(axis=1)):.1f} us")
print(f"wait_prev per step: median {np.median(prv[~np.isnan(ts)]):.3f} us, mean {prv[~np.isnan(ts)].mean():.3f}")
if len(sys.argv) > 5:  # one chunk's timeline
    y, z = (int(a) for a in sys.argv[5].split(","))
    c = idx[(y, z)]
    ps = [idx[p] for p in ((y - 1, z), (y, z - 1)) if p in idx]
    for s in range(0, min(nst[c], 60)):
        print(s, f"{ts[c, s]:8.2f}", " ".join(f"{ts[p, min(s + lag, nst[p] - 1)]:8.2f}" for p in ps), f"rec {rec[c, s]:.2f} prev {prv[c, s]:.2f}")
