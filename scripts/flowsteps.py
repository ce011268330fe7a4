import sys
import numpy as np
t = np.loadtxt(sys.argv[1], dtype=np.int64)
for name, a in (("chunk 5", t[:2048]), ("chunk n/2+5", t[2048:])):
    clk, polls = a[0::2], a[1::2]
    n = int((clk > 0).sum())
    d = np.diff(clk[:n]) / 100.0
    pd = np.diff(polls[:n])
    print(name, "steps", n, "total", (clk[n - 1] - clk[0]) / 100.0, "us; step time median", np.median(d), "mean", d.mean().round(2),
          "; steps with a poll", int((pd > 0).sum()))
    print("  step us:", " ".join(f"{x:.1f}{'*' if q else ''}" for x, q in zip(d[:60], pd[:60])))
    print("  ...     ", " ".join(f"{x:.1f}{'*' if q else ''}" for x, q in zip(d[120:170], pd[120:170])))
