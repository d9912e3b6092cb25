"""Summarise the feat_stream timing probe (tools/stream_dbg.sh): per wave, mean cycles of the compute phase (step start ->
first barrier), the wait in the first barrier, the write phase and the wait in the second barrier."""
import sys
import numpy as np

rows = np.loadtxt(sys.argv[1], dtype=np.int64)
waves = sorted(set(rows[:, 0]))
print("wave role   compute  wait1  write  wait2   step")
for w in waves:
    r = rows[rows[:, 0] == w]
    r = r[(r[:, 3] > 0)]
    if len(r) < 3:
        continue
    t0, t1, t2, t3 = r[:, 3], r[:, 4], r[:, 5], r[:, 6]
    step = np.diff(t0)
    print("%4d %4d  %8.0f %6.0f %6.0f %6.0f %6.0f" % (w, r[0, 1], np.mean(t1 - t0), np.mean(t2 - t1), np.mean(t3 - t2),
                                                  np.mean(t0[1:] - t3[:-1]), np.mean(step)))
