import sys, numpy as np
t = np.fromfile(sys.argv[1], dtype=np.int64).reshape(-1, 4)
n = int(sys.argv[2]); P = int(sys.argv[3])
start, end, spins, first = t[:,0], t[:,1], t[:,2], t[:,3]
t0 = start.min()
dur = (end - start) / 100.0   # us (100 MHz)
S = len(t)
print("strips", S, "kernel span us", (end.max() - t0) / 100.0)
print("strip duration us: first round mean %.1f, last round mean %.1f" % (dur[:P].mean(), dur[-min(P,S)//2:].mean()))
print("ns per step (strip duration / n): first round %.1f  later %.1f" % (dur[:P].mean()*1000/n, dur[P:].mean()*1000/n if S > P else 0))
ds = np.diff(start[:P]) / 100.0
print("start-to-start delay between consecutive strips in round 1 (us): mean %.2f median %.2f  => total fill %.1f ms" % (ds.mean(), np.median(ds), (start[min(P,S)-1]-t0)/100.0/1000))
de = np.diff(end) / 100.0
print("end-to-end delay (us): mean %.2f" % de.mean())
print("poll spins per strip: mean %.0f (per chunk %.2f)" % (spins.mean(), spins.mean() / (n / 64)))
print("first 2 chunks time us (mean)", ((first - start)/100.0)[:P].mean())
