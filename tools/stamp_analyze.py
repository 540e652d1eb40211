"""tools/stamp_analyze.py <dump.bin> idx idx ...: mean deltas between the listed stamp indices (32 u64 per wave)"""
import sys, numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 32)
idx = [int(x) for x in sys.argv[2:]]
ok = np.all(a[:, idx] != 0, axis=1)
a = a[ok].astype(np.int64)
print(f"{ok.sum()} waves with all stamps")
tot = 0
for i, j in zip(idx[:-1], idx[1:]):
    d = a[:, j] - a[:, i]
    tot += d.mean()
    print(f"  {i:2d} -> {j:2d}: mean {d.mean():9.0f}  p10 {np.percentile(d,10):9.0f}  p90 {np.percentile(d,90):9.0f}")
print(f"  sum {tot:.0f}")
