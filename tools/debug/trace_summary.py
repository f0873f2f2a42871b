"""Fold a rocprofv3 --kernel-trace CSV: the last replay's kernels in start order with gaps, per-name totals, busy / idle time."""
import csv, sys, collections
path = sys.argv[1]
rows = list(csv.DictReader(open(path)))
ks = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows]
ks.sort()
# split into bursts separated by > 200 us of idle
bursts, cur = [], [ks[0]]
for k in ks[1:]:
    if k[0] - max(e for _, e, _ in cur) > 200_000:
        bursts.append(cur); cur = [k]
    else:
        cur.append(k)
bursts.append(cur)
last = bursts[-1]
t0 = last[0][0]; t1 = max(e for _, e, _ in last)
print(f'{len(bursts)} bursts; last: {len(last)} kernels, span {(t1 - t0) / 1e3:.1f} us, sum of durations {sum(e - s for s, e, _ in last) / 1e3:.1f} us')
# busy time (union of intervals)
busy, ce = 0, t0
for s, e, _ in last:
    if e > ce:
        busy += e - max(s, ce); ce = e
print(f'union busy {busy / 1e3:.1f} us, idle {(t1 - t0 - busy) / 1e3:.1f} us')
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in last:
    n = n.split('(')[0][:70]
    agg[n][0] += 1; agg[n][1] += e - s
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f'{d / 1e3:9.1f} us x{c:4d}  {n}')
if len(sys.argv) > 2:
    for s, e, n in last[:int(sys.argv[2])]:
        print(f'{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:7.1f}  {n.split("(")[0][:60]}')
