"""Fold a rocprofv3 --kernel-trace CSV of tools/debug/{step,student_fwd}_trace.py: the LAST burst of kernels (one graph replay):
span, union-busy time, per-kernel totals; optional: the first N kernels in start order."""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return re.sub(r'\(.*', '', n)[:64]
gap = int(float(sys.argv[3]) * 1000) if len(sys.argv) > 3 else 300_000
bursts, cur, ce = [], [ks[0]], ks[0][1]
for k in ks[1:]:
    if k[0] - ce > gap:
        bursts.append(cur); cur = []
    cur.append(k); ce = max(ce, k[1])
bursts.append(cur)
last = bursts[-1]
t0, t1 = last[0][0], max(e for _, e, _ in last)
busy, ce = 0, t0
for s, e, _ in last:
    if e > ce:
        busy += e - max(s, ce); ce = e
print(f'{len(bursts)} bursts; last: {len(last)} kernels, span {(t1 - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us, sum of durations {sum(e - s for s, e, _ in last) / 1e3:.1f} us')
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in last:
    agg[short(n)][0] += 1; agg[short(n)][1] += e - s
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f'{d / 1e3:9.1f} us x{c:4d}  {n}')
if len(sys.argv) > 2 and int(sys.argv[2]):
    for s, e, n in last[:int(sys.argv[2])]:
        print(f'{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:7.1f}  {short(n)}')
