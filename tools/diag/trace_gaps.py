"""Idle time and overlap inside one replayed step from a rocprofv3 --kernel-trace csv:  python tools/diag/trace_gaps.py <kernel_trace.csv> [step_ms]
Takes the LAST complete step (the kernels between two consecutive adam_advance launches), prints busy / idle of the GPU as a whole,
the time two kernels overlap, the number of launches and the largest gaps with the kernels around them."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])[:60]) for r in rows), key=lambda e: e[0])
marks = [i for i, e in enumerate(ev) if 'adam_advance' in e[2]]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -1
for k in range(len(marks) - 1):
    seg = ev[marks[k] + 1:marks[k + 1] + 1]
    print('  step %d: %d launches, %.3f ms' % (k, len(seg), (max(e[1] for e in seg) - seg[0][0]) / 1e6))
a, b = (marks[which - 1], marks[which]) if which < 0 else (marks[which], marks[which + 1])
st = ev[a + 1:b + 1]
t0, t1 = st[0][0], max(e[1] for e in st)
print('step: %d launches, %.3f ms from first start to last end' % (len(st), (t1 - t0) / 1e6))
# union of busy intervals
pts = sorted([(s, 1) for s, e, _ in st] + [(e, -1) for s, e, _ in st])
busy = over = 0; depth = 0; last = t0
for t, dlt in pts:
    if depth >= 1: busy += t - last
    if depth >= 2: over += t - last
    depth += dlt; last = t
print('GPU busy %.3f ms, idle %.3f ms, >=2 kernels running %.3f ms; sum of kernel durations %.3f ms' % (busy / 1e6, (t1 - t0 - busy) / 1e6, over / 1e6, sum(e - s for s, e, _ in st) / 1e6))
# gaps between consecutive busy intervals
gaps = []
cur_end = st[0][1]; prev = st[0][2]
for s, e, n in st[1:]:
    if s > cur_end:
        gaps.append((s - cur_end, prev, n))
    if e > cur_end:
        cur_end = e; prev = n
gaps.sort(reverse=True)
print('gaps: %d, total %.3f ms; > 5 us: %d (%.3f ms)' % (len(gaps), sum(g[0] for g in gaps) / 1e6, sum(1 for g in gaps if g[0] > 5000), sum(g[0] for g in gaps if g[0] > 5000) / 1e6))
for g in gaps[:12]:
    print('  %7.1f us  after %-50s before %s' % (g[0] / 1e3, g[1][:50], g[2][:50]))
short = [(e - s, n) for s, e, n in st if e - s < 10000]
print('kernels shorter than 10 us: %d, %.3f ms in total' % (len(short), sum(d for d, _ in short) / 1e6))
