"""One replayed step of a rocprofv3 --kernel-trace csv as a timeline (start us, duration us, gap in front if the GPU was idle, queue, kernel):
  python tools/diag/step_timeline.py <kernel_trace.csv> [min_ms max_ms]   (the last step whose length lies in [min_ms, max_ms], default 6..8)"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
lo, hi = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (6.0, 8.0)
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])[:70], r.get('Queue_Id', '?')) for r in rows))
marks = [i for i, e in enumerate(ev) if 'adam_advance' in e[2]]
best = None
for k in range(len(marks) - 1):
    seg = ev[marks[k] + 1:marks[k + 1] + 1]
    if lo < (max(e[1] for e in seg) - seg[0][0]) / 1e6 < hi:
        best = seg
t0 = best[0][0]; cur_end = t0
for s, e, n, q in best:
    gap = (s - cur_end) / 1e3
    print('%8.1f %7.1f %s q%s %s' % ((s - t0) / 1e3, (e - s) / 1e3, 'G%5.1f' % gap if gap > 0 else '      ', q, n))
    cur_end = max(cur_end, e)
