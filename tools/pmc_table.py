"""Per-kernel averages of the counters in one rocprofv3 --pmc csv (gpurun_out/<tag>/**/run_counter_collection.csv)."""
import collections, csv, glob, re, sys
d = sys.argv[1]
f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
if not f:
    print('no counter csv under', d); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])[:70] + ' g' + r['Grid_Size']
    acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
pat = sys.argv[2] if len(sys.argv) > 2 else ''
for k in sorted(acc):
    if pat and not re.search(pat, k):
        continue
    n = max(len(v) for v in acc[k].values())
    print('%-90s n=%d  ' % (k, n) + '  '.join('%s=%.4g' % (c, sum(v) / len(v)) for c, v in sorted(acc[k].items())))
