"""Turn the rocprofv3 outputs under gpurun_out/ (kernel stats + FETCH_SIZE / WRITE_SIZE passes) into the
summaries committed under profiles/ (run in the build container after a gpurun profiling call)."""
import collections, csv, glob, json, re, sys

def short(n): return re.sub(r'\(anonymous namespace\)::', '', n)

tag = sys.argv[1] if len(sys.argv) > 1 else 'round1'
rows = list(csv.DictReader(open(glob.glob('gpurun_out/r1prof/runc*kernel_stats.csv')[0])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
with open('profiles/%s_kernel_stats.csv' % tag, 'w') as f:
    f.write('# rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline   (MI355X, batch 32, 3 covariates;\n')
    f.write('# 10 timed steps replayed from the hipGraph + warm-up/capture + one eager pass for the per-kernel HIP events)\n')
    f.write('Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n')
    for r in rows[:70]:
        f.write('"%s",%s,%s,%s,%.2f,%s,%s\n' % (short(r['Name'])[:120], r['Calls'], r['TotalDurationNs'], r['AverageNs'],
                                               100 * float(r['TotalDurationNs']) / tot, r['MinNs'], r['MaxNs']))
# per-dispatch traffic, grouped by (kernel, grid size) so that layers sharing a kernel instance stay apart
traffic = collections.defaultdict(dict)
for name, ctr in (('r1fetch', 'FETCH_SIZE'), ('r1write', 'WRITE_SIZE')):
    for r in csv.DictReader(open(glob.glob('gpurun_out/%s/runc*counter_collection.csv' % name)[0])):
        if r['Counter_Name'] == ctr:
            traffic[(short(r['Kernel_Name'])[:120], r['Grid_Size'])].setdefault(ctr, []).append(float(r['Counter_Value']))
out = {}
for (k, grid), v in traffic.items():
    if 'FETCH_SIZE' in v and 'WRITE_SIZE' in v:
        f_ = sum(v['FETCH_SIZE']) / len(v['FETCH_SIZE']) * 1024; w_ = sum(v['WRITE_SIZE']) / len(v['WRITE_SIZE']) * 1024
        # MI355X_MICROARCH.md (HBM): counters are in KB; on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced
        # read stream (LDS-DMA / dwordx4); WRITE_SIZE is exact.  Both the raw and the doubled figure are kept.
        out['%s | grid %s' % (k, grid)] = {'fetch_bytes_raw': f_, 'fetch_bytes_x2': 2 * f_, 'write_bytes': w_,
                                           'hbm_bytes_per_launch': 2 * f_ + w_, 'launches': len(v['FETCH_SIZE'])}
json.dump(out, open('profiles/%s_hbm_traffic.json' % tag, 'w'), indent=1, sort_keys=True)
for k in sorted(out, key=lambda k: -out[k]['hbm_bytes_per_launch'])[:10]:
    print('%-95s fetch(raw) %7.1f MB write %7.1f MB' % (k[:95], out[k]['fetch_bytes_raw'] / 1e6, out[k]['write_bytes'] / 1e6))
tr = list(csv.DictReader(open(glob.glob('gpurun_out/r1prof/runc*kernel_trace.csv')[0])))
print('dispatches in trace', len(tr))

# traffic keyed the way bench.py names its kernels (entry point : layer / direction), for roofline.traffic
TABLE = {'vg_wgrad3d:convt5/bwd': 'wgrad_rows_k<1, 2, 3, 3, 3, 1, false, false', 'vg_wgrad3d:convt4/bwd': 'wgrad_rows_k<8, 3, 5, 3, 3, 2, false, false',
         'vg_corr3d:convt4/bwd': 'corr3d_plane_k<8, 5, 3, 3, 2', 'vg_corr3d:convt5/fwd': 'corr3d_plane_k<1, 3, 3, 3, 1, 2, 2, 4',
         'vg_corr3d:convt5/bwd': 'corr3d_direct_k<8, 3, 3, 3, 1, 1, 1, 4', 'vg_corr3d:convt3/fwd': 'corr3d_plane_k<8, 3, 3, 3, 1, 1, 1, 4',
         'vg_tconv3d_s2:convt4/fwd': 'tconv3d_s2_k<8, 5, 3, 3', 'vg_tconv3d_s2_stats:convt4/fwd': 'tconv3d_s2_k<8, 5, 3, 3', 'vg_wgrad3d:convt3/bwd': 'wgrad_rows_k<8, 2, 3, 3, 3, 1, false, false'}
by_layer = {}
for key, sub in TABLE.items():
    cands = [v for k, v in out.items() if sub in k]
    if cands:
        best = max(cands, key=lambda v: v['hbm_bytes_per_launch'])      # the decoder launch (128 samples) of a shared instance
        by_layer[key] = best
json.dump(by_layer, open('profiles/%s_traffic_by_layer.json' % tag, 'w'), indent=1, sort_keys=True)
print({k: round(v['hbm_bytes_per_launch'] / 1e6, 1) for k, v in by_layer.items()})
