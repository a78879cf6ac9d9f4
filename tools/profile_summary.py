"""Turn the rocprofv3 outputs under gpurun_out/ into the summaries committed under profiles/ (run in the build container after
tools/profile_run.sh ran on the GPU box):
  profiles/<tag>_kernel_stats.csv        rocprofv3 --kernel-trace --stats of `python bench.py` (configs[2])
  profiles/<tag>_hbm_traffic.json        FETCH_SIZE / WRITE_SIZE per kernel instance (separate --pmc passes)
  profiles/<tag>_traffic_by_layer.json   the same keyed the way bench.py names its kernels, + _meta {kernel_source_sha, batch, covariates}
  profiles/<tag>_mfma_util.json          per kernel: MFMA busy fraction, VALU / SALU / LDS instructions per MFMA (one --pmc pass)
  profiles/<tag>_fetch_calibration.json  FETCH_SIZE of a known 1 GiB read for the three read shapes (tools/micro/fetch_calib.hip)
usage: python tools/profile_summary.py <tag> <batch> <covariates>"""
import collections, csv, glob, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def short(n):
    return re.sub(r'\(anonymous namespace\)::', '', n)


def one(pattern):
    f = glob.glob(os.path.join(ROOT, pattern), recursive=True)
    return f[0] if f else None


tag = sys.argv[1] if len(sys.argv) > 1 else 'round2'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
C = int(sys.argv[3]) if len(sys.argv) > 3 else 8
out_dir = os.path.join(ROOT, 'profiles')

# ---- kernel stats: the step as shipped (two overlapping launch chains) and the same launches serialised (a kernel's own duration)
for sub, outname, how in (('stats_serial', '%s_kernel_stats_serial.csv', 'VG_SIDE_STREAM=0 VG_OVERLAP_GAINS=0 (every launch on one stream: a kernel\'s own duration) '),
                          ('stats', '%s_kernel_stats.csv', '')):
  f = one('gpurun_out/%s_%s/**/*kernel_stats.csv' % (tag, sub))
  if f:
      rows = list(csv.DictReader(open(f)))
      tot = sum(float(r['TotalDurationNs']) for r in rows)
      with open(os.path.join(out_dir, outname % tag), 'w') as o:
          o.write('# %srocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline   (MI355X, BASELINE configs[2]: batch %d, %d covariates;\n' % (how, B, C))
          o.write('# 10 timed steps replayed from the hipGraph + warm-up/capture + one eager pass for the per-kernel HIP events)\n')
          o.write('Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n')
          for r in rows[:90]:
              o.write('"%s",%s,%s,%s,%.2f,%s,%s\n' % (short(r['Name'])[:140], r['Calls'], r['TotalDurationNs'], r['AverageNs'],
                                                      100 * float(r['TotalDurationNs']) / tot, r['MinNs'], r['MaxNs']))
      ours = sum(float(r['TotalDurationNs']) for r in rows if re.search(r'_k<|_k\(|_k$|adam|gather_f32|pack_weights|gain_|chol', short(r['Name'])))
      print('kernel time: %.1f ms total, %.1f %% in this library\'s kernels, %d distinct kernels, %d launches' %
            (tot / 1e6, 100 * ours / tot, len(rows), sum(int(r['Calls']) for r in rows)))

# ---- per-dispatch traffic, grouped by (kernel, grid size) so that layers sharing a kernel instance stay apart
traffic = collections.defaultdict(dict)
for sub, ctr in (('fetch', 'FETCH_SIZE'), ('write', 'WRITE_SIZE')):
    f = one('gpurun_out/%s_%s/**/*counter_collection.csv' % (tag, sub))
    if not f:
        continue
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == ctr:
            traffic[(short(r['Kernel_Name'])[:140], r['Grid_Size'])].setdefault(ctr, []).append(float(r['Counter_Value']))
out = {}
for (k, grid), v in traffic.items():
    if 'FETCH_SIZE' in v and 'WRITE_SIZE' in v:
        f_ = sum(v['FETCH_SIZE']) / len(v['FETCH_SIZE']) * 1024; w_ = sum(v['WRITE_SIZE']) / len(v['WRITE_SIZE']) * 1024
        # MI355X_MICROARCH.md (HBM): counters are in KB; on gfx950 FETCH_SIZE reports half the bytes of a coalesced read stream (measured for the
        # three read shapes used here: <tag>_fetch_calibration.json); WRITE_SIZE is exact.  Both the raw and the doubled figure are kept.
        out['%s | grid %s' % (k, grid)] = {'fetch_bytes_raw': f_, 'fetch_bytes_x2': 2 * f_, 'write_bytes': w_,
                                           'hbm_bytes_per_launch': 2 * f_ + w_, 'launches': len(v['FETCH_SIZE'])}
if out:
    json.dump(out, open(os.path.join(out_dir, '%s_hbm_traffic.json' % tag), 'w'), indent=1, sort_keys=True)
    # keyed the way bench.py names its kernels (entry point : layer / direction): the decoder launch (largest traffic) of each instance
    TABLE = {'vg_tconv3d_s2_stats:convt2/fwd': r'tconv3d_s2_k<8, 3, 3, 3', 'vg_conv_mm:convt4/fwd': r'conv_mm_k<8, 4, 4, 3, 2, 2, 1',
             'vg_conv_mm:convt3/fwd': r'conv_mm_k<8, 1, 3, 9', 'vg_conv_mm:convt3/bwd': r'conv_mm_k<8, 1, 3, 7, 0, 0, 0, true',
             'vg_corr3d:convt4/bwd': r'corr3d_plane_k<8, 5, 3, 3, 2', 'vg_corr3d:convt5/fwd': r'corr3d_plane_k<1, 3, 3, 3, 1, 4, 2, 4',
             'vg_wgrad3d_grouped:convt5/bwd': r'wgrad_rows_k<1, 2, 3, 3, 3, 1', 'vg_wgrad3d:convt4/bwd': r'wgrad_rows_k<8, 2, 5, 3, 3, 2',
             'vg_wgrad3d:convt3/bwd': r'wgrad_rows_k<8, 2, 3, 3, 3, 1', 'vg_wgrad3d:convt2/bwd': r'wgrad_rows_k<16, 2, 3, 3, 3, 2, true',
             'vg_bn_bwd_apply_tconv1:convt5/bwd': r'bn_tconv1_k<1, 8>'}
    by_layer = {}
    for key, sub in TABLE.items():
        cands = [v for k, v in out.items() if sub in k]
        if cands:
            by_layer[key] = max(cands, key=lambda v: v['hbm_bytes_per_launch'])
    import bench
    by_layer['_meta'] = {'kernel_source_sha': bench.kernel_source_sha(), 'batch': B, 'covariates': C,
                         'note': 'FETCH_SIZE x2 + WRITE_SIZE per launch, separate rocprofv3 --pmc passes of bench.py'}
    json.dump(by_layer, open(os.path.join(out_dir, '%s_traffic_by_layer.json' % tag), 'w'), indent=1, sort_keys=True)
    for k in sorted(by_layer, key=lambda k: -(by_layer[k].get('hbm_bytes_per_launch', 0) if k != '_meta' else 0)):
        if k != '_meta':
            print('%-40s fetch(raw) %8.1f MB  write %8.1f MB' % (k, by_layer[k]['fetch_bytes_raw'] / 1e6, by_layer[k]['write_bytes'] / 1e6))

# ---- matrix-core utilisation
f = one('gpurun_out/%s_mfma/**/*counter_collection.csv' % tag)
if f:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[(short(r['Kernel_Name'])[:140], r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
    util = {}
    for (k, grid), c in acc.items():
        m = sum(c.get('SQ_INSTS_MFMA', [0])) / max(len(c.get('SQ_INSTS_MFMA', [1])), 1)
        if m <= 0:
            continue
        avg = lambda n: sum(c[n]) / len(c[n]) if n in c else None
        busy, cu = avg('SQ_VALU_MFMA_BUSY_CYCLES'), avg('SQ_BUSY_CU_CYCLES')
        util['%s | grid %s' % (k, grid)] = {
            'mfma_insts': m, 'valu_per_mfma': (avg('SQ_INSTS_VALU') or 0) / m, 'salu_per_mfma': (avg('SQ_INSTS_SALU') or 0) / m,
            'lds_per_mfma': (avg('SQ_INSTS_LDS') or 0) / m if 'SQ_INSTS_LDS' in c else None,
            # SQ_VALU_MFMA_BUSY_CYCLES: cycles a SIMD's matrix pipe is busy, summed over the SIMDs (= 32 x the number of v_mfma_f32_16x16x4_f32);
            # SQ_BUSY_CU_CYCLES: cycles a CU is busy, summed over the CUs; 4 SIMDs per CU.  (Cross-check against the kernel's wall time:
            # insts * 32 / (1024 SIMDs * duration * 2.4 GHz) agrees within 15 %.)
            'mfma_busy_fraction': busy / (4.0 * cu) if busy and cu else None, 'launches': len(c['SQ_INSTS_MFMA'])}
    json.dump(util, open(os.path.join(out_dir, '%s_mfma_util.json' % tag), 'w'), indent=1, sort_keys=True)
    for k in sorted(util, key=lambda k: -util[k]['mfma_insts'])[:12]:
        u = util[k]
        print('%-90s MFMA busy %s  VALU/MFMA %.1f  SALU/MFMA %.1f' % (k[:90], ('%.2f' % u['mfma_busy_fraction']) if u['mfma_busy_fraction'] else '-', u['valu_per_mfma'], u['salu_per_mfma']))

# ---- FETCH_SIZE calibration
f = one('gpurun_out/%s_calib/**/*counter_collection.csv' % tag)
if f:
    cal = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == 'FETCH_SIZE':
            cal[short(r['Kernel_Name']).split('(')[0]].append(float(r['Counter_Value']) * 1024)
    res = {k: {'bytes_read': 2 ** 30, 'fetch_size_bytes': sum(v) / len(v), 'factor': (2 ** 30) / (sum(v) / len(v))} for k, v in cal.items()}
    json.dump(res, open(os.path.join(out_dir, '%s_fetch_calibration.json' % tag), 'w'), indent=1, sort_keys=True)
    print({k: round(v['factor'], 3) for k, v in res.items()})
