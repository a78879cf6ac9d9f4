#!/usr/bin/env python3
"""bench.py -- fMRI volumes/sec through the full VAE-GAM train step on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by
torch.distributed.run with one rank per GPU.  Prints ONE JSON line on rank 0.

Workload (`config.workload`): BASELINE.json configs[2], the largest single-GPU configuration -- the full
VAE-GAM (GP regressors, HRF on `task`, GLM least-squares regulariser on) on the synthetic checker-control
set (2 subjects x 98 volumes of 41x49x35, 'Large3' control signal), 8 covariates incl. 6 motion,
batch 64 PER GPU (weak scaling), random-init weights (seed 1), fp32, data resident in HBM.
(`--batch 32 --covariates 3` runs configs[1].)
A step = forward + backward + (N>1: RCCL all-reduce of the flat gradient buffer) + fused Adam on
one minibatch.  `value` = N * batch * K / t, t = max over ranks of the barrier-bracketed wall time.

roofline: the dominant kernel's ALGORITHMIC bytes (what it must read + write once, DESIGN.md
"Kernels") / its mean duration, measured with HIP events recorded on the launching stream inside
this run, against 8 TB/s; `traffic` = the PMC-measured HBM bytes per launch of that kernel from the
committed rocprofv3 passes (profiles/), emitted only while the kernel sources are byte-identical to
the ones profiled (hash check), else null.
cpu_baseline: the CPU oracle (oracle/vaegam_oracle.py, a "port" of the reference checked against
reference goldens) timed on the host cores this process may use (count stated) on a bounded sample
of the same workload, rank 0, N=1 only -- compute-only (`value`) and as-shipped-equivalent
(`as_shipped_value`: + the per-forward image / figure logging and host copies the reference's
forward performs in training, vae_reg_GP.py:331-337,370-372,381-398, emulated by oracle/as_shipped.py).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md)
FP32_PEAK_TF = 157.3
ALG_BYTES_PER_VOL = {3: 69.6e6, 8: 139.2e6, 12: 194.8e6}
ALG_GFLOP_PER_VOL = {3: 1.600, 8: 3.241, 12: 4.555}
HIRES_ALG_BYTES_PER_VOL, HIRES_ALG_GFLOP_PER_VOL = 1739e6, 51.40   # 82x98x70, 12 covariates


def host_threads():
    """(threads, note): every CPU this process may really use -- its affinity mask, cut to the cgroup CPU quota of the box
    (threads beyond the quota only get throttled) -- and how that number came about, incl. the host's physical cores."""
    n_aff = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    n, quota = n_aff, None
    try:
        q, p = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            quota = float(q) / float(p)
            n = min(n, max(1, int(quota)))
    except Exception:
        pass
    phys = None
    try:
        cores = set(); pid = cid = None
        for line in open('/proc/cpuinfo'):
            if line.startswith('physical id'):
                pid = line.split(':')[1].strip()
            elif line.startswith('core id'):
                cid = line.split(':')[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    cores.add((pid, cid))
                pid = cid = None
        phys = len(cores) or None
    except Exception:
        pass
    note = 'host: %s physical cores, %d logical CPUs in the affinity mask, cgroup cpu quota %s -> %d threads' % (
        phys if phys else '?', n_aff, ('%.1f' % quota) if quota else 'none', max(1, n))
    return max(1, n), note


def kernel_source_sha():
    """sha256 over the kernel sources + the C-ABI header: ties a committed PMC traffic figure to the code it was measured on."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, 'vae-gam_amd', 'csrc')
    for f in sorted(os.listdir(csrc)):
        if f.endswith(('.hip', '.h')):
            h.update(f.encode()); h.update(open(os.path.join(csrc, f), 'rb').read())
    h.update(open(os.path.join(ROOT, 'include', 'vaegam.h'), 'rb').read())
    return h.hexdigest()[:16]


def alg_bytes_per_launch(key, model, B):
    """Algorithmic HBM bytes of ONE launch of a kernel: every element it must read or write, once.
    key = '<entry point>:<layer>/<fwd|bwd>' as tagged by vae_gam_amd.ops."""
    import numpy as np
    fn, tag = key.split(':')
    if '/' not in tag:
        return None
    lname, direction = tag.split('/')
    geom = model.geom
    G = model.num_covariates + 1
    if lname.startswith('convt'):
        i = int(lname[5:]) - 1; sp = geom.dec[i]; sizes = geom.dec_sizes(); N = G * B
    elif lname.startswith('conv'):
        i = int(lname[4:]) - 1; sp = geom.enc[i]; sizes = geom.enc_sizes(); N = B
    else:
        return None
    n_in = N * sp.ci * int(np.prod(sizes[i])); n_out = N * sp.co * int(np.prod(sizes[i + 1]))
    if fn in ('vg_corr3d', 'vg_tconv3d_s2', 'vg_tconv3d_s2_stats', 'vg_conv_mm'):
        has_mask = direction == 'bwd' and lname not in ('convt1', 'convt3', 'convt5', 'conv3', 'conv5')
        extra = n_in if has_mask else 0                       # data gradient of a layer fed by a plain ReLU: + the saved activation (mask)
        return 4 * (n_in + n_out + extra)
    if fn in ('vg_wgrad3d', 'vg_wgrad3d_grouped'):
        return 4 * (n_in + n_out)
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=64, help='minibatch per GPU (configs[2]: 64; configs[1]: 32)')
    ap.add_argument('--covariates', type=int, default=8, help='configs[2]: 8 (full model); configs[1]: 3')
    ap.add_argument('--subjects', type=int, default=2)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-steps', type=int, default=5, help='timed CPU-oracle steps (median; after 2 warm-ups, SURVEY 8d); ~5 s each at batch 64 / 8 covariates on 16 threads')
    ap.add_argument('--eager', action='store_true', help='launch kernels eagerly instead of replaying a captured hipGraph')
    ap.add_argument('--kernel-table', action='store_true', help='print the per-kernel HIP-event table to stderr')
    ap.add_argument('--hires', action='store_true', help="BASELINE configs[4]'s per-GPU slice: 82x98x70 volumes, 12 covariates, 64 GP inducing points "
                    '(gp_jitter 1e-4, SURVEY H2), batch 64 per GPU; --batch / --steps as given (defaults here: 64, 10 steps after 3 warm-ups)')
    a = ap.parse_args()

    import numpy as np
    import torch
    import vae_gam_amd  # noqa: F401
    from vae_gam_amd import ops, synthetic, _lib
    from vae_gam_amd.DataClass_GP import DeviceResidentData
    from vae_gam_amd.vae_reg_GP import VAE

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != a.gpus and world > 1:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: no GPU visible (the hot path has no CPU fallback)')
    local_rank = local_rank % torch.cuda.device_count()        # (tests: several ranks may share one GPU over gloo)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    _lib.get_lib()
    dp = None
    if world > 1 or os.environ.get('VG_DP_FORCE') == '1':      # VG_DP_FORCE: exercise the collectives' code path with one rank
        from vae_gam_amd import dp as dpmod
        dp = dpmod.DataParallelContext.from_env()

    if a.hires:
        a.covariates = 12
        if a.steps == 30 and a.warmup == 5:
            a.steps, a.warmup = 10, 3                                    # 64 volumes of 82x98x70 x 13 decoder passes: ~0.14 s per step
    B, C = a.batch, a.covariates
    # weak scaling: 2 synthetic subjects per GPU (configs[3]: 16 subjects on 8 GPUs), and always at least two global minibatches
    a.subjects = max(a.subjects, 2 * world)
    while not a.hires and a.subjects * 98 < 2 * B * world:
        a.subjects += 1
    img = (82, 98, 70) if a.hires else (41, 49, 35)
    vps = 98 if not a.hires else max(2 * B * world // a.subjects + 1, 16)       # hi-res: just enough volumes for two global minibatches (0.56 M floats each)
    ds = synthetic.make_dataset(num_subjects=a.subjects, vols_per_subject=vps, num_covariates=C, seed=0, img_shape=img)
    torch.manual_seed(1)                                             # CLI default seed (multsubj_reg_run_GP.py:31)
    model = VAE(num_covariates=C, glm_maps=ds['glm'], xu_ranges=ds['xu_ranges'], device_name='cuda', data_parallel=dp,
                **(dict(img_shape=img, num_inducing_pts=64, gp_jitter=1e-4) if a.hires else {}))
    data = DeviceResidentData(torch.from_numpy(ds['volumes']), torch.from_numpy(ds['covariates']),
                              torch.from_numpy(ds['subjid']), batch_size=B * world, shuffle=True, seed=0, device=dev,
                              rank=rank, world=world)
    batches = list(iter(data))                                       # index tensors resolved once; volumes stay in HBM
    torch.manual_seed(1234 + 0)                                      # identical device noise stream on every rank

    def run_steps(n, first=0):
        for s in range(first, first + n):
            smp = batches[s % len(batches)]
            model.train_step(smp['subjid'], smp['covariates'], smp['volume'])

    def barrier():
        if dp is not None:
            dp.barrier()

    # The step is captured into a hipGraph at every N.  The RCCL collectives inside the captured step were exercised on the
    # one-GPU development box with a single-rank communicator (VG_DP_FORCE=1: 8.9 ms eager -> 4.4 ms replayed); a capture
    # that fails on any rank makes ALL ranks fall back to eager launches (vae_reg_GP._capture_step).  VG_DP_GRAPH=0 forces eager.
    model.use_hip_graph = (not a.eager) and (dp is None or os.environ.get('VG_DP_GRAPH', '1') != '0')
    run_steps(a.warmup)
    graphed = bool(model._graphs) and all(v is not False for v in model._graphs.values())
    torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
    if not graphed:
        ops.PROFILE = {}                                             # HIP events on the launch stream, timed region
    t0 = time.perf_counter()
    run_steps(a.steps, a.warmup)
    torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if graphed:
        # events cannot be recorded inside a replayed graph: time the same launches, same stream, eagerly,
        # right after the timed region (the kernels and their arguments are identical)
        # The step overlaps two launch chains (parameter gradients and the gain block on a second stream): inside it a kernel's
        # elapsed time includes the time it shares the GPU.  The per-kernel pass therefore runs the same launches one after the other
        # (second stream off), so that a kernel's duration -- and the roofline fraction derived from it -- is the kernel's own;
        # `ms_per_step` / `value` above are the overlapped step as shipped.  profiles/<tag>_kernel_stats_serial.csv is rocprofv3's
        # view of the same serial launches (VG_SIDE_STREAM=0 VG_OVERLAP_GAINS=0), <tag>_kernel_stats.csv of the step as shipped.
        model.use_hip_graph = False
        ops.SIDE_STREAM = False
        model.overlap_gains = False
        ops.PROFILE = {}
        run_steps(max(3, min(a.steps, 10)), a.warmup + a.steps)
        torch.cuda.synchronize()
    prof, ops.PROFILE = ops.PROFILE, None
    prof_steps = a.steps if not graphed else max(3, min(a.steps, 10))
    if dp is not None:
        dt = dp.max_scalar(dt)
    ms_per_step = 1e3 * dt / a.steps
    value = world * B * a.steps / dt

    # ---- per-kernel table from the events of the timed region
    rows = []
    for key, evs in prof.items():
        tot = sum(e0.elapsed_time(e1) for e0, e1 in evs)             # ms
        rows.append((tot, key, len(evs)))
    rows.sort(reverse=True)
    roofline = None
    # HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/, tools/profile_summary.py): used only while the
    # kernel sources hash to what was profiled and the workload is the profiled one -- a stale figure is dropped, not shown
    traffic_tab, traffic_ok = {}, False
    try:
        cands = sorted(f for f in os.listdir(os.path.join(ROOT, 'profiles')) if f.endswith('_traffic_by_layer.json'))
        traffic_tab = json.load(open(os.path.join(ROOT, 'profiles', cands[-1])))
        meta = traffic_tab.get('_meta', {})
        traffic_ok = (meta.get('kernel_source_sha') == kernel_source_sha() and meta.get('batch') == B and meta.get('covariates') == C and not a.hires)
    except Exception:
        pass
    for tot, key, n in rows:
        ab = alg_bytes_per_launch(key, model, B)
        if ab is None:
            continue
        per = tot / n                                                # ms per launch
        ach = ab / (per * 1e-3) / 1e9
        roofline = {'bound': 'hbm', 'kernel': key, 'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': round(ach / HBM_PEAK_GBS, 4),
                    'traffic': (traffic_tab.get(key) or {}).get('hbm_bytes_per_launch') if traffic_ok else None,
                    'alg_bytes_per_launch': ab,
                    'avg_launch_us': round(per * 1e3, 2), 'launches': n,
                    'timed': 'launches serialised (second stream off) in the per-kernel pass' if graphed else 'inside the timed region (overlapped chains)',
                    'share_of_kernel_time': round(tot / max(sum(r[0] for r in rows), 1e-9), 3)}
        break
    if a.kernel_table and rank == 0:
        tsum = sum(r[0] for r in rows)
        print('%-40s %8s %10s %7s' % ('kernel:layer', 'calls', 'ms/step', 'share'), file=sys.stderr)
        for tot, key, n in rows:
            print('%-40s %8d %10.4f %6.1f%%' % (key, n, tot / prof_steps, 100 * tot / tsum), file=sys.stderr)
        print('sum of HIP kernel time %.3f ms/step (events), wall %.3f ms/step (%s)' % (tsum / prof_steps, ms_per_step, 'hipGraph replay' if graphed else 'eager'), file=sys.stderr)

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and not a.hires:     # (hi-res: one CPU step is minutes; the 41x49x35 line carries the CPU baseline)
        sys.path.insert(0, os.path.join(ROOT, 'oracle'))
        import bridge
        import vaegam_oracle as O
        import as_shipped
        nthr, thr_note = host_threads()
        print('[cpu_baseline] ' + thr_note, file=sys.stderr)
        torch.set_num_threads(nthr)
        cfg = bridge.oracle_config(model, glm_cdist=True)            # torch.cdist as the reference (vae_reg_GP.py:388)
        params = bridge.params_from_model(model)
        opt = O.AdamState(lr=cfg.lr)
        glm = torch.from_numpy(ds['glm'])
        smp = batches[0]
        xc, cc = smp['volume'].cpu(), smp['covariates'].cpu()
        gen = torch.Generator().manual_seed(0)
        for _ in range(2):                                           # warm-ups (median of >= 5 after 2)
            O.train_step(params, opt, cfg, xc, cc, glm, O.draw_noise(B, cfg, gen))
        ts, out_last = [], None
        for _ in range(max(1, a.cpu_steps)):
            t1 = time.perf_counter()
            out_last, _g = O.train_step(params, opt, cfg, xc, cc, glm, O.draw_noise(B, cfg, gen))
            ts.append(time.perf_counter() - t1)
        med = sorted(ts)[len(ts) // 2]
        # as-shipped-equivalent: the reference's forward, in training, also copies (C+2) maps of B x V floats to host arrays and
        # logs 9 x B image slices + C gain figures per call; that host work is timed once on this step's own outputs
        t_log = as_shipped.time_forward_logging(out_last, cc, cfg)
        cpu = {'value': round(B / med, 2), 'unit': 'volumes/s', 'cores': nthr, 'kind': 'port',
               'as_shipped_value': round(B / (med + t_log), 2),
               'sample': '%d train step(s) of batch %d, %d covariates after 2 warm-ups (same synthetic minibatch), median, PyTorch CPU '
                         'fp32, compute only; as_shipped_value adds the %.2f s of per-forward logging + host copies the reference '
                         'performs in training (emulated once on the same outputs); %s' % (len(ts), B, C, t_log, thr_note)}

    if rank == 0:
        out = {
            'metric': 'fMRI volumes/sec/train-step (%s)' % ('82x98x70' if a.hires else '41x49x35'), 'value': round(value, 1), 'unit': 'volumes/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(ms_per_step, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': (("configs[4] per-GPU slice (high-res stress: 64 GP inducing points, gp_jitter 1e-4)" if a.hires else 'configs[1]' if (B, C) == (32, 3)
                                     else 'configs[2] full VAE-GAM' if (B, C) == (64, 8) else 'custom') +
                                    ': B=%d/GPU C=%d, %s, %d subj x %d vol synthetic checker, fwd+bwd+Adam, %s'
                                    % (B, C, 'x'.join(str(v) for v in img), a.subjects, vps, 'hipGraph replay' if graphed else 'eager launches')),
                       'detail': 'BASELINE.json configs[%s]: synthetic checker control (Large3 glyph, block design), GP regressors on the '
                                 'continuous covariates, HRF on task, GLM regulariser on; gain / GP algebra on device in fp64'
                                 % ('4 (its 1/8 share)' if a.hires else '2' if (B, C) == (64, 8) else '1' if (B, C) == (32, 3) else '-'),
                       'global_batch': B * world, 'covariates': C, 'parallelism': 'dp%d' % world,
                       **({'dp': ('batch-norm statistics and loss normalisation over the global minibatch (all-reduced), one gradient all-reduce; dp_gain=%s: ' % model.dp_gain) +
                                ('gains drawn per rank from its own slice (block-diagonal approximation of the joint B x B gain draw; the HRF runs along the global batch '
                                 'across ranks) -- NOT the 1-rank global-batch computation, which the default dp_gain=global reproduces'
                                 if model.dp_gain == 'local' else 'joint gain draw of the global minibatch on every rank = the 1-rank global-batch step')} if world > 1 else {})},
            'roofline': roofline,
            'step_roofline': {'hbm_frac': round(value / world * (HIRES_ALG_BYTES_PER_VOL if a.hires else ALG_BYTES_PER_VOL.get(C, 0)) / (HBM_PEAK_GBS * 1e9), 4),
                              'fp32_frac': round(value / world * (HIRES_ALG_GFLOP_PER_VOL if a.hires else ALG_GFLOP_PER_VOL.get(C, 0)) / (FP32_PEAK_TF * 1e3), 4),
                              'alg_bytes_per_volume': HIRES_ALG_BYTES_PER_VOL if a.hires else ALG_BYTES_PER_VOL.get(C),
                              'gflop_per_volume': HIRES_ALG_GFLOP_PER_VOL if a.hires else ALG_GFLOP_PER_VOL.get(C)},
            'cpu_baseline': cpu,
        }
        print(json.dumps(out), flush=True)
    if dp is not None:
        dp.shutdown()


if __name__ == '__main__':
    main()
