"""Instruction histogram per kernel of a hipcc -S device listing:  python tools/isa_stats.py file.s [name-regex]
(build the listing with  hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o file.s src.hip)"""
import collections, re, subprocess, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ''
meta = {}
for m in re.finditer(r'\.set (\S+)\.num_vgpr, (\d+)', s):
    meta.setdefault(m.group(1), [0, 0])[1] = int(m.group(2))
for m in re.finditer(r'\.set (\S+)\.numbered_sgpr, (\d+)', s):
    meta.setdefault(m.group(1), [0, 0])[0] = int(m.group(2))
KEYS = ['v_fma_f32', 'v_fmac_f32', 'v_pk_fma_f32', 'v_mfma', 'v_pk_mul_f32', 'v_mul_f32', 'v_add', 'v_max_f32', 'v_cndmask_b32', 'ds_read', 'ds_write', 'global_load_lds',
        'global_load', 'global_store', 's_load', 's_waitcnt', 's_barrier', 'scratch_', 's_cbranch', 'v_readlane', 'v_mov_b32']
for m in re.finditer(r'^(_Z\S+):[^\n]*\n(.*?)^\.Lfunc_end', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if pat and not re.search(pat, name):
        continue
    ins = [l.split()[0] for l in body.split('\n') if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
    c = collections.Counter()
    for i in ins:
        for k in KEYS:
            if i.startswith(k):
                c[k] += 1; break
    try:
        dn = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', name], capture_output=True, text=True).stdout.strip()[:110]
    except Exception:
        dn = name
    sg, vg = meta.get(name, (0, 0))
    print('%s\n   insts %d  sgpr %d vgpr %d | ' % (dn, len(ins), sg, vg) + '  '.join('%s %d' % (k, c[k]) for k in KEYS if c[k]))
