"""Instruction mix of every loop of one kernel in a hipcc -S --offload-device-only listing:  python tools/diag/isa_loops.py <file.s> <mangled-name substring>
(loops = backward branches; per loop: instructions, MFMA / other vector / scalar / LDS / global counts)"""
import re, sys, collections
L = open(sys.argv[1]).read().split('\n')
start = next(i for i, l in enumerate(L) if l.startswith('_ZN') and sys.argv[2] in l and ':' in l)
end = next(i for i in range(start, len(L)) if L[i].strip().startswith('s_endpgm'))
L = L[start:end + 1]
lab = {}
for i, l in enumerate(L):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: lab[m.group(1)] = i
loops = []
for i, l in enumerate(L):
    m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in lab and lab[m.group(1)] < i:
        loops.append((lab[m.group(1)], i))
print('%d lines, %d loops' % (len(L), len(loops)))
for a, b in sorted(set(loops), key=lambda x: x[1] - x[0]):
    body = [l.split()[0] for l in L[a:b + 1] if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
    c = collections.Counter(body)
    nm = sum(v for k, v in c.items() if k.startswith('v_mfma'))
    fma = sum(v for k, v in c.items() if k.startswith(('v_fma', 'v_pk_fma', 'v_fmac', 'v_mac')))
    valu = sum(v for k, v in c.items() if k.startswith('v_') and not k.startswith('v_mfma'))
    print('lines %5d-%5d: %5d instr  mfma %4d  fma %4d  valu %5d  salu %4d  lds %4d  vmem %3d  waits %3d' % (
        a, b, len(body), nm, fma, valu, sum(v for k, v in c.items() if k.startswith('s_') and k != 's_waitcnt' and k != 's_nop'),
        sum(v for k, v in c.items() if k.startswith('ds_')), sum(v for k, v in c.items() if k.startswith(('global_', 'buffer_', 'flat_'))), c['s_waitcnt']))
