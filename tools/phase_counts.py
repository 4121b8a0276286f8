"""Static VALU instruction counts of the phases of k_georef_rows' row step (tools/phase_counts.hip: every phase a kernel of
its own), from the gfx950 ISA hipcc emits:  python tools/phase_counts.py   (needs only hipcc, no GPU)."""
import os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = tempfile.mkdtemp(prefix='amt_phase_')
src = open(os.path.join(root, 'tools', 'phase_counts.hip')).read().replace('/root/repo', root)
open(os.path.join(tmp, 'phase_counts.hip'), 'w').write(src)
subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '--offload-arch=gfx950', '-std=c++17', '-save-temps', '-c', 'phase_counts.hip', '-o', 'pc.o'],
                      cwd=tmp, stderr=subprocess.DEVNULL)
s = open(os.path.join(tmp, 'phase_counts-hip-amdgcn-amd-amdhsa-gfx950.s')).read()
out = {}
for m in re.finditer(r'^(ph_\w+):[^\n]*\n(.*?)s_endpgm', s, re.S | re.M):
    ins = [l.split()[0] for l in m.group(2).split('\n') if re.match(r'^\t[a-z]', l)]
    cat = dict(f64=0, trans=0, mov=0, sel=0, cmp=0, other=0)
    for i in ins:
        if not i.startswith('v_'):
            continue
        if re.match(r'v_(fma|fmac|mul|add|min|max)_f64', i): cat['f64'] += 1
        elif re.match(r'v_(rcp|rsq|sqrt)_f64', i): cat['trans'] += 1
        elif i.startswith('v_mov'): cat['mov'] += 1
        elif i.startswith('v_cndmask'): cat['sel'] += 1
        elif i.startswith('v_cmp'): cat['cmp'] += 1
        else: cat['other'] += 1
    out[m.group(1)] = cat
base = out['ph_empty']
print('%-20s %5s %5s %5s %5s %5s %5s   (minus the load / store scaffold of ph_empty)' % ('phase', 'f64', 'trans', 'mov', 'sel', 'cmp', 'other'))
for k, c in out.items():
    if k == 'ph_empty':
        continue
    print('%-20s %5d %5d %5d %5d %5d %5d' % (k[3:], c['f64'], c['trans'], max(0, c['mov'] - base['mov']), c['sel'], c['cmp'],
                                             max(0, c['other'] - base['other'])))
