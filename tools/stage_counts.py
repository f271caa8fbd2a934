"""Static instruction counts per marked stage (PT_MARK) of k_bounce_q<LAST=0,GEN=0,MESH=0> in a -DPT_MARKERS -save-temps build:
   cd project2-pathtracer_amd/build/mark && hipcc --offload-arch=gfx950 <Makefile FLAGS> -DPT_MARKERS -save-temps -c ../../csrc/pt_k_queue.hip -o m.o
usage: python tools/stage_counts.py [file.s] [mangled-name substring]"""
import re, sys
path = sys.argv[1] if len(sys.argv) > 1 else "project2-pathtracer_amd/build/mark/pt_k_queue-hip-amdgcn-amd-amdhsa-gfx950.s"
key = sys.argv[2] if len(sys.argv) > 2 else "k_bounce_qILb0ELb0ELb0EEE"
s = open(path).read()
m = re.search(r'^(_ZN[^\n:]*' + re.escape(key) + r'[^\n:]*):', s, re.M)
start = m.end(); end = s.index('.Lfunc_end', start)
cur = 'prologue'; counts = {}; order = []
for l in s[start:end].split('\n'):
    t = l.strip()
    mm = re.match(r'; PTMARK (\w+)', t)
    if mm:
        cur = mm.group(1)
        continue
    if not t or t.startswith(';') or t.startswith('.') or t.endswith(':'):
        continue
    op = t.split()[0]
    if cur not in counts:
        counts[cur] = dict(valu=0, salu=0, lds=0, vmem=0, other=0); order.append(cur)
    c = counts[cur]
    if op.startswith('v_'): c['valu'] += 1
    elif op.startswith('s_'): c['salu'] += 1
    elif op.startswith('ds_'): c['lds'] += 1
    elif op.startswith(('buffer_', 'global_', 'flat_')): c['vmem'] += 1
    else: c['other'] += 1
for k in order:
    print("%-14s" % k, counts[k])
