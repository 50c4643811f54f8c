"""Quick census of the biggest basic blocks of one kernel in a hipcc --save-temps .s file (development aid).
usage: loop_census.py file.s kernel_name"""
import collections, re, sys
s = open(sys.argv[1]).read()
i = s.index(sys.argv[2] + ':')
j = s.index('.end_amdhsa_kernel', i) if '.end_amdhsa_kernel' in s[i:] else len(s)
j = s.find('.Lfunc_end', i)
blocks, cur, name = [], [], 'entry'
for ln in s[i:j].split('\n'):
    m = re.match(r'^(\.LBB\d+_\d+):', ln)
    if m:
        blocks.append((name, cur)); name = m.group(1); cur = []
    else:
        t = ln.strip()
        if t and not t.startswith(';') and not t.startswith('.'):
            cur.append(t)
blocks.append((name, cur))
for n, b in blocks:
    if len(b) > 80:
        c = collections.Counter()
        for t in b:
            op = t.split()[0]
            k = ('mfma' if op.startswith('v_mfma') else op if op.startswith('v_accvgpr') or op.startswith('ds_') or op.startswith('global_') or op.startswith('scratch_') or op.startswith('buffer_')
                 else 'trans' if op.startswith(('v_exp', 'v_rcp')) else 'v_pk' if op.startswith('v_pk') else 'valu' if op.startswith('v_')
                 else op if op in ('s_waitcnt', 's_nop', 's_barrier') else 'salu' if op.startswith('s_') else op)
            c[k] += 1
        print(n, len(b), dict(c.most_common(30)))
