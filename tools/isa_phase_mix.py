import sys
lines=open('/tmp/cp.s').read().split('\n')
for name in ('ILi9ELi64ELi3ELi0','ILi9ELi128ELi3ELi0'):
    start=[i for i,l in enumerate(lines) if l.startswith('_ZN12_GLOBAL__N_114pconv_q_kernel'+name+'EEEvNS_6PConvPE:')][0]
    end=start+[i for i,l in enumerate(lines[start:]) if 's_endpgm' in l][0]
    body=lines[start:end]
    def cls(l):
        l=l.strip()
        if not l or l.startswith(('.',';')) or l.endswith(':'): return None
        op=l.split()[0]
        if op.startswith('v_mfma'): return 'mfma'
        if op.startswith('buffer_load') or op.startswith('global_load'): return 'dma'
        if op.startswith('ds_'): return 'lds'
        if op.startswith('s_waitcnt'): return 'wait'
        if op.startswith('s_load'): return 'sload'
        if op.startswith('s_'): return 'salu'
        if op.startswith('v_'): return 'valu'
        return 'other'
    bars=[i for i,l in enumerate(body) if 's_barrier' in l]
    print(name, len(body))
    for a,b in zip(bars[:-1],bars[1:]):
        h={}
        for l in body[a:b]:
            c=cls(l)
            if c: h[c]=h.get(c,0)+1
        print('  ',a,b,h)
