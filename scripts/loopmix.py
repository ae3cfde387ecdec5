"""instruction mix between consecutive s_barrier instructions around MFMA clusters of one kernel in a .s file"""
import re, sys, collections
s=open(sys.argv[1]).read(); pat=sys.argv[2]
starts=[(m.start(), m.group(1)) for m in re.finditer(r'\n(_ZN[A-Za-z0-9_]+):', s)]
for i,(pos,name) in enumerate(starts):
    if pat not in name: continue
    end=starts[i+1][0] if i+1<len(starts) else len(s)
    body=[l.strip() for l in s[pos:end].split('\n')]
    body=[l for l in body if l and not l.startswith((';','.'))]
    print(name[:70])
    # segments split at s_barrier / labels with mfma
    seg=[]; cur=[]
    for t in body:
        cur.append(t)
        if t.startswith('s_barrier') or t.startswith('s_endpgm'):
            seg.append(cur); cur=[]
    seg.append(cur)
    for sg in seg:
        c=collections.Counter()
        for t in sg:
            if t.endswith(':'): c['labels']+=1; continue
            op=t.split()[0]
            if op.startswith('v_mfma'): c['mfma']+=1
            elif op.startswith(('v_readlane','v_writelane')): c['lane']+=1
            elif op.startswith('v_'): c['valu']+=1
            elif op.startswith('s_waitcnt'): c['wait']+=1
            elif op.startswith(('s_cbranch','s_branch')): c['br']+=1
            elif op.startswith('s_'): c['salu']+=1
            elif op.startswith('ds_'): c['ds']+=1
            elif op.startswith(('global_','buffer_','scratch_')): c['vmem']+=1
        if c['mfma']: print('  ', dict(c))
