#!/usr/bin/env python3
"""Live VGPRs per basic block of one kernel in an AMDGPU assembly listing (hipcc -S): defs / uses per instruction, CFG from the
labels and branches, backward data flow; prints the maximum and every block whose peak is at least `threshold`.

    bash profiles/dev_tu_rows4.sh                      # 2-second build of k_grid_rows<4,1> alone -> /tmp/rows4_dev.s
    python3 profiles/isa_liveness.py /tmp/rows4_dev.s _Z11k_grid_rowsILi4ELi1E 110

The allocator's total (VGPRs: N) says how many registers a kernel needs, this says WHERE (round 5: the store phase of the image
variant, not the QL loop, set k_grid_rows<4,1>'s 232).  Approximate: EXEC-masked definitions are treated as kills."""
import re,sys
lines=open(sys.argv[1]).read().split('\n')
kern=sys.argv[2]
start=[i for i,l in enumerate(lines) if l.startswith(kern)][0]
end=[i for i,l in enumerate(lines) if i>start and l.strip().startswith('.Lfunc_end')][0]-1   # (a kernel may hold several s_endpgm)
ins=[]  # (lineno, text, label)
labels={}
for i in range(start+1,end+1):
    l=lines[i].split(';')[0].rstrip()
    if not l.strip(): continue
    if re.match(r'^\.LBB\S+:',l): labels[l.split(':')[0]]=len(ins); continue
    if l.strip().startswith('.') : continue
    ins.append((i,l.strip()))
def regs(tok):
    out=[]
    for a,b in re.findall(r'\bv\[(\d+):(\d+)\]',tok): out+=list(range(int(a),int(b)+1))
    tok2=re.sub(r'v\[\d+:\d+\]','',tok)
    out+=[int(x) for x in re.findall(r'\bv(\d+)\b',tok2)]
    return out
nodef=('global_store','ds_write','scratch_store','buffer_store','v_cmp','v_readfirstlane','v_readlane','s_','ds_append','global_atomic')
n=len(ins)
defs=[set() for _ in range(n)]; uses=[set() for _ in range(n)]; succ=[[] for _ in range(n)]
for k,(ln,t) in enumerate(ins):
    parts=t.split(None,1)
    op=parts[0]; ops=parts[1] if len(parts)>1 else ''
    opl=[o.strip() for o in ops.split(',')]
    if op.startswith(nodef) or op.startswith('v_cmpx'):
        for o in opl: uses[k]|=set(regs(o))
    else:
        if opl:
            d=set(regs(opl[0])); defs[k]|=d
            for o in opl[1:]: uses[k]|=set(regs(o))
            if 'fmac' in op or 'mac_' in op or op.startswith('v_dot') or 'dpp' in t or op.startswith('v_cndmask')==False and False: uses[k]|=d
    # successors
    if op=='s_branch':
        succ[k]=[labels[opl[0]]]
    elif op.startswith('s_cbranch'):
        succ[k]=[labels[opl[0]]]+([k+1] if k+1<n else [])
    elif op=='s_endpgm': succ[k]=[]
    else: succ[k]=[k+1] if k+1<n else []
livein=[set() for _ in range(n)]
changed=True
while changed:
    changed=False
    for k in range(n-1,-1,-1):
        out=set()
        for s_ in succ[k]: out|=livein[s_]
        # partial (exec-masked) defs: do not kill
        new=uses[k]|(out-defs[k]) if False else uses[k]|out- (defs[k]-uses[k])
        new=uses[k]|(out-defs[k])
        if new!=livein[k]: livein[k]=new; changed=True
cnt=[len(x) for x in livein]
mx=max(cnt); print('max live',mx)
# report per label region max
inv={v:k for k,v in labels.items()}
cur='entry'; best={}
for k in range(n):
    if k in inv: cur=inv[k]
    if cur not in best or cnt[k]>best[cur][0]: best[cur]=(cnt[k],ins[k][0])
for c,(v,ln) in best.items():
    if v>=int(sys.argv[3]) : print(c,v,'line',ln+1)
