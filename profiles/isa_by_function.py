#!/usr/bin/env python3
"""Static instruction counts of one kernel by the source FUNCTION each instruction was generated from (inlined callees keep their
own file and line in hipcc's line tables), and the kernel's loops with their sizes.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -gline-tables-only -S --cuda-device-only \
          -o /tmp/fhx_k2_g.s fithic_amd/csrc/fhx_k2.hip
    python profiles/isa_by_function.py /tmp/fhx_k2_g.s <mangled kernel name>

Companion of isa_budget.py (which cuts k2_classify by hand-named source regions)."""
import re, collections, sys
S=open(sys.argv[1]).read().split('\n')
name=sys.argv[2]
start=[i for i,l in enumerate(S) if l.startswith(name+':')][0]
end=[i for i in range(start,len(S)) if S[i].strip().startswith('.Lfunc_end')][0]
files={}
for l in S:
    m=re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?',l)
    if m: files[int(m.group(1))]=(m.group(3) or m.group(2))
cur=('?',0)
cnt=collections.Counter(); cntv=collections.Counter()
for l in S[start:end]:
    t=l.strip()
    m=re.match(r'\.loc\s+(\d+)\s+(\d+)',t)
    if m: cur=(files.get(int(m.group(1)),'?').split('/')[-1],int(m.group(2))); continue
    if not t or t.startswith('.') or t.startswith(';') or t.endswith(':'): continue
    op=t.split()[0]
    cnt[cur]+=1
    if op.startswith('v_'): cntv[cur]+=1
def funcs(path):
    L=open(path).read().split('\n'); out=[]
    for i,l in enumerate(L):
        m=re.match(r'^\s*(?:template.*>\s*)?(?:__device__|static|inline|__global__|__host__).*?\b(\w+)\s*\(',l)
        if m and not l.strip().endswith(';'): out.append((i+1,m.group(1)))
    return out
F={'fhx_bdtrc.hpp':funcs('fithic_amd/csrc/fhx_bdtrc.hpp'),'fhx_k2.hip':funcs('fithic_amd/csrc/fhx_k2.hip')}
agg=collections.Counter(); aggv=collections.Counter()
for (f,ln),c in cnt.items():
    fn='?'
    for a,n in F.get(f,[]):
        if a<=ln: fn=n
    agg[(f,fn)]+=c; aggv[(f,fn)]+=cntv[(f,ln)]
print('total static', sum(agg.values()), 'of which VALU', sum(aggv.values()))
print('%-22s %-28s %8s %8s' % ('file', 'function', 'static', 'VALU'))
for k, c in agg.most_common(40):
    print('%-22s %-28s %8d %8d' % (k[0], k[1], c, aggv[k]))
body = S[start:end]
labels = {}
for i, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        labels[m.group(1)] = i
def nvalu(a, b):
    v = s_ = 0
    for l in body[a:b]:
        t = l.strip()
        if not t or t.startswith('.') or t.startswith(';') or t.endswith(':'):
            continue
        op = t.split()[0]
        v += op.startswith('v_')
        s_ += op.startswith('s_')
    return v, s_
print()
print('loops (backward branches): label, VALU, SALU inside')
for i, l in enumerate(body):
    m = re.match(r'\s*s_cbranch_\w+\s+(\.LBB\d+_\d+)', l) or re.match(r'\s*s_branch\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        v, s_ = nvalu(labels[m.group(1)], i)
        if v >= 20:
            print('  %-12s %5d %5d' % (m.group(1), v, s_))
