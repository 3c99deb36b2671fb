import re,sys,collections
f=sys.argv[1]; pat=sys.argv[2]
lines=open(f).read().split('\n')
start=None
for i,l in enumerate(lines):
    if re.match(r'^_Z[A-Za-z0-9_]+:',l) and pat in l: start=i;break
end=start
while not lines[end].strip().startswith('s_endpgm'): end+=1
k=lines[start:end]
# blocks: label lines; membership by comment "in Loop: Header=BBx" or "Loop Header"
blocks=[];cur=None
for l in k:
    if l.startswith('.LBB'):
        cur={'label':l,'ins':[]};blocks.append(cur)
    elif cur is not None:
        s=l.strip()
        if s and not s.startswith(';') and not s.startswith('.'): cur['ins'].append(s.split()[0])
hdrs=collections.Counter()
for b in blocks:
    m=re.search(r'Header=(BB\d+_\d+)',b['label'])
    if m: hdrs[m.group(1)]+=len(b['ins'])
    m=re.match(r'^\.L(BB\d+_\d+):.*Loop Header',b['label'])
    if m: hdrs[m.group(1)]+=len(b['ins'])
top=hdrs.most_common(1)[0][0]
c=collections.Counter()
for b in blocks:
    if ('Header=%s '%top in b['label']+' ') or re.match(r'^\.L%s:'%top,b['label']):
        c.update(b['ins'])
tot=sum(c.values()); valu=sum(v for kk,v in c.items() if kk.startswith('v_'))
print('loop',top,'instr',tot,'VALU',valu)
for kk,v in c.most_common(int(sys.argv[3]) if len(sys.argv)>3 else 24): print('%6d %s'%(v,kk))
