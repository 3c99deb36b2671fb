import re,sys
f=sys.argv[1]; pat=sys.argv[2] if len(sys.argv)>2 else ''
name=None
for line in open(f):
    m=re.match(r'^(_Z[A-Za-z0-9_]+):',line)
    if m: name=m.group(1)
    m=re.match(r'^; (NumVgprs|NumAgprs|ScratchSize|Occupancy): (\d+)',line)
    if m and name:
        d=globals().setdefault('cur',{}); d[m.group(1)]=int(m.group(2))
        if m.group(1)=='Occupancy':
            if pat in name: print(name[:70], d)
            globals()['cur']={}
