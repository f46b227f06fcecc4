import re,collections,sys
rows=[l for l in open(sys.argv[1]) if ' CONV ' in l]
g=collections.defaultdict(lambda:[0,0,0.0,0.0])
for l in rows:
    m=re.search(r'(\S+)us\s+(\S+)TF\s+(\S+)GB/s\s+N16 (\d+)x(\d+) K(\d+) Co(\d+) t(\d+) is(\d) os(\d)',l)
    us=float(m.group(1)); tf=float(m.group(2)); gb=float(m.group(3)); H=int(m.group(4)); t=int(m.group(8))
    kind=('fwd' if l.startswith('fwd') else 'dgrad')
    key=(kind,t if t in(1,9) else 'par',H)
    g[key][0]+=us; g[key][1]+=1; g[key][2]+=tf*us; g[key][3]+=gb*us
tot=0
for k,v in sorted(g.items(), key=lambda kv:-kv[1][0]):
    tot+=v[0]
    print(k, f"{v[0]:.0f}us n={v[1]} avg={v[0]/v[1]:.0f}us avgTF={v[2]/v[0]:.0f} avgGB/s={v[3]/v[0]:.0f}")
print(tot)
