import sys,re
names=["top","postB","bar1","postC","q0w","q0","q1w","q1","q2w","q2","q3w","q3","q4w","q4","q5w","q5","qend","interpT","planeW","postD","bar2","postE","bar3"]
for line in sys.stdin:
    m=re.match(r"PROF launch (\d+) wave (\d+) layer (\d+): (.*)",line)
    if not m: 
        if line.startswith("exp"): print(line.strip())
        continue
    v=[int(x) for x in m.group(4).split()]
    v=[(x+2**32)%2**32 for x in v]
    base=v[0]
    d=[(v[i]-v[i-1])%2**32 for i in range(1,len(v))]
    print("L%s w%s l%s tot=%6d | "%(m.group(1),m.group(2),m.group(3),(v[22]-v[0])%2**32)+" ".join("%s=%d"%(names[i+1],d[i]) for i in range(len(d))))
