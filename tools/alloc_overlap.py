import re,sys
live={}
bad=0
for line in open(sys.argv[1]):
    m=re.match(r"alloc (\S+) \+(\d+): \[(\S+), (\S+)\) (\d+) bytes",line)
    if m:
        ctx,off,lo,hi=m.group(1),m.group(2),int(m.group(3),16),int(m.group(4),16)
        for k,(l2,h2) in live.items():
            if lo<h2 and l2<hi:
                bad+=1; print("OVERLAP", (ctx,off,hex(lo),hex(hi)), "with", k, hex(l2),hex(h2))
        live[(ctx,off)]=(lo,hi); continue
    m=re.match(r"free (\S+) \+(\d+):",line)
    if m: live.pop((m.group(1),m.group(2)),None)
print("allocs live",len(live),"overlaps",bad)
