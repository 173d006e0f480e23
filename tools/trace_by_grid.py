import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
d=collections.defaultdict(list)
for r in rows:
    k=r["Kernel_Name"][:58]; g=(r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Grid_Size_Y"))
    d[(k,g)].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
out=sorted(d.items(), key=lambda kv:-sum(kv[1]))
for (k,g),v in out[:int(sys.argv[2]) if len(sys.argv)>2 else 14]:
    print(f"{k:58s} grid={g} n={len(v)} avg={sum(v)/len(v)/1e3:.1f}us min={min(v)/1e3:.1f} tot={sum(v)/1e6:.2f}ms")
