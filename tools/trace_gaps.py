import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
ev=sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]) for r in rows))
t0=ev[0][0]
last_end=ev[0][1]; last_name=ev[0][2]
for s,e,n in ev[1:]:
    gap=(s-last_end)/1e6
    if gap>8:
        print(f"gap {gap:7.1f} ms at t={ (last_end-t0)/1e6:9.1f} ms  after [{last_name}] before [{n}]")
    if e>last_end: last_end=e; last_name=n
