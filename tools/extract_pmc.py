"""Reduce a rocprofv3 --pmc counter_collection.csv to the rows of one kernel (keeps profiles/ small)."""
import csv
import glob
import sys

src_dir, kernel_substr, counter, out = sys.argv[1:5]
f = glob.glob(src_dir + "/**/*_counter_collection.csv", recursive=True)[0]
with open(out, "w") as o:
    o.write("dispatch_index,kernel,counter,value,duration_ns\n")
    i = 0
    for r in csv.DictReader(open(f)):
        if kernel_substr in r["Kernel_Name"] and r["Counter_Name"] == counter:
            o.write(f"{i},{kernel_substr},{counter},{r['Counter_Value']},{int(r['End_Timestamp']) - int(r['Start_Timestamp'])}\n")
            i += 1
print("wrote", i, "rows to", out)
