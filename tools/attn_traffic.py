"""HBM traffic of the decode-attention kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of
`bench.py --steps 1 --warmup 0 --no-cpu-baseline`: averages the last 240 full-batch dispatches (gsv_t2s_time_step's
back-to-back launches at the end-of-run cache length), applies the gfx950 read correction and writes the summary JSON
(+ a per-dispatch CSV) that bench.py quotes as roofline.traffic.

usage: attn_traffic.py <fetch_dir> <write_dir> <out_json> <out_csv> <algorithmic_bytes>"""
import csv
import glob
import json
import sys


def rows(d, counter):
    f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
    out = []
    for r in csv.DictReader(open(f)):
        if "decode_attn_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            grid = int(r.get("Grid_Size", 0) or 0)
            out.append((grid, float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return out


def main():
    fd, wd, oj, oc, alg = sys.argv[1:6]
    fr, wr = rows(fd, "FETCH_SIZE"), rows(wd, "WRITE_SIZE")
    gmax = max(g for g, _, _ in fr)
    fr = [r for r in fr if r[0] == gmax][-240:]
    wr = [r for r in wr if r[0] == gmax][-240:]
    fk = sum(v for _, v, _ in fr) / len(fr)
    wk = sum(v for _, v, _ in wr) / len(wr)
    dur = sum(t for _, _, t in fr) / len(fr) / 1e3
    traffic = int(2 * fk * 1024 + wk * 1024)
    alg = int(alg)
    json.dump({
        "kernel": "decode_attn_kernel<_Float16,32> (K/V loaded non-temporal)",
        "command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 1 --warmup 0 "
                   "--no-cpu-baseline (WRITE_SIZE in a second, separate pass); reduced with tools/attn_traffic.py",
        "dispatches_averaged": len(fr),
        "selection": "the last 240 full-batch dispatches = gsv_t2s_time_step's back-to-back launches at the end-of-run cache length",
        "FETCH_SIZE_KB_avg": round(fk, 2), "WRITE_SIZE_KB_avg": round(wk, 2), "avg_duration_us_under_pmc": round(dur, 2),
        "gfx950_correction": "FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads: read bytes = 2 * FETCH_SIZE * 1024 "
                             "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE * 1024 as is",
        "traffic_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": alg,
        "traffic_over_algorithmic": round(traffic / alg, 4)}, open(oj, "w"), indent=1)
    with open(oc, "w") as o:
        o.write("b32_dispatch_index,FETCH_SIZE_KB,WRITE_SIZE_KB,duration_ns\n")
        for i, (a, b) in enumerate(zip(fr, wr)):
            o.write(f"{i},{a[1]},{b[1]},{a[2]}\n")
    print(open(oj).read())


if __name__ == "__main__":
    main()
