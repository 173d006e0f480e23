"""HBM traffic of the persistent AR decode engine from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of
`bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras`: takes the full-batch launches of t2s_mega_kernel (the
longest dispatches), applies the gfx950 read correction and writes the summary JSON that bench.py quotes as roofline.traffic.

usage: mega_traffic.py <fetch_dir> <write_dir> <out_json> <algorithmic_bytes>"""
import csv
import glob
import json
import sys


def rows(d, counter):
    f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
    out = []
    for r in csv.DictReader(open(f)):
        if "t2s_mega_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            out.append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return out


def main():
    fd, wd, oj, alg = sys.argv[1:5]
    fr, wr = rows(fd, "FETCH_SIZE"), rows(wd, "WRITE_SIZE")
    # the B = 32 launches: selected by the counter itself -- the single-utterance latency passes of the bench run the same
    # 100 steps in almost the same time (the engine is latency-bound), but move a small fraction of the bytes
    fmax = max(v for v, _ in fr)
    fr = [r for r in fr if r[0] > 0.5 * fmax]
    wmax = max(v for v, _ in wr)
    wr = [r for r in wr if r[0] > 0.5 * wmax]
    fk = sum(v for v, _ in fr) / len(fr)
    wk = sum(v for v, _ in wr) / len(wr)
    dur = sum(t for _, t in fr) / len(fr) / 1e3
    traffic = int(2 * fk * 1024 + wk * 1024)
    alg = int(alg)
    json.dump({
        "kernel": "t2s_mega_kernel (persistent AR decode engine, 100 steps per launch, B = 32)",
        "command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 2 --warmup 1 "
                   "--no-cpu-baseline --no-extras (WRITE_SIZE in a second, separate pass); reduced with tools/mega_traffic.py",
        "dispatches_averaged": len(fr), "FETCH_SIZE_KB_avg": round(fk, 2), "WRITE_SIZE_KB_avg": round(wk, 2),
        "avg_duration_us_under_pmc": round(dur, 1),
        "gfx950_correction": "FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads: read bytes = 2 * FETCH_SIZE * 1024 "
                             "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE * 1024 as is.  The counters are the L2s' memory-side "
                             "requests: weight re-reads served by the Infinity Cache are counted, L2 hits are not",
        "traffic_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": alg,
        "traffic_over_algorithmic": round(traffic / alg, 4)}, open(oj, "w"), indent=1)
    print(open(oj).read())


if __name__ == "__main__":
    main()
