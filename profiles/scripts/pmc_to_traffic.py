"""HBM bytes per launch of the W passes from two rocprofv3 counter passes over the bench itself:

   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT/pmc_fetch -- python3 bench.py --steps 5 --warmup 12 --no-cpu-baseline
   rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d OUT/pmc_write -- python3 bench.py --steps 5 --warmup 12 --no-cpu-baseline
   python profiles/scripts/pmc_to_traffic.py FETCH.csv WRITE.csv n > profiles/w_pass_traffic.json

Corrections as MI355X_MICROARCH.md (HBM / rocprofv3 section) prescribes for gfx950: both counters are
in KiB; FETCH_SIZE reports half of the bytes of a wide coalesced streaming read (x2); WRITE_SIZE is
exact.  A kernel is launched in more than one shape during a bench run (the back-to-back timing
launches after the timed region store one vector less than the in-iteration launches): the launches
of the in-iteration shape are the ones whose WRITE_SIZE is within 1 % of the largest seen.
"""
import collections
import csv
import json
import statistics
import sys

import os

# PP=1 in the environment: the bench was driven through the ping-pong entry (3 store streams in the
# storing pass: 218 B/row instead of 234); UB=7: uniform bounds detected (l, u, nbd not streamed: 17 B/row
# less in each of the two passes of the iteration)
PP = os.environ.get("PP", "1") == "1"
UB = int(os.environ.get("UB", "0"))
_B = (0 if UB & 1 else 8) + (0 if UB & 2 else 8) + (0 if UB & 4 else 1)
_S = "_ub%d" % UB if UB else ""
KERNELS = {
    "update_scan" + _S: ("update_scan_kernel<double, 10, true, true, true", lambda n: (177 + _B) * n),
    ("subsm_update_pp" if PP else "subsm_update") + _S: ("subsm_update_kernel<double, 10, true, true, false>",
                                                         lambda n: ((201 if PP else 217) + _B) * n),
    "cmprlb_wtv": ("cmprlb_wtv_kernel<double, 10, true, true, true, false>", lambda n: 177 * n),
    "wtv": ("wtv_kernel<double, 10, true>", lambda n: 168 * n),
}


def read(path, counter):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            d[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return d


def main():
    fetch, write, n = read(sys.argv[1], "FETCH_SIZE"), read(sys.argv[2], "WRITE_SIZE"), int(sys.argv[3])
    out = {}
    for key, (sub, alg) in KERNELS.items():
        fk = [k for k in fetch if sub in k]
        wk = [k for k in write if sub in k]
        if not fk or not wk:
            continue
        f, w = fetch[fk[0]], write[wk[0]]
        wmax = max(w)
        fmax = max(f)
        f_in = [v for v in f if v >= 0.99 * fmax]  # (launches of shorter shapes read less)
        # (a read-only pass writes its partial sums only: no shapes to tell apart there)
        w_in = [v for v in w if v >= 0.99 * wmax] if wmax > 0.01 * fmax else w
        fb = 2.0 * 1024.0 * statistics.median(f_in)
        wb = 1024.0 * statistics.median(w_in)
        out[key] = {
            "kernel": sub, "n": n, "col": 10,
            "launches_seen": len(f), "launches_of_the_in_iteration_shape": len(w_in),
            "FETCH_SIZE_KiB_raw_median": statistics.median(f_in),
            "WRITE_SIZE_KiB_raw_median": statistics.median(w_in),
            "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of a wide coalesced streaming read "
                          "(MI355X_MICROARCH.md, HBM) -> x2; WRITE_SIZE exact; KiB -> x1024",
            "hbm_read_bytes_per_launch": fb, "hbm_write_bytes_per_launch": wb,
            "hbm_bytes_per_launch": fb + wb, "hbm_bytes_per_row": (fb + wb) / n,
            "algorithmic_bytes_per_launch": alg(n),
            "traffic_over_algorithmic": (fb + wb) / alg(n),
        }
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
