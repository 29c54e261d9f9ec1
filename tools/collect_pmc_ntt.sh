#!/bin/bash
# HBM traffic of k_ntt_pass (23 x 2^19 bn256::Fr, clocks up): FETCH_SIZE and WRITE_SIZE in separate --pmc passes with --kernel-trace only, the same two passes over
# tools/pmc_calib (known byte counts: a 16-B-per-lane stream and a 64-B gather), per-pass means of the LAST ten transforms.   usage: tools/collect_pmc_ntt.sh
set -e
out=gpurun_out/pmc_ntt; rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
[ -x tools/pmc_calib ] || hipcc -O3 --offload-arch=gfx950 tools/pmc_calib.hip -o tools/pmc_calib
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/main_$c -- python3 tools/ntt_pmc.py 20 > $out/main_$c.log 2>&1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/calib_$c -- ./tools/pmc_calib > $out/calib_$c.log 2>&1
done
python3 - <<'PY'
import csv, glob, json
from collections import defaultdict
def per_dispatch(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc, names = defaultdict(float), {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            acc[int(r["Dispatch_Id"])] += float(r["Counter_Value"]); names[int(r["Dispatch_Id"])] = r["Kernel_Name"]
    return [(names[i], acc[i]) for i in sorted(acc)]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    ntt = [v for n, v in per_dispatch("gpurun_out/pmc_ntt/main_" + c, c) if "k_ntt_pass" in n]
    plain = ntt[3 * 20: 3 * 25]                      # the five plain transforms after the warm-up (three passes each)
    coset = ntt[3 * 25:]                             # five coeff_to_extended
    out[c] = {"plain_by_pass_KiB": [sum(plain[p::3]) / 5 for p in range(3)], "coset_by_pass_KiB": [sum(coset[p::3]) / max(1, len(coset) // 3) for p in range(3)]}
    cal = dict((n.split("(")[0], v) for n, v in per_dispatch("gpurun_out/pmc_ntt/calib_" + c, c))
    out[c]["calib_KiB"] = {k.replace("void ", ""): v for k, v in cal.items()}
elems = 23 << 19
alg = elems * 32
rep = {"workload": "23 x 2^19 bn256::Fr, three passes; per-pass means of five transforms after 20 warm-up transforms", "algorithmic_bytes_read_per_pass": alg, "algorithmic_bytes_written_per_pass": alg, "raw": out}
g = next((v for k, v in out["FETCH_SIZE"]["calib_KiB"].items() if "k_gather64" in k), None)
s = next((v for k, v in out["FETCH_SIZE"]["calib_KiB"].items() if "k_stream16" in k), None)
rep["fetch_factor_gather64"] = g * 1024 / ((1 << 24) * 68) if g else None
rep["fetch_factor_stream16"] = s * 1024 / ((1 << 24) * 64) if s else None
json.dump(rep, open("gpurun_out/pmc_ntt/ntt_traffic.json", "w"), indent=1)
for kind in ("plain", "coset"):
    for p in range(3):
        f, w = out["FETCH_SIZE"][kind + "_by_pass_KiB"][p] * 1024, out["WRITE_SIZE"][kind + "_by_pass_KiB"][p] * 1024
        print("%s pass %d: FETCH_SIZE %.1f MB raw (%.1f MB at the stream factor %.3f, %.1f MB at the gather factor %.3f) | WRITE_SIZE %.1f MB | algorithmic %.1f + %.1f MB" %
              (kind, p, f / 1e6, f / 1e6 / (rep["fetch_factor_stream16"] or 1), rep["fetch_factor_stream16"] or 0, f / 1e6 / (rep["fetch_factor_gather64"] or 1), rep["fetch_factor_gather64"] or 0, w / 1e6, alg / 1e6, alg / 1e6))
PY
