#!/bin/bash
# LDS bank conflicts per kernel (SQ_LDS_BANK_CONFLICT cycles against SQ_ACTIVE_INST_LDS cycles), one step at a time and one k = 17 proof:
#   bash tools/lds_conflicts.sh > gpurun_out/lds_conflicts.txt
export TMPDIR=/tmp
out=gpurun_out/ldsc; rm -rf $out; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $out/step -- python3 bench.py --in-process --inflight 1 --no-single-stream --no-cpu-baseline --proof-k 0 --proofs 0 --steps 5 --warmup 2 --preheat-s 0 --full-out "" > $out/step.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $out/proof -- python3 tools/profile_native_proof.py 17 delay_enc 3 > $out/proof.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for tag in ("step", "proof"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob("gpurun_out/ldsc/%s/**/*counter_collection.csv" % tag, recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if (r["Dispatch_Id"]) not in seen: seen.add(r["Dispatch_Id"]); cnt[k] += 1
    print("== %s: kernel, launches, LDS-active cycles per launch, bank-conflict cycles per launch, conflict share" % tag)
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_LDS_BANK_CONFLICT", 0)):
        n = max(cnt[k], 1); a = v.get("SQ_ACTIVE_INST_LDS", 0) / n; c = v.get("SQ_LDS_BANK_CONFLICT", 0) / n
        if a + c == 0: continue
        print("%-46s %5d %14.0f %14.0f %6.1f %%" % (k, n, a, c, 100 * c / max(a, 1)))
PY
