export TMPDIR=/tmp
for wb in 16 17; do
rm -rf gpurun_out/kc$wb
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kc$wb -o kc -- python3 bench.py --in-process --inflight 1 --no-single-stream --no-cpu-baseline --proof-k 0 --proofs 0 --window-bits $wb --full-out "" > gpurun_out/kc$wb.json 2> gpurun_out/kc$wb.err
echo "== window $wb"; python3 - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/kc$wb/kc_kernel_stats.csv')))
for r in rows[:14]:
    print("%-28s calls %5s avg %9.1f us" % (r['Name'].split('<')[0].replace('void ','')[:28], r['Calls'], float(r['AverageNs'])/1e3))
PY
find gpurun_out/kc$wb -name "*kernel_trace.csv" -delete
done
