for w in 3 4; do for f in 1 4; do
python bench.py --no-cpu-baseline --proof-k 0 --proofs 0 --no-single-stream --acc-waves $w --inflight $f --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('waves $w inflight $f', d['value'], d['ms_per_step'], d['breakdown_ms_per_step'])"
done; done
