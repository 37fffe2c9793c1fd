#!/bin/bash
# fast and slow processes of the same binary: is the difference inside the library's blocking calls (trips: PCIe / device side) or between them (the core's own work)?
for i in 1 2 3 4 5 6 7 8; do
  SPARTAN_OPTIONS=testing.unlock=1,host.callstats=1 BENCH_NO_GATHER_PROBE=1 python bench.py --no-cpu-baseline --concurrent 0 --steps 20 --warmup 2 --no-side-metrics --no-strong > /tmp/o.json 2> /tmp/o.err
  python - <<'PY'
import json,re
d=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1])
rows={}
for line in open('/tmp/o.err'):
    m=re.match(r"\[callstats\]\s+(\S+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)", line)
    if m: rows[m.group(1)]=(int(m.group(2)), float(m.group(3)), float(m.group(4)))   # the last table wins (the last proof)
tot=sum(v[1] for v in rows.values())
g=lambda k: rows.get(k,(0,0,0))
print("ms/proof %.2f | inside library %.2f | job_wait %.2f bind2 %.1f us ipa_round %.1f us bind_eq %.1f us commit_small %.1f us eq_expand %.1f us" % (d['ms_per_step'], tot, g('sp_job_wait')[1], g('sp_sumcheck_bind2_eval_batched')[2], g('sp_ipa_round_lr')[2], g('sp_sumcheck_bind_eval_batched_eq')[2], g('sp_host_commit_small')[2], g('sp_eq_expand')[2]))
PY
done
