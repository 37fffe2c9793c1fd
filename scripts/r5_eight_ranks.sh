mkdir -p gpurun_out/r5i
env -u RANK -u WORLD_SIZE -u LOCAL_RANK BENCH_DIST_BACKEND=gloo BENCH_FORCE_DEVICE=0 BENCH_NO_PROF=1 timeout 900 python bench.py --gpus 8 --log2-cons 12 --steps 2 --warmup 1 --strong-log2 10,12,14 > gpurun_out/r5i/eight_ranks.json 2> gpurun_out/r5i/eight_ranks.err
echo rc=$?
tail -c 600 gpurun_out/r5i/eight_ranks.err
python3 -c "
import json
l=[x for x in open('gpurun_out/r5i/eight_ranks.json') if x.startswith('{')]
j=json.loads(l[-1]); print(j['n_gpus'], j['n_ranks_seen'], j['scaling'], round(j['value']), j['ms_per_step'])
for s in j['strong']: print({k:s.get(k) for k in ('log2_cons','ms_per_step','ms_per_step_unsharded','speedup_vs_unsharded','all_gathers_per_proof','all_gather_bytes_per_proof','amdahl','error','shards')})
"
