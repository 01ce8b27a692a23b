cd $GRAFT_REPO_ROOT
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 ) > gpurun_out/r06_zz_gputests.txt
tail -2 gpurun_out/r06_zz_gputests.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r06_zz_bench_driver_args.json 2> gpurun_out/r06_zz_bench.err; python -c "
import json
d=json.loads([l for l in open('gpurun_out/r06_zz_bench_driver_args.json') if l.startswith('{')][-1])
print('driver-args line: ms_per_step', d['ms_per_step'], 'value', d['value'], 'kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'], d['config']['host_wait'])
"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
