mkdir -p gpurun_out/r3
bash tools/prof_bench.sh r03 > gpurun_out/r3_prof.log 2>&1
bash tools/prof_bench_pmc.sh > gpurun_out/r3_prof_pmc.log 2>&1
python bench.py > gpurun_out/r3/bench4.json 2> gpurun_out/r3/bench4.err
python bench.py --no-extras --cpu-images 0 --steps 50 --warmup 3 > gpurun_out/r3/bench4_steps50.json 2>/dev/null
python -c "
import json
for f in ('gpurun_out/r3/bench4.json','gpurun_out/r3/bench4_steps50.json'):
    d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d['mfma_frac_end_to_end'], d['roofline']['frac'], d['roofline'].get('frac_rocprof'))
d=json.load(open('gpurun_out/r3/bench4.json'))
print({k:round(v['avg_ms'],4) for k,v in d['kernels'].items()})
print(d['h2d_inclusive'].get('frac_of_resident'), d['parity_vs_oracle_sample']['flips'], d['parity_vs_oracle_sample']['flips_unexplained'], d['cpu_baseline']['value'], d['cpu_baseline']['reference_module']['value'])
"
