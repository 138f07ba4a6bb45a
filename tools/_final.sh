mkdir -p gpurun_out/r3
(python -m pytest tests -m gpu -q 2>&1 | tail -12) > gpurun_out/r3/gputest7.txt
bash tools/prof_bench.sh r03 > gpurun_out/r3_prof.log 2>&1
bash tools/prof_bench_pmc.sh > gpurun_out/r3_prof_pmc.log 2>&1
python bench.py > gpurun_out/r3/bench5.json 2> gpurun_out/r3/bench5.err
grep -E "passed|failed|FAILED" gpurun_out/r3/gputest7.txt
python -c "
import json
d=json.load(open('gpurun_out/r3/bench5.json')); print(d['value'], d['ms_per_step'], d['mfma_frac_end_to_end'], d['roofline']['frac'], d['roofline'].get('frac_rocprof'))
print({k:round(v['avg_ms'],4) for k,v in d['kernels'].items()})
print(d['h2d_inclusive'].get('frac_of_resident'), d['parity_vs_oracle_sample']['flips'], d['parity_vs_oracle_sample']['flips_unexplained'])
"
