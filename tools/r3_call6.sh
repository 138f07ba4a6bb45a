mkdir -p gpurun_out/r3
(python -m pytest tests -m gpu -q 2>&1 | tail -30) > gpurun_out/r3/gputest4.txt
python bench.py > gpurun_out/r3/bench2.json 2> gpurun_out/r3/bench2.err
grep -E "passed|failed|FAILED" gpurun_out/r3/gputest4.txt | tail -5
python -c "
import json; d=json.load(open('gpurun_out/r3/bench2.json'))
print(d['value'], d['ms_per_step'], d['mfma_frac_end_to_end'], d['roofline']['frac'])
print({k:round(v['avg_ms'],4) for k,v in d['kernels'].items()})
print(d['h2d_inclusive'].get('frac_of_resident'), d['parity_vs_oracle_sample']['flips'], d['parity_vs_oracle_sample']['flips_unexplained'])
"
