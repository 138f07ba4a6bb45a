mkdir -p gpurun_out/r3
(python -m pytest tests -m gpu -q 2>&1 | tail -30) > gpurun_out/r3/gputest5.txt
python bench.py > gpurun_out/r3/bench3.json 2> gpurun_out/r3/bench3.err
grep -E "passed|failed|FAILED|skipped" gpurun_out/r3/gputest5.txt | tail -5
python -c "
import json; d=json.load(open('gpurun_out/r3/bench3.json'))
print(d['value'], d['ms_per_step'], d['mfma_frac_end_to_end'], d['roofline']['frac'], d['roofline'].get('frac_rocprof'), d.get('fp16_range_alarm_rows'))
print(d['h2d_inclusive'].get('frac_of_resident'), d['parity_vs_oracle_sample']['flips'], d['parity_vs_oracle_sample']['flips_unexplained'])
"
