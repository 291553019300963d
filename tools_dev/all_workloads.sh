# developer helper: one bench line per workload into gpurun_out/<tag>_<workload>.json
T=${1:-w}
for w in ns_aec_8k ns_agc_mix_32k ns g711 mfft; do
  python bench.py --workload $w --steps 40 --warmup 8 --no-cpu > gpurun_out/${T}_$w.json 2> gpurun_out/${T}_$w.err
  python -c "
import json;d=json.load(open('gpurun_out/${T}_$w.json'));print('$w',d['value'],d['ms_per_step'],d.get('stage_ms'),d['roofline']['kernel'],d['roofline']['frac'],d.get('whole_step_hbm_frac'))"
done
python bench.py --packets-per-step 2 --steps 40 --warmup 8 --no-cpu > gpurun_out/${T}_chain_p2.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/${T}_chain_p2.json'));print('chain p2',d['value'],d['ms_per_step'])"
python bench.py --packets-per-step 4 --steps 40 --warmup 8 --no-cpu > gpurun_out/${T}_chain_p4.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/${T}_chain_p4.json'));print('chain p4',d['value'],d['ms_per_step'])"
