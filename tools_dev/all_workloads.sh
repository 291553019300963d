# developer helper: one bench line per workload into gpurun_out/<tag>_<workload>.json (the default chain with its CPU baseline)
T=${1:-w}
show() { python -c "
import json,sys;d=json.load(open(sys.argv[1]));r=d.get('roofline') or {}
print(sys.argv[2],'%.4g'%d['value'],'%.4f ms'%d['ms_per_step'],d.get('stage_ms'),r.get('kernel'),r.get('frac'),r.get('traffic'),(r.get('valu_issue') or {}).get('frac'),d.get('whole_step_hbm_frac'),(d.get('parity_checked') or {}).get('max_lsb'),d.get('pcie_inclusive'))" "$1" "$2"; }
python bench.py > gpurun_out/${T}_chain.json 2> gpurun_out/${T}_chain.err; show gpurun_out/${T}_chain.json chain
python -c "
import json;d=json.load(open('gpurun_out/${T}_chain.json'));print('cpu_baseline',d['cpu_baseline'])"
for w in nsx aecm rtp_chain ns_aec_8k ns_agc_mix_32k ns g711 mfft; do
  python bench.py --workload $w --no-cpu > gpurun_out/${T}_$w.json 2> gpurun_out/${T}_$w.err
  show gpurun_out/${T}_$w.json $w
done
for p in 2 4; do
  python bench.py --packets-per-step $p --steps 100 --no-cpu > gpurun_out/${T}_chain_p$p.json 2>/dev/null; show gpurun_out/${T}_chain_p$p.json "chain p$p"
done
