# developer helper: VAD / AGC parity tests + the default bench line
T=${1:-t}
python -m pytest tests/test_vadagc_gpu.py tests/test_configs_gpu.py tests/test_edges_gpu.py -m gpu -x -q > gpurun_out/${T}_pytest.log 2>&1; echo pytest rc=$?; tail -3 gpurun_out/${T}_pytest.log
python bench.py --steps 40 --warmup 8 --no-cpu > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; python -c "
import json;d=json.load(open('gpurun_out/${T}_bench.json'));print(d['ms_per_step'],d['stage_ms'],d['roofline']['frac'])"
