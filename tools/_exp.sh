python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_pins_gpu.py -x -q -m gpu 2>&1 | tail -2
python bench.py --steps 400 --warmup 20 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], l['batch8'], l['roofline']['cost_volume_b8_f16']['us'], l['roofline']['us_per_launch'], l['hires'])"
