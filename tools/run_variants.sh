set -u
O=gpurun_out/r03; mkdir -p $O
{
echo "# configs[2] B=65536, contract family / passing: one launch (STAGED=-1) vs resumable (automatic) ; axis_aligned 0 (scan) / 1 (promised) / -1 (general path)"
for st in -1 0; do STAGED=$st python tools/quick_rate.py cfg2 65536 2 2>&1 | tail -1; STAGED=$st FAMILY=passing python tools/quick_rate.py cfg2 65536 3 2>&1 | tail -1; done
for ax in 1 -1; do AX=$ax python tools/quick_rate.py cfg2 65536 2 2>&1 | tail -1; done
echo "# configs[1] B=1024 (latency kernel W=3): one launch vs resumable (automatic); axis path"
for st in -1 0; do STAGED=$st python tools/quick_rate.py cfg1 1024 6 2>&1 | tail -1; STAGED=$st FAMILY=passing python tools/quick_rate.py cfg1 1024 6 2>&1 | tail -1; done
python tools/ab_cfg1.py 2>&1 | tail -3
} > $O/variants.txt 2>&1
python tools/exp_polish2.py cfg2 1024 2>&1 | grep -v amdgpu.ids > $O/exp_polish_cfg2.jsonl
python tools/exp_polish2.py cfg1 1024 2>&1 | grep -v amdgpu.ids > $O/exp_polish_cfg1.jsonl
python tools/exp_cfg1_order2.py 2>&1 | grep -v amdgpu.ids > $O/exp_cfg1_order2.txt
python -m pytest tests -q -m gpu --durations=12 > $O/gpu_suite.log 2>&1; grep -E "passed|failed" $O/gpu_suite.log | tail -2
