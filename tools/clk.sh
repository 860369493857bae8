#!/bin/bash
rocm-smi --showclocks 2>&1 | grep -E "sclk|mclk|fclk" | head -5
rocm-smi --showperflevel 2>&1 | grep -i perf | head -3
python tools/tune2.py lukvle1_1e4 2>&1 | grep -v amdgpu.ids | grep -v leaf_cols
echo "--- set perf level high"
rocm-smi --setperflevel high 2>&1 | tail -2
rocm-smi --showclocks 2>&1 | grep -E "sclk" | head -3
python tools/tune2.py lukvle1_1e4 grid_1e5 2>&1 | grep -v amdgpu.ids | grep -v leaf_cols
rocm-smi --setperflevel auto 2>&1 | tail -1
