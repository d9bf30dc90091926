#!/bin/sh
python -m pytest tests/test_gpu_api.py -q --tb=short -k only_enqueues 2>&1 | grep -v "^$" | grep "^FAILED\|passed\|failed\|\.py:[0-9]*: in\|Error" | head -20
for i in 1 2 3; do python -m pytest tests/test_gpu_manipulate.py -q --tb=line -k polling 2>&1 | tail -3; done
