#!/bin/bash
timeout 600 python -m pytest tests/test_generator_gpu.py -q -m gpu -k "batch8" -s 2>&1 | grep -v "Warning\|warn\|stage\[" | tail -20
