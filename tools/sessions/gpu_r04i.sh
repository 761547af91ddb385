#!/bin/bash
set -o pipefail
O=gpurun_out/r04h; mkdir -p $O
python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_parity.py::test_assemble_matches_calcGSM > $O/gpu_tests2.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests2.log; tail -4 $O/gpu_tests2.log
