#!/bin/bash
# first GPU call of round 6: correctness of the new layout / builds, then the A/B probe
mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "every_build or bound_heldout or lossless or every_packed or default_packs or snapshot" > gpurun_out/r06/pytest_variants.log 2>&1
tail -5 gpurun_out/r06/pytest_variants.log
python -m pytest tests/test_gpu_golden.py tests/test_gpu_tiled.py -x -q -m gpu > gpurun_out/r06/pytest_golden_tiled.log 2>&1
tail -3 gpurun_out/r06/pytest_golden_tiled.log
bash tools/variant_probe.sh r06 2 > gpurun_out/r06/variant_probe.log 2>&1
cat gpurun_out/r06/variant_probe.jsonl
