#!/bin/bash
OUT=gpurun_out/r06; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
( time python -m pytest tests -q -m gpu -x ) > $OUT/pytest_full.log 2>&1; tail -4 $OUT/pytest_full.log
bash tools/profile_round.sh r06 > $OUT/profile_round.log 2>&1; tail -3 $OUT/profile_round.log
