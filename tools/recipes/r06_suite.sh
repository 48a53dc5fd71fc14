#!/bin/bash
# ON THE GPU BOX: the whole -m gpu suite, timed.  gpurun -- bash tools/recipes/r06_suite.sh <tag> [pytest args]
set -u
export TMPDIR=/tmp
TAG=$1; shift
mkdir -p gpurun_out
( time python -m pytest tests -m gpu -q -x "$@" ) > gpurun_out/suite_$TAG.log 2>&1
echo "rc $?" >> gpurun_out/suite_$TAG.log
tail -15 gpurun_out/suite_$TAG.log
