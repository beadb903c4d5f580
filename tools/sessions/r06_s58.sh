#!/bin/bash
# round 6, GPU session 58: the kernel formulations the default build leaves out (make EXPERIMENTAL=1: stream kernel, queued renderer, mode 3,
# trace-rate experiment) built from the final tree — their tests (marker `formulations`) and the whole GPU suite on that library
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s58; mkdir -p $O
export MCPT_LIB=$PWD/monte-carlo-path-tracing_amd/exp/formulations/libmcpt_hip.so
timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_suite_experimental_library.log 2>&1
echo "pytest rc=$?" >> $O/gpu_suite_experimental_library.log
tail -5 $O/gpu_suite_experimental_library.log
