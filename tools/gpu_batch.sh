#!/bin/bash
# One gpurun call = one batch: tools/gpu_batch.sh <name> ; the batch body lives in tools/batches/<name>.sh (scratch, not shipped)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
bash "tools/batches/$1.sh" > "gpurun_out/batch_$1.log" 2>&1
echo "batch $1 rc=$?" >> "gpurun_out/batch_$1.log"
tail -c 6000 "gpurun_out/batch_$1.log"
