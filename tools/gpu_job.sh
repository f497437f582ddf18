#!/bin/bash
# Run a command on the GPU box from a FROZEN copy of the tree: gpurun snapshots /root/repo only after its queue wait, so
# edits made while a call is queued would otherwise travel half-finished.  The copy lives under .stage/<unique>/ (ignored by
# git, part of the snapshot); the command runs inside it with GRAFT_REPO_ROOT pointing at it and gpurun_out/ linked to the
# real one, so results are merged back as usual.  A job removes its OWN copy when gpurun returns (trap) and, on start, only
# copies older than six hours (left by a killed shell): a second job issued while the first is still queued never deletes the
# first one's tree.
#   tools/gpu_job.sh <timeout_s> '<command>'
set -eu
T=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/.stage"
find "$ROOT/.stage" -mindepth 1 -maxdepth 1 -type d -mmin +360 -exec rm -rf {} + 2>/dev/null || true
S=$(mktemp -d "$ROOT/.stage/job.XXXXXX")
N=$(basename "$S")
trap 'rm -rf "$S"' EXIT
(cd "$ROOT" && tar cf - --exclude=./.git --exclude=./gpurun_out --exclude=./.stage --exclude=__pycache__ --exclude=./.pytest_cache \
  --exclude=./video_dqn_amd/lib/obj --exclude=./experiments --exclude='./profiles/*.csv' .) | (cd "$S" && tar xf -)
ln -s ../../gpurun_out "$S/gpurun_out"
rc=0
/usr/local/graft/bin/gpurun --timeout "$T" -- "mkdir -p gpurun_out && cd .stage/$N && export GRAFT_REPO_ROOT=\$PWD && $*" || rc=$?
exit $rc
