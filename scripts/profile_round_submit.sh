#!/bin/bash
# Submits scripts/profile_round.sh for the COMMITTED tree: refuses a dirty working tree (the profiles a round commits must describe a
# commit) and passes HEAD's hash, which ends up in profiles/traffic_per_kernel*.json and in the bench lines' traffic_source.
#   scripts/profile_round_submit.sh <tag> [gpurun timeout in s]
set -eu
TAG=${1:?tag}
TMO=${2:-2400}
cd "$(dirname "$0")/.."
if [ -n "$(git status --porcelain -- . ':!gpurun_out' ':!profiles')" ]; then
    echo "profile_round_submit: the working tree has uncommitted changes; commit first" >&2
    git status --short | head -20 >&2
    exit 1
fi
make -C citlab-article-separation-new_amd/csrc -j4 > /dev/null
COMMIT=$(git rev-parse --short HEAD)
exec gpurun --timeout "$TMO" -- "bash scripts/profile_round.sh $TAG $COMMIT > gpurun_out/${TAG}_round.log 2>&1; tail -5 gpurun_out/${TAG}_round.log"
